#!/usr/bin/env python3
"""Per-shape timing of eavqa_gemm on the hot path's shapes (cfg2: M = 64 x 42 = 2688).

    python tools/gemm_bench.py [--iters 50] [--general]

Prints TFLOP/s per (M, N, K) from HIP events around `iters` back-to-back launches on random data.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from eavqa_amd import _lib, ops

SHAPES = [  # (M, N, K, what)
    (32, 7680, 2560, "decode qkv"), (32, 2560, 2560, "decode proj"), (32, 10240, 2560, "decode fc1"), (32, 2560, 10240, "decode fc2"),
    (32, 50272, 2560, "decode lm_head"), (64, 12800, 6400, "mlp fc2 fwd"),
    (1943, 1280, 64, "epi proj"), (1943, 3840, 64, "epi qkv"), (1943, 5120, 64, "epi fc1"), (1943, 1280, 3840, "packed da"),
    (1943, 6144, 2048, "opt13 qkv"), (1943, 2048, 2048, "opt13 proj"), (1943, 8192, 2048, "opt13 fc1"), (1943, 2048, 8192, "opt13 fc2"),
    (1943, 2048, 6144, "opt13 da"), (1303, 50272, 2048, "opt13 head"), (1303, 2048, 50304, "opt13 dhead"),
    (16448, 3072, 1024, "vitL64 qkv"), (16448, 1024, 1024, "vitL64 proj"), (16448, 4096, 1024, "vitL64 fc1"), (16448, 1024, 4096, "vitL64 fc2"),
    (2048, 12288, 4096, "opt67 qkv"), (2048, 4096, 4096, "opt67 proj"), (2048, 16384, 4096, "opt67 fc1"), (2048, 4096, 16384, "opt67 fc2"),
    (2688, 1280, 32, "fixed K=32"), (2688, 1280, 320, "fixed K=320"), (1943, 1280, 1280, "packed proj"), (1943, 3840, 1280, "packed qkv"),
    (1943, 5120, 1280, "packed fc1"), (1943, 1280, 5120, "packed fc2"),
    (2688, 3840, 1280, "qkv fwd"), (2688, 1280, 1280, "proj fwd / dctx"), (2688, 5120, 1280, "fc1 fwd / du"),
    (2688, 1280, 5120, "fc2 fwd / da2"), (2688, 1280, 3840, "dqkv->da"), (2688, 50257, 1280, "lm_head fwd"),
    (2688, 1280, 50304, "lm_head dgrad"), (3200, 2304, 768, "vit qkv"), (3200, 768, 768, "vit proj"),
    (1024, 1280, 5120, "probe 80 tiles"), (2048, 1280, 5120, "probe 160 tiles"), (2048, 2048, 5120, "probe 256 tiles"),
    (2048, 3072, 5120, "probe 384 tiles"), (2048, 4096, 5120, "probe 512 tiles"), (4096, 4096, 5120, "probe 1024 tiles"),
    (41120, 3072, 1024, "vitL qkv"), (41120, 1024, 1024, "vitL proj"), (41120, 4096, 1024, "vitL fc1"), (41120, 1024, 4096, "vitL fc2"),
    (4800, 7680, 2560, "prefill qkv"), (4800, 2560, 2560, "prefill proj"), (4800, 10240, 2560, "prefill fc1"), (4800, 2560, 10240, "prefill fc2"),
    (41120, 4096, 64, "kslope 64"), (41120, 4096, 256, "kslope 256"), (41120, 4096, 512, "kslope 512"), (41120, 4096, 2048, "kslope 2048"),
    (41120, 4096, 4096, "kslope 4096"), (41120, 4096, 4160, "kslope 4160 (odd pitch)"), (41120, 4096, 1088, "kslope 1088 (odd pitch)"), (41120, 4096, 1024, "kslope 1024"), (40960, 4096, 1024, "kslope full tiles 1024"), (8192, 8192, 1024, "one round 1024"), (8192, 8192, 4096, "one round 4096"),
    (32, 6144, 2048, "t5dec qkv"), (32, 2048, 2048, "t5dec o"), (32, 10240, 2048, "t5dec wi"), (32, 2048, 5120, "t5dec wo"),
    (160, 6144, 2048, "t5dec5 qkv"), (160, 2048, 2048, "t5dec5 o"), (160, 10240, 2048, "t5dec5 wi"), (160, 2048, 5120, "t5dec5 wo"),
    (288, 2048, 2048, "t5dec9 o"), (288, 2048, 5120, "t5dec9 wo"),
    (640, 6144, 2048, "t0enc qkv"), (640, 2048, 2048, "t0enc o"), (640, 10240, 2048, "t0enc wi"), (640, 2048, 5120, "t0enc wo"),
    (640, 2048, 6144, "t0enc dqkv"), (640, 2048, 10240, "t0enc dwi"), (640, 5120, 2048, "t0enc dwo"), (640, 4096, 2048, "t0enc crosskv"),
    (2048, 2048, 2048, "t0dec o"), (2048, 2048, 5120, "t0dec wo"), (2048, 6144, 2048, "t0dec qkv"), (2048, 10240, 2048, "t0dec wi"),
    (2048, 2048, 10240, "t0dec dwi"), (168, 768, 768, "cfg1B4 proj"), (168, 3072, 768, "cfg1B4 fc1"), (168, 768, 3072, "cfg1B4 fc2"),
    (3200, 3072, 768, "vit fc1"), (3200, 768, 3072, "vit fc2"), (4096, 4096, 4096, "square 4k"), (8192, 8192, 8192, "square 8k"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--general", action="store_true", help="force the general (register-staged) kernel")
    ap.add_argument("--epilogue", action="store_true", help="bias + gelu_new + aux_out like fc1")
    ap.add_argument("--stagger", type=int, default=0)
    ap.add_argument("--ablate", type=int, default=0, help="timing-only ablation variant of the fast kernel (wrong results)")
    ap.add_argument("--big", type=int, default=0, help="0 auto, 1 never 256x256, 2 always 256x256")
    ap.add_argument("--shape", type=int, default=0, help="0 auto, 1 never a shaped tile, 2..6 always 128x80 / 128x96 / 256x128 / 256x160 / 256x192")
    ap.add_argument("--k64", type=int, default=0, help="full-line family: 0 by cost model, 1 never (round-1 kernels), 2..12 force K64_SHAPES[id - 2] (csrc/gemm_k64.hip)")
    ap.add_argument("--group-n", type=int, default=0, help="256 x 256 kernel tile order: 0 default, 1 m fastest, 2.. column groups of value - 1")
    ap.add_argument("--epi", default="", help="epilogue of the timed call: '' plain bf16 out | 'fc1' bias + gelu_new + aux_out | 'quick' bias + quick_gelu | "
                                              "'res32' bias + fp32 residual in place | 'res16' bias + half residual in place | 'bwd' gelu_new derivative at aux_in")
    ap.add_argument("--no-split", action="store_true", help="256 x 256 kernel: keep a ragged last tile row in the same launch")
    ap.add_argument("--deep", type=int, default=0, help="8-stage ring: 0 auto, 1 never, 2 always")
    ap.add_argument("--check", action="store_true", help="compare the result with a torch matmul")
    ap.add_argument("--cold", action="store_true", help="rotate over enough copies of the weight to defeat L2 + Infinity Cache")
    ap.add_argument("--only", default="", help="comma-separated substrings of the shape names to run")
    args = ap.parse_args()
    ops.KernelSelect.gemm = (args.stagger | (args.ablate << 4) | (int(args.general) << 7) | (args.k64 << 8) | (args.big << 14) | (args.deep << 16)
                             | (args.shape << 18) | (args.group_n << 21) | (int(args.no_split) << 25))
    dev = "cuda"
    only = [w for w in args.only.split(",") if w]
    for M, N, K, what in SHAPES:
        if only and not any(w in what for w in only):
            continue
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        bs = [b]
        if args.cold:
            bs = [b.clone() for _ in range(max(2, int(6e8 / (2.0 * N * K)) + 1))]
        kw = {}
        if args.epilogue or args.epi == "fc1":
            kw = dict(bias=torch.zeros(N, device=dev), act="gelu_new", aux_out=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
        elif args.epi == "quick":
            kw = dict(bias=torch.zeros(N, device=dev), act="quick_gelu")
        elif args.epi == "bwd":
            kw = dict(act="gelu_new", aux_in=torch.randn(M, N, device=dev).to(torch.bfloat16))
        elif args.epi in ("res32", "res16"):
            out = torch.zeros(M, N, device=dev, dtype=torch.float32 if args.epi == "res32" else torch.float16)
            kw = dict(bias=torch.zeros(N, device=dev), residual=out)
        for _ in range(3):
            ops.gemm(a, b, out=out, **kw)
        if args.check and not kw:
            ref = (a[:256].float() @ b.float().T)
            err = (out[:256].float() - ref).abs().max().item()
            tail = (out[-64:].float() - a[-64:].float() @ b.float().T).abs().max().item()
            print(f"    check: max|err| first 256 rows {err:.4f}, last 64 rows {tail:.4f} (|ref| ~ {ref.abs().max().item():.2f})")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = max(5, min(args.iters, int(2e12 / (2.0 * M * N * K)) + 5))
        e0.record()
        for i in range(iters):
            ops.gemm(a, bs[i % len(bs)], out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print(f"{what:18s} M={M:5d} N={N:6d} K={K:6d}  {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
