#!/usr/bin/env python3
"""Decode-step kernels in isolation, weights / caches rotated so that every launch reads cold HBM.

    python tools/decode_bench.py [--iters 40]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eavqa_amd import ops


def timed(fn, n):
    fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--hot", action="store_true", help="reuse one weight / cache buffer (served from L2 / Infinity Cache)")
    ap.add_argument("--sweep", action="store_true", help="split-K: every admissible ks x load-window depth (8 / 16) per decode shape")
    a = ap.parse_args()
    dev, bf = "cuda", torch.bfloat16
    E, H, hd, F = 2560, 32, 80, 10240
    if a.sweep:
        for N, K, what in ((3 * E, E, "qkv"), (E, E, "proj"), (F, E, "fc1"), (E, F, "fc2")):
            nb = max(2, int(6e8 / (2.0 * N * K)) + 1)
            ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(nb)]
            x = torch.randn(32, K, device=dev).to(bf)
            mb = N * K * 2 / 1e6
            plan = ops.gemm_splitk(x, ws[0]).shape[0]
            for ks in (1, 2, 4, 5, 8, 10, 16, 20):
                if K % (32 * ks) or (K // ks) * 2 * 32 > 150 * 1024:
                    continue
                part = torch.empty((ks, 32, N), device=dev, dtype=torch.float32)
                row = []
                for nf in (1, 2):
                    for nw in (4, 8):
                        for unroll in (8, 16):
                            sel = unroll | (nw << 8) | (nf << 12)
                            us = timed(lambda i: ops.gemm_splitk(x, ws[i % nb], ks=ks, unroll=sel, out=part), a.iters)
                            row.append(f"{us:5.1f}")
                print(f"{what:5s} ks={ks:2d}{'*' if ks == plan else ' '} KS={K // ks:5d}  [cols/wave 16: 4w U8 U16 | 8w U8 U16 || cols/wave 32: 4w U8 U16 | 8w U8 U16] us: "
                      + " ".join(row) + f"   (stream floor {mb / 5.0:5.1f} us at 5 TB/s)", flush=True)
            del ws
        return
    print("-- attention decode (OPT-2.7B heads, 160 cached keys), KV rotated over 12 layers' worth")
    for B in (8, 16, 32, 64, 128):
        Smax, Sk = 169, 160
        nc = 1 if a.hot else 12
        caches = [(torch.randn(B * Smax, E, device=dev).to(bf), torch.randn(B * Smax, E, device=dev).to(bf)) for _ in range(nc)]
        q = torch.randn(B, E, device=dev).to(bf)
        us = timed(lambda i: ops.attention_fwd(q, caches[i % nc][0], caches[i % nc][1], B, H, 1, Sk, hd, causal=True, scale=hd ** -0.5,
                                               kv_batch_rows=Smax), a.iters)
        mb = 2 * B * Sk * E * 2 / 1e6
        print(f"B={B:4d}  {us:7.1f} us  {mb:6.1f} MB  {mb / us / 1e6 * 1e6 / 1e3:6.2f} TB/s")
        del caches
    print("-- split-K GEMM, M = 32, weights rotated over > 600 MB")
    for N, K, what in ((3 * E, E, "qkv"), (E, E, "proj"), (F, E, "fc1"), (E, F, "fc2"), (50272, E, "lm_head")):
        nb = 1 if a.hot else max(2, int(6e8 / (2.0 * N * K)) + 1)
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(nb)]
        x = torch.randn(32, K, device=dev).to(bf)
        try:
            us = timed(lambda i: ops.gemm_splitk(x, ws[i % nb]), a.iters)
            ks = ops.gemm_splitk(x, ws[0]).shape[0]
        except Exception as e:
            print(f"{what:8s} N={N:6d} K={K:6d}  split-K unsupported: {e}")
            us, ks = None, 0
        us2 = timed(lambda i: ops.gemm(x, ws[i % nb]), a.iters)
        mb = N * K * 2 / 1e6
        if us is not None:
            print(f"{what:8s} N={N:6d} K={K:6d} ks={ks:2d}  split-K {us:7.1f} us {mb / us / 1e3 * 1e3 / 1e3:5.2f} TB/s   skinny {us2:7.1f} us {mb / us2:5.2f} TB/s")
        else:
            print(f"{what:8s} N={N:6d} K={K:6d}  skinny {us2:7.1f} us {mb / us2:5.2f} TB/s")
        del ws


if __name__ == "__main__":
    main()
