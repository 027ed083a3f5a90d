#!/usr/bin/env python3
"""Reference point only (never on the product path): what torch.matmul - i.e. the vendor GEMM library behind PyTorch-ROCm - reaches on
the GEMM shapes of this repo's workloads, next to eavqa_gemm on the same operands.  Answers "is the shape or the kernel the limit?".

    python tools/vendor_gemm_ref.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eavqa_amd import ops

SHAPES = [(1864, 1280, 1280, "cfg2 out-proj"), (1864, 3840, 1280, "cfg2 qkv"), (1864, 5120, 1280, "cfg2 fc1"), (1864, 1280, 5120, "cfg2 fc2"),
          (4800, 7680, 2560, "prefill qkv"), (4800, 2560, 2560, "prefill proj"), (4800, 10240, 2560, "prefill fc1"), (4800, 2560, 10240, "prefill fc2"),
          (41120, 3072, 1024, "vitL qkv"), (41120, 1024, 1024, "vitL proj"), (41120, 4096, 1024, "vitL fc1"), (41120, 1024, 4096, "vitL fc2"),
          (2048, 4096, 4096, "opt67 proj"), (2048, 16384, 4096, "opt67 fc1"), (4096, 4096, 4096, "square 4k"), (8192, 8192, 8192, "square 8k")]


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev = "cuda"
    print(f"{'shape':16s} {'M':>6} {'N':>6} {'K':>6} | torch.matmul us  TF/s | eavqa_gemm us  TF/s")
    for M, N, K, what in SHAPES:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        iters = max(5, min(50, int(2e12 / (2.0 * M * N * K)) + 5))
        bt = b.t()
        t_vendor = timed(lambda: torch.matmul(a, bt, out=out), iters)
        t_ours = timed(lambda: ops.gemm(a, b, out=out), iters)
        fl = 2.0 * M * N * K
        print(f"{what:16s} {M:6d} {N:6d} {K:6d} | {t_vendor:12.1f} {fl / t_vendor / 1e6:6.0f} | {t_ours:10.1f} {fl / t_ours / 1e6:6.0f}", flush=True)


if __name__ == "__main__":
    main()
