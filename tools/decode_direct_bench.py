#!/usr/bin/env python3
"""The decode-step GEMMs of OPT-2.7B / OPT-6.7B / T0-3B at M = 32, weights rotated so that every launch streams cold HBM:
split-K + its consumer pass (csrc/decode.hip) against the direct kernel (csrc/decode_direct.hip) in its variants.

    python tools/decode_direct_bench.py [--iters 30] [--model opt2.7b|opt6.7b|t0]
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
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--model", default="opt2.7b")
    ap.add_argument("--M", type=int, default=32)
    a = ap.parse_args()
    dev, bf = "cuda", torch.bfloat16
    M = a.M
    if a.model == "opt2.7b":
        E, F = 2560, 10240
        shapes = [("qkv", 3 * E, E, "layer", False), ("out", E, E, None, False), ("fc1", F, E, "layer", False), ("fc2", E, F, None, False)]
    elif a.model == "opt6.7b":
        E, F = 4096, 16384
        shapes = [("qkv", 3 * E, E, "layer", False), ("out", E, E, None, False), ("fc1", F, E, "layer", False), ("fc2", E, F, None, False)]
    else:
        E, F = 2048, 5120
        shapes = [("qkv", 3 * E, E, "rms", False), ("o", E, E, None, False), ("wi", F, E, "rms", True), ("wo_ff", E, F, None, False)]
    for what, N, K, norm, gated in shapes:
        rows = 2 * N if gated else N
        nb = max(2, int(7e8 / (2.0 * rows * K)) + 1)
        ws = [(torch.randn(rows, K, device=dev) * 0.02).to(bf) for _ in range(nb)]
        xb = torch.randn(M, K, device=dev).to(bf)
        xf = torch.randn(M, K, device=dev)
        gamma, beta = torch.ones(K, device=dev), torch.zeros(K, device=dev)
        bias = torch.zeros(rows, device=dev)
        mb = rows * K * 2 / 1e6
        line = f"{what:6s} N={N:6d} K={K:6d} {mb:6.1f} MB |"
        # split-K + finish (what the round-3 step runs)
        if not gated:
            try:
                part = ops.gemm_splitk(xb, ws[0])
                outb = torch.empty((M, N), device=dev, dtype=bf)
                us = timed(lambda i: (ops.gemm_splitk(xb, ws[i % nb], out=part), ops.splitk_finish(part, [outb], bias=bias)), a.iters)
                us1 = timed(lambda i: ops.gemm_splitk(xb, ws[i % nb], out=part), a.iters)
                line += f" split-K ks={part.shape[0]:2d} {us1:6.1f} us (+finish {us:6.1f}) |"
            except Exception as e:
                line += f" split-K n/a ({e}) |"
        # direct, bf16 A
        out = torch.empty((M, N), device=dev, dtype=bf)
        for sel, tag in ((0, "nt"), (0x10, "plain")):
            us = timed(lambda i: ops.gemm_decode(xb, ws[i % nb], [out], gated=gated, bias=None if gated else bias, act="relu", sel=sel), a.iters)
            line += f" direct[{tag}] {us:6.1f} us {mb / us:5.2f} TB/s |"
        for nf in ((2, 4) if gated else (1, 2, 3, 4)):
            us = timed(lambda i: ops.gemm_decode(xb, ws[i % nb], [out], gated=gated, act="relu", sel=nf), a.iters)
            line += f" nf{nf} {us:6.1f}"
        us = timed(lambda i: ops.gemm_decode(xb, ws[i % nb], [out], gated=gated, act="relu", sel=0x200), a.iters)
        line += f" | rows/2 {us:6.1f}"
        if norm:
            # a producer's statistics for the fp32 stream
            wsq = (torch.randn(K, K, device=dev) * 0.02).to(bf)
            x1 = torch.empty((M, K), device=dev, dtype=torch.float32)
            st = ops.gemm_decode(xb, wsq, [x1], residual=xf, want_stats=True)
            cols = ops.gemm_decode_cols(M, K, K)
            for nf in ((0,) if gated else (0, 1, 2, 3)):
                try:
                    us = timed(lambda i: ops.gemm_decode(x1, ws[i % nb], [out], norm=norm, gamma=gamma, beta=beta if norm == "layer" else None,
                                                         stats_in=st, stats_in_cols=cols, gated=gated, act="relu", sel=nf), a.iters)
                    line += f" | {norm}-on-load nf{nf} {us:6.1f} us {mb / us:5.2f} TB/s"
                except Exception as e:
                    line += f" | {norm}-on-load nf{nf} n/a"
        print(line, flush=True)
        del ws


if __name__ == "__main__":
    main()
