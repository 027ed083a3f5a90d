#!/usr/bin/env python3
"""A / B of the decode-step GEMM kernels at M = 32 on cold weights, meant to run under `rocprofv3 --kernel-trace` (kernel times from
tools/rocpd_stats.py; the event times printed include host launch overhead): split-K with non-temporal / plain weight loads, the direct
kernel in its natural layout with and without the rotated K walk.   python tools/decode_gemm_ab.py [--model opt2.7b|t0]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eavqa_amd import ops

def timed(fn, n=12):
    fn(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

ap = argparse.ArgumentParser(); ap.add_argument("--model", default="opt2.7b"); a = ap.parse_args()
dev, bf, M = "cuda", torch.bfloat16, 32
shapes = {"opt2.7b": (("qkv", 7680, 2560), ("out", 2560, 2560), ("fc1", 10240, 2560), ("fc2", 2560, 10240)),
          "t0": (("qkv", 6144, 2048), ("o", 2048, 2048), ("wi", 10240, 2048), ("wo_ff", 2048, 5120))}[a.model]
for what, N, K in shapes:
    nb = max(2, int(7e8 / (2.0 * N * K)) + 1)
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(nb)]
    x = torch.randn(M, K, device=dev).to(bf)
    out = torch.empty((M, N), device=dev, dtype=bf)
    part = ops.gemm_splitk(x, ws[0])
    line = f"{what:5s} N={N:6d} K={K:6d} ks={part.shape[0]:2d} |"
    for sel, tag in ((0, "sk nt"), (0x10000, "sk plain")):
        us = timed(lambda i: ops.gemm_splitk(x, ws[i % nb], unroll=sel or 0, out=part) if sel else ops.gemm_splitk(x, ws[i % nb], out=part))
        line += f" {tag} {us:6.1f} |"
    for sel, tag in ((0, "direct"), (0x80, "direct rot"), (0x10, "direct plain"), (0x20, "no A"), (0x40, "no B"), (0x60, "neither")):
        us = timed(lambda i: ops.gemm_decode(x, ws[i % nb], [out], sel=sel))
        line += f" {tag} {us:6.1f} |"
    print(line, flush=True)
    del ws
