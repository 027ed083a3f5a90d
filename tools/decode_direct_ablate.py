#!/usr/bin/env python3
"""Where the direct decode GEMM's time goes: the same launch with the A loads, the B loads, or both dropped by an empty buffer descriptor
(eavqa_gemm_decode_ex sel bits 5 / 6).  Run under rocprofv3 --kernel-trace and read tools/rocpd_stats.py, or take the event times printed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eavqa_amd import ops

def timed(fn, n=20):
    fn(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

dev, bf = "cuda", torch.bfloat16
M = 32
for what, N, K in (("qkv", 7680, 2560), ("out", 2560, 2560), ("fc1", 10240, 2560), ("fc2", 2560, 10240)):
    nb = max(2, int(7e8 / (2.0 * N * K)) + 1)
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(nb)]
    x = torch.randn(M, K, device=dev).to(bf)
    out = torch.empty((M, N), device=dev, dtype=bf)
    line = f"{what:4s} N={N:6d} K={K:6d} |"
    for sel, tag in ((0, "full"), (0x80, "rotated"), (0x20, "no A"), (0x40, "no B"), (0x60, "neither")):
        us = timed(lambda i: ops.gemm_decode(x, ws[i % nb], [out], sel=sel))
        line += f" {tag} {us:6.1f} us |"
    print(line, flush=True)
    del ws
