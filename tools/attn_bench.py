#!/usr/bin/env python3
"""Training-shape attention kernels in isolation (GPT-2-large heads: H = 20, hd = 64, S = 42 padded / packed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eavqa_amd import ops

def timed(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

H, hd = 20, 64
E = H * hd
for B in (4, 16, 64, 256):
    S = 42
    M = B * S
    qkv = torch.randn(M, 3 * E, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    do = torch.randn(M, E, device="cuda").to(torch.bfloat16)
    o, lse = ops.attention_fwd(q, k, v, B, H, S, S, hd, causal=True, scale=hd ** -0.5, save_lse=True)
    tf = timed(lambda: ops.attention_fwd(q, k, v, B, H, S, S, hd, causal=True, scale=hd ** -0.5, save_lse=True))
    tb = timed(lambda: ops.attention_bwd(q, k, v, o, do, lse, B, H, S, S, hd, causal=True, scale=hd ** -0.5))
    mb_f, mb_b = M * E * 2 * 4 / 1e6, M * E * 2 * 8 / 1e6
    print(f"B={B:4d} rows={M:6d}  fwd {tf:6.1f} us ({mb_f / tf * 1e-6 * 1e6 / 1e3:5.2f} GB/ms)  bwd {tb:6.1f} us ({mb_b / tb:5.2f} MB/us)")
