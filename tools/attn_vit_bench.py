#!/usr/bin/env python3
"""CLIP-tower attention in isolation: the K / V-resident forward against the streamed-tile forward, interleaved rounds in one process
(ViT-L/14: 257 tokens x 16 heads; ViT-L/14@336: 577; ViT-B/32: 50 x 12 heads), on the fused [rows, 3W] qkv layout of the tower."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eavqa_amd import ops


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, B, H, N in (("ViT-L/14 x160", 160, 16, 257), ("ViT-L/14 x64", 64, 16, 257), ("ViT-L/14@336 x32", 32, 16, 577), ("ViT-B/32 x64", 64, 12, 50)):
    hd = 64
    E = H * hd
    qkv = torch.randn(B * N, 3 * E, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    res = {}
    for rnd in range(3):
        for label, path in (("resident", 8), ("streamed", 4)):
            ops.KernelSelect.attention = path
            res.setdefault(label, []).append(timed(lambda: ops.attention_fwd(q, k, v, B, H, N, N, hd, causal=False, scale=hd ** -0.5)))
    ops.KernelSelect.attention = 0
    flop = 4.0 * B * H * N * N * hd
    byts = B * N * E * 2 * 4.0
    for label, ts in res.items():
        t = min(ts)
        print(f"{name:18s} {label:9s} {t:8.1f} us (median {sorted(ts)[1]:8.1f})  {flop / t / 1e6:7.1f} TFLOP/s  {byts / t / 1e3:7.1f} GB/s of q+k+v+o once")
