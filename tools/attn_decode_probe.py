#!/usr/bin/env python3
"""Decode attention alone at the few-shot shape (B = 32, 32 heads x 80, ~320 cached keys), KV rotated over 8 copies (cold HBM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eavqa_amd import ops
dev, bf = "cuda", torch.bfloat16
E, H, hd, B = 2560, 32, 80, 32
Smax, Sk = 330, int(sys.argv[1]) if len(sys.argv) > 1 else 320
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 24
caches = [(torch.randn(B * Smax, E, device=dev).to(bf), torch.randn(B * Smax, E, device=dev).to(bf)) for _ in range(8)]
q = torch.randn(B, E, device=dev).to(bf)
f = lambda i: ops.attention_fwd(q, caches[i % 8][0], caches[i % 8][1], B, H, 1, Sk, hd, causal=True, scale=hd ** -0.5, kv_batch_rows=Smax)
f(0); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(iters): f(i)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
mb = 2 * B * Sk * E * 2 / 1e6
print(f"B={B} Sk={Sk}: {us:.1f} us, {mb:.1f} MB of K+V = {mb / us:.2f} TB/s", flush=True)
