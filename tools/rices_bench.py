#!/usr/bin/env python3
"""RICES retrieval at the reference's size: 443 757 train question embeddings x 768, k = 2048 (get_question_knn.py:64-76),
timed per tile of 1024 queries."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eavqa_amd import ops

Nd, D, Q, k = 443757, 768, 1024, 2048
db = ops.l2_normalize_rows_(torch.randn(Nd, D, device="cuda"))
q = ops.l2_normalize_rows_(torch.randn(Q, D, device="cuda"))
for _ in range(2):
    s = ops.gemm(q, db, out_f32=True)
    v, i = ops.topk_rows(s, k)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record(); s = ops.gemm(q, db, out_f32=True); e[1].record(); v, i = ops.topk_rows(s, k); e[2].record()
torch.cuda.synchronize()
g, t = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
print(f"scores GEMM [{Q} x {Nd} x {D}] fp32: {g:.1f} ms ({2.0 * Q * Nd * D / g / 1e9:.1f} TFLOP/s);  top-{k}: {t:.1f} ms "
      f"({Q * Nd * 4 / t / 1e9:.2f} TB/s of scores per pass-equivalent);  214 354 val queries -> {(g + t) * 214354 / Q / 1e3:.1f} s")
