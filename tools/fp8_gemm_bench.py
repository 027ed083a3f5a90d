#!/usr/bin/env python3
"""eavqa_gemm_fp8 on the cfg5 (OPT-6.7B, M = 2048) shapes, every tile of FP8_SHAPES forced in turn (tile 0 = the cost model's pick)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eavqa_amd import ops
SHAPES = [(2048, 12288, 4096, "qkv"), (2048, 4096, 4096, "proj"), (2048, 16384, 4096, "fc1 / d_fc2"), (2048, 4096, 16384, "fc2 / d_fc1"),
          (2048, 4096, 12288, "d_qkv"), (1024, 50304, 4096, "lm_head"), (1024, 4096, 50304, "d_head")]
NAMES = ["auto", "128x80", "256x128", "256x160", "128x128", "128x256"]     # round 3 also tried 256x192 / 256x256 (loader / consumer, 12 waves):
# 168 VGPRs, 68 / 276 B of scratch per lane, 1 062 / 333 TFLOP/s on the QKV shape against 1 903 for 256x128 - not kept
dev = "cuda"
for M, N, K, what in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    aq, asc = ops.quantize_rows_fp8(a)
    wq, wsc = ops.quantize_rows_fp8(w)
    bq = wq
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = None
    row = []
    for tile in range(len(NAMES)):
        try:
            for _ in range(2):
                ops.gemm_fp8(aq, asc, bq, 1.0, out=out, tile=tile)
            if ref is None:
                ref = out.float().clone()
            err = (out.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                ops.gemm_fp8(aq, asc, bq, 1.0, out=out, tile=tile)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            row.append(f"{NAMES[tile]} {us:6.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF" + ("" if err < 2e-2 else f" ERR {err:.2e}"))
        except Exception as e:
            row.append(f"{NAMES[tile]} failed: {e}")
    print(f"{what:12s} M={M} N={N} K={K}: " + " | ".join(row), flush=True)
