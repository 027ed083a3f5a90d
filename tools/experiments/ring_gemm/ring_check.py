#!/usr/bin/env python3
"""Parity of the 256 x 256 ring kernel (knob big_mode 3) against torch on ragged and full shapes, with the epilogue variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eavqa_amd import ops
ok = True
for deep in (0, 1, 2):
    ops.KernelSelect.gemm = (3 << 14) | (deep << 16)
    for (M, N, K) in [(256, 256, 64), (256, 256, 32 * 7), (300, 520, 1024), (1000, 1280, 64 * 3), (41120, 1024, 1024), (4100, 4096, 1024), (77, 300, 128)]:
        torch.manual_seed(M + N + K)
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        b = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        ref = a.float() @ b.float().T + bias
        out = ops.gemm(a, b, bias=bias)
        e1 = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        out2 = ops.gemm(a, b, bias=bias, act="quick_gelu")
        r2 = ref * torch.sigmoid(1.702 * ref)
        e2 = (out2.float() - r2).abs().max().item() / r2.abs().max().item()
        res = torch.randn(M, N, device="cuda")
        out3 = ops.gemm(a, b, bias=bias, residual=res.clone())
        e3 = (out3.float() - (ref + res)).abs().max().item() / ref.abs().max().item()
        good = max(e1, e2, e3) < 6e-3
        ok &= good
        print(f"deep={deep} M={M} N={N} K={K}: rel err plain {e1:.2e} quick {e2:.2e} residual {e3:.2e} {'ok' if good else 'FAIL'}", flush=True)
ops.KernelSelect.gemm = 0
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
