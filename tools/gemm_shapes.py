#!/usr/bin/env python3
"""Every eavqa_gemm call of one training step, by shape, timed IN SITU (events around each launch, inside the real step: cold
weights, the real predecessor kernels), against the same shape timed alone in a loop.

    python tools/gemm_shapes.py [--workload cfg2] [--steps 4]
"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eavqa_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--fp8", action="store_true", help="frozen LM in fp8 (cfg5): only the bf16 GEMMs (mapper, CLIP tower) are listed")
    ap.add_argument("--vit", default=None, help="time the GEMMs of a CLIP encode instead (e.g. ViT-L/14), --images per call")
    ap.add_argument("--images", type=int, default=160)
    a = ap.parse_args()
    dev = "cuda:0"
    if a.vit:
        from eavqa_amd.models.clip_vit import KNOWN_VITS, ClipVisionEncoder, random_init_vit_state_dict
        vcfg = KNOWN_VITS[a.vit]
        vit = ClipVisionEncoder(vcfg, random_init_vit_state_dict(vcfg, 2021, dev), torch.bfloat16, dev)
        px = torch.randn(a.images, 3, vcfg.image, vcfg.image, device=dev)

        class _S:
            def step(self):
                vit.encode_image(px)

            def flush(self):
                pass
        stepper = _S()
        return analyse(stepper, a, dev)
    w, vcfg, lcfg, vit, model, opt, batch, pad = bench.build_workload(a.workload, torch.bfloat16, dev, 0, weight_format="fp8" if a.fp8 else "native")
    from eavqa_amd.trainers.data_parallel import GradSync
    stepper = bench.Stepper(vit, model, opt, batch, pad, GradSync(model.clip_project.flat.grad, 1), overlap_vit=False)
    return analyse(stepper, a, dev)


def analyse(stepper, a, dev):
    for _ in range(3):
        stepper.step()
    stepper.flush()
    torch.cuda.synchronize()

    records = []
    real = ops.call

    def spy(name, *args):
        if name not in ("eavqa_gemm", "eavqa_gemm_ex"):
            return real(name, *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real(name, *args)
        e1.record()
        dt, a_kc, b_kc, M, N, K = args[:6]
        records.append(((dt, a_kc, b_kc, M, N, K, bool(args[14]), args[15], bool(args[19])), e0, e1))

    ops.call = spy
    for _ in range(a.steps):
        stepper.step()
    stepper.flush()
    torch.cuda.synchronize()
    ops.call = real

    agg = collections.OrderedDict()
    for key, e0, e1 in records:
        agg.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)
    print(f"{'dt':>2} {'a_kc':>4} {'b_kc':>4} {'M':>6} {'N':>6} {'K':>6} bias act res | calls/step   in-situ us   TF/s    alone us   TF/s  cold-B us   TF/s   ms/step")
    total = 0.0
    rows = []
    for key, ts in agg.items():
        dt, a_kc, b_kc, M, N, K, bias, act, res = key
        ts.sort()
        med = ts[len(ts) // 2]
        # the same shape alone, back to back
        dtype = torch.bfloat16 if dt == 1 else torch.float32
        A = torch.randn((M, K) if a_kc else (K, M), device=dev).to(dtype)
        B = torch.randn((N, K) if b_kc else (K, N), device=dev).to(dtype)
        for _ in range(3):
            ops.gemm(A, B, a_kc=bool(a_kc), b_kc=bool(b_kc))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(A, B, a_kc=bool(a_kc), b_kc=bool(b_kc))
        e1.record()
        torch.cuda.synchronize()
        alone = e0.elapsed_time(e1) * 1e3 / 20
        # ... and alone with COLD weights: rotate over enough copies of B to overflow the 256 MiB Infinity Cache
        nb = min(64, int(4e8 / (2.0 * N * K)) + 2)
        Bs = [B] + [B.clone() for _ in range(nb - 1)] if nb * N * K * 2 < 6e9 else [B]
        for i in range(3):
            ops.gemm(A, Bs[i % len(Bs)], a_kc=bool(a_kc), b_kc=bool(b_kc))
        e0.record()
        for i in range(2 * len(Bs)):
            ops.gemm(A, Bs[i % len(Bs)], a_kc=bool(a_kc), b_kc=bool(b_kc))
        e1.record()
        torch.cuda.synchronize()
        cold = e0.elapsed_time(e1) * 1e3 / (2 * len(Bs))
        del Bs
        fl = 2.0 * M * N * K
        per_step = len(ts) / a.steps
        ms = sum(ts) / a.steps / 1e3
        total += ms
        rows.append((ms, f"{dt:>2} {a_kc:>4} {b_kc:>4} {M:>6} {N:>6} {K:>6} {int(bias):>4} {act:>3} {int(res):>3} | {per_step:10.1f} {med:12.1f} {fl / med / 1e6:6.0f} {alone:11.1f} {fl / alone / 1e6:6.0f} {cold:10.1f} {fl / cold / 1e6:6.0f} {ms:9.3f}"))
    for _, line in sorted(rows, reverse=True):
        print(line)
    print(f"total GEMM time per step (events, includes the event gaps): {total:.2f} ms")


if __name__ == "__main__":
    main()
