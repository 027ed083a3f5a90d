#!/usr/bin/env python3
"""Cached generation of a cfg5-class LM (OPT-6.7B, B = 32, 150-position few-shot prompt, 10 new tokens) with bf16 weights and with the
weights held in e4m3 (`lm_weight_format="fp8"`, eavqa_lm_block_forward_fp8): prefill / decode milliseconds (HIP events inside
greedy_decode), bytes per decode step and the fraction of 8 TB/s.

    python tools/fp8_decode_bench.py [--lm facebook/opt-6.7b] [--formats native,fp8]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eavqa_amd.data.synthetic import fewshot_batch
from eavqa_amd.models.clipcap import ClipCaptionPrefix
from eavqa_amd.models.lm import KNOWN_CONFIGS, FrozenCausalLM, LMConfig, random_init_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--lm", default="facebook/opt-6.7b")
ap.add_argument("--formats", default="native,fp8")
a = ap.parse_args()
dev, dtype = "cuda:0", torch.bfloat16
B, shots, seg, L, new, D = 32, 4, 20, 10, 10, 768
lcfg = LMConfig.from_hf_dict(KNOWN_CONFIGS[a.lm])
E, F, V, NL = lcfg.n_embd, lcfg.ffn, lcfg.vocab, lcfg.n_layer
for fmt in a.formats.split(","):
    lm = FrozenCausalLM(lcfg, random_init_state_dict(lcfg, 2021, dev), dtype, dev, weight_format=fmt)
    torch.manual_seed(2021)
    model = ClipCaptionPrefix(prefix_length=L, prefix_size=D, mapping_type="mlp", lm=lm, dtype=dtype, device=dev).eval()
    b = fewshot_batch(B, V, shots, seg, V - 1, image_size=32, device=dev)
    emb = torch.randn(B, shots + 1, D, device=dev)
    run = lambda marks=None: model.generate_fewshot(b["input_ids"], emb, b["attention_mask"], num_shots=shots, special_token_id=V - 1,
                                                     max_length=new, pad_token_id=1, eos_token_id=None, marks=marks)
    run(); torch.cuda.synchronize()
    best = None
    for _ in range(3):
        marks = [("start", None)]
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        run(marks); torch.cuda.synchronize()
        ev = dict(marks[1:])
        t = (e0.elapsed_time(ev["prefill"]), ev["prefill"].elapsed_time(ev["decode"]))
        best = t if best is None or t[1] < best[1] else best
    S0 = (shots + 1) * (1 + seg) + (L - 1) * (shots + 1)
    steps = new - 1
    wbytes = (1.0 if fmt == "fp8" else 2.0) * (NL * (4 * E * E + 2 * E * F) + E * V)
    kvb = sum(2.0 * NL * B * (S0 + t + 1) * E * 2 for t in range(steps)) / steps
    ms = best[1] / steps
    print(f"{a.lm} weights {fmt:6s}: mapper + prefill {best[0]:7.2f} ms, decode {best[1]:7.2f} ms = {ms:.3f} ms/step; {wbytes / 1e9:.2f} GB weights + "
          f"{kvb / 1e9:.2f} GB K/V per step = {(wbytes + kvb) / ms / 1e9:.2f} TB/s = {(wbytes + kvb) / ms / 1e9 / 8:.3f} of 8 TB/s", flush=True)
    del model, lm
    torch.cuda.empty_cache()
