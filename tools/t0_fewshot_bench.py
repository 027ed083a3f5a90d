#!/usr/bin/env python3
"""Few-shot VQA generate with the reference's headline model shape: CLIP ViT-L/14 -> MLP mapper -> T0_3B (T5 v1.1 XL encoder-decoder,
random-init weights), 32 questions x (4 shots + query), 20 text tokens per segment, prefix 10, max_length 10 (`VCT0Prefix.generate`,
reference vct0.py:396-491 driven as in few_shot_vqa_executor.py:195-205).  Prints questions/s and the phase times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eavqa_amd.data.synthetic import fewshot_batch
from eavqa_amd.models.clip_vit import KNOWN_VITS, ClipVisionEncoder, random_init_vit_state_dict
from eavqa_amd.models.vct0 import VCT0Prefix

dev, dtype = "cuda:0", torch.bfloat16
B, shots, seg, L, new = 32, 4, 20, 10, 10
vcfg = KNOWN_VITS["ViT-L/14"]
vit = ClipVisionEncoder(vcfg, random_init_vit_state_dict(vcfg, 2021, dev), dtype, dev)
torch.manual_seed(2021)
model = VCT0Prefix(prefix_length=L, prefix_size=vcfg.proj, mapping_type="mlp", model_version="bigscience/T0_3B", dtype=dtype, device=dev)
V = model.lm.cfg.vocab
b = fewshot_batch(B, V, shots, seg, 32099, image_size=vcfg.image, device=dev)
n_img = shots + 1
px = b["pixel_values"].reshape(B * n_img, *b["pixel_values"].shape[2:])

def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e

def run(marks=None):
    if marks is not None: marks.append(ev())
    emb = vit.encode_image(px).view(B, n_img, -1)
    if marks is not None: marks.append(ev())
    out = model.generate(prefix=emb, question_tokens=b["input_ids"], question_mask=b["attention_mask"], num_shots=shots, max_length=new)
    if marks is not None: marks.append(ev())
    return out

out = run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): out = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
m = []; run(m); torch.cuda.synchronize()
print(f"T0_3B few-shot: {B / dt:.1f} questions/s ({dt * 1e3:.1f} ms per batch of {B}); ViT-L encode {m[0].elapsed_time(m[1]):.1f} ms, "
      f"mapper + T5 encoder + {new - 1} greedy decoder steps {m[1].elapsed_time(m[2]):.1f} ms; output {tuple(out.shape)}")

# ---- where the T5 time goes: encoder / cross K,V / one decoder pass of t tokens, host enqueue time vs total
lm = model.lm
emb = vit.encode_image(px).view(B, n_img, -1)
rows = model._project(emb)
enc, mask, S = model._encode_interleaved(b["input_ids"], b["attention_mask"], rows, n_img, 32099)
torch.cuda.synchronize()
def hd(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    h = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return h * 1e3, (time.perf_counter() - t0) / n * 1e3
h, d = hd(lambda: model._encode_interleaved(b["input_ids"], b["attention_mask"], rows, n_img, 32099))
print(f"T5 encoder over {S} positions x {B}: host {h:.1f} ms, total {d:.1f} ms")
h, d = hd(lambda: lm.cross_kv(enc))
print(f"cross K/V of 24 layers: host {h:.1f} ms, total {d:.1f} ms")
kv = lm.cross_kv(enc)
for t in (1, 5, 9):
    ids = torch.zeros((B, t), dtype=torch.int64, device=dev)
    def step():
        y = lm.embed(ids)
        hid, _ = lm.decode(y, enc, mask, B, t, S, kv=kv)
        return lm.logits(hid.view(B, t, -1)[:, -1].contiguous())
    h, d = hd(step)
    print(f"decoder pass over {t} token(s): host {h:.1f} ms, total {d:.1f} ms")

# ---- the cached greedy step alone (eavqa_t5_decoder_step), the library's route against the round-3 call sequence
from eavqa_amd.models.t5 import _StepDriver
t_max = new
cache = [(torch.empty((B * t_max, lm.cfg.inner), device=dev, dtype=dtype), torch.empty((B * t_max, lm.cfg.inner), device=dev, dtype=dtype)) for _ in lm.dec]
driver = _StepDriver(lm, cache, kv, B, t_max)
rel = lm.rel_table(True, t_max)
y0 = lm.embed(torch.zeros((B, 1), dtype=torch.int64, device=dev)[:, 0].contiguous())
for route in (0, 1):
    lm.step_route = route
    for t in range(1, t_max):
        driver.step(y0.clone(), mask, t, S, rel)
    torch.cuda.synchronize()
    e0, e1 = ev(), None
    for rep in range(3):
        for t in range(1, t_max):
            driver.step(y0.clone(), mask, t, S, rel)
    e1 = ev(); torch.cuda.synchronize()
    w = sum(p.numel() * 2 for blk in lm.dec for p in (blk.w_qkv, blk.w_o, blk.w_q_ca, blk.w_o_ca, blk.w_i, blk.w_o_ff))
    kvb = len(lm.dec) * B * S * 2 * lm.cfg.inner * 2
    ms = e0.elapsed_time(e1) / (3 * (t_max - 1))
    print(f"cached decoder step, route {route} ({'split-K' if route == 0 else 'round-3 sequence'}): {ms:.3f} ms per step; weights {w / 1e9:.2f} GB + cross K/V "
          f"{kvb / 1e9:.2f} GB per step = {(w + kvb) / ms / 1e9:.2f} TB/s = {(w + kvb) / ms / 1e9 / 8:.3f} of 8 TB/s")
lm.step_route = 0
