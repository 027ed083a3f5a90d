#!/usr/bin/env python3
"""Phase timing of the few-shot generate path (ViT encode / mapper+prefill / decode steps) with host vs device time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eavqa_amd import ops
from eavqa_amd.data.synthetic import fewshot_batch
from eavqa_amd.models.clip_vit import KNOWN_VITS, ClipVisionEncoder, random_init_vit_state_dict
from eavqa_amd.models.clipcap import ClipCaptionPrefix
from eavqa_amd.models.lm import KNOWN_CONFIGS, FrozenCausalLM, LMConfig, random_init_state_dict

f = bench.FEWSHOT
dev, dtype = "cuda:0", torch.bfloat16
vcfg = KNOWN_VITS[f["vit"]]; lcfg = LMConfig.from_hf_dict(KNOWN_CONFIGS[f["lm"]])
vit = ClipVisionEncoder(vcfg, random_init_vit_state_dict(vcfg, 2021, dev), dtype, dev)
lm = FrozenCausalLM(lcfg, random_init_state_dict(lcfg, 2021, dev), dtype, dev)
model = ClipCaptionPrefix(prefix_length=10, prefix_size=vcfg.proj, mapping_type="mlp", lm=lm, dtype=dtype, device=dev).eval()
b = fewshot_batch(f["batch"], lcfg.vocab, f["shots"], f["seg_len"], lcfg.vocab - 1, image_size=vcfg.image, device=dev)
B, n_img = f["batch"], f["shots"] + 1
px = b["pixel_values"].reshape(B * n_img, *b["pixel_values"].shape[2:])

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    th = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return r, th * 1e3, (time.perf_counter() - t0) / n * 1e3

emb, h, d = timed(lambda: vit.encode_image(px))
print(f"ViT-L/14 encode of {B*n_img} images: host {h:.1f} ms, total {d:.1f} ms")
emb = emb.view(B, n_img, -1)
for new in (1, 10):
    _, h, d = timed(lambda: model.generate_fewshot(b["input_ids"], emb, b["attention_mask"], num_shots=f["shots"],
                    special_token_id=lcfg.vocab - 1, max_length=new, pad_token_id=1, eos_token_id=None))
    print(f"mapper + prefill + {new} token(s): host {h:.1f} ms, total {d:.1f} ms")
