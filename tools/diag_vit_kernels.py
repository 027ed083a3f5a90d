#!/usr/bin/env python3
"""Per-kernel device time inside ONE ViT-L/14 encode of the few-shot batch (torch.profiler kernel rows): where the time between the GEMMs goes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eavqa_amd.models.clip_vit import KNOWN_VITS, ClipVisionEncoder, random_init_vit_state_dict
from torch.profiler import profile, ProfilerActivity

f = bench.FEWSHOT
dev = "cuda:0"
vcfg = KNOWN_VITS[f["vit"]]
vit = ClipVisionEncoder(vcfg, random_init_vit_state_dict(vcfg, 2021, dev), torch.bfloat16, dev)
n = f["batch"] * (f["shots"] + 1)
px = torch.randn(n, 3, vcfg.image, vcfg.image, device=dev)
for _ in range(2):
    vit.encode_image(px)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); vit.encode_image(px); e1.record(); torch.cuda.synchronize()
print(f"encode of {n} images: {e0.elapsed_time(e1):.2f} ms")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    vit.encode_image(px); torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
agg = {}
for e in ev:
    a = agg.setdefault(e.name[:90], [0, 0.0]); a[0] += 1; a[1] += e.device_time
tot = sum(v[1] for v in agg.values())
t0 = min(e.time_range.start for e in ev); t1 = max(e.time_range.end for e in ev)
print(f"kernel time sum {tot/1e3:.2f} ms over a span of {(t1-t0)/1e3:.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"{v[1]/1e3:8.2f} ms {v[0]:5d} x {v[1]/v[0]:8.1f} us  {k}")
