#!/usr/bin/env python3
"""Which torch (ATen) kernels run inside one training step of a bench workload, and from which Python line: the hot path is supposed to
launch none besides allocation / tiny index bookkeeping.  torch.profiler with stacks, one step after warm-up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eavqa_amd import ops
from eavqa_amd.trainers.data_parallel import GradSync

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dtype_name = sys.argv[2] if len(sys.argv) > 2 else "bf16"
w, vcfg, lcfg, vit, model, opt, batch, pad = bench.build_workload(name, torch.bfloat16, "cuda:0", 0, None, "fp8" if dtype_name == "fp8" else "native")
sync = GradSync(model.clip_project.flat.grad, 1)
st = bench.Stepper(vit, model, opt, batch, pad, sync, overlap_vit=False)
for _ in range(3):
    st.step()
st.flush(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    st.step(); st.flush(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f"{e.key:28s} calls {e.count:4d}  device {e.device_time_total:9.1f} us   shapes {str(e.input_shapes)[:70]}")
    for fr in e.stack[:6]:
        if "eavqa" in fr or "bench" in fr or "explicit" in fr:
            print("      ", fr)
