#!/usr/bin/env python3
"""How host-bound is a training step?  Host enqueue time vs GPU completion time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eavqa_amd.trainers.data_parallel import GradSync

w, vcfg, lcfg, vit, model, opt, batch, pad = bench.build_workload("cfg2", torch.bfloat16, "cuda:0", 0)
st = bench.Stepper(vit, model, opt, batch, pad, GradSync(model.clip_project.flat.grad, 1))
for _ in range(3):
    st.step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    st.step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {1e3*t_host/n:.2f} ms/step, completion {1e3*t_all/n:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): st.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
