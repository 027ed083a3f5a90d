"""Diagnostic: per-parameter gradient error of the cfg3 real-size step (fp32) against the oracle, packed and padded."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import importlib.util
spec = importlib.util.spec_from_file_location("tf", os.path.join(ROOT, "tests", "test_fullsize_gpu.py")); tf = importlib.util.module_from_spec(spec); spec.loader.exec_module(tf)
from eavqa_amd.models.clip_vit import KNOWN_VITS, random_init_vit_state_dict
name, vit_name, lm_name, B = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
L = 10
cfg, sd, b = tf._train_case(vit_name, lm_name, B, seed=31)
if len(sys.argv) > 5:
    cfg.n_layer = int(sys.argv[5])
vcfg = KNOWN_VITS[vit_name]
vsd = tf._perturb(random_init_vit_state_dict(vcfg, 2021, "cpu"), 5)
runs = {}
for pack in (True, False):
    runs[pack] = tf._hip_train(cfg, sd, vit_name, vsd, torch.float32, b, L, pack=pack)
emb, loss, logits, grads = tf._oracle_train(cfg, sd, vcfg, vsd, runs[True]["mapper"], b, L)
for pack, r in runs.items():
    print(f"pack={pack} loss {r['loss']:.6f} oracle {loss.item():.6f}")
    for k, g in grads.items():
        d = (r["grads"][k] - g).abs()
        i = int(d.argmax())
        print(f"   {k:16s} max|d| {d.max().item():.3e}  max|g| {g.abs().max().item():.3e}  rel {d.max().item()/g.abs().max().item():.3e}  at {i} (hip {r['grads'][k].flatten()[i].item():.4e} ref {g.flatten()[i].item():.4e})  mean|d|/mean|g| {d.mean().item()/g.abs().mean().item():.3e}")
d = (runs[True]["grads"]["model.2.weight"] - runs[False]["grads"]["model.2.weight"]).abs().max().item()
print("packed vs padded dW2 max|d|", d)
