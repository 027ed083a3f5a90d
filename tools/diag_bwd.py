"""Diagnostic: mapper-gradient error of the fp32 HIP path against the oracle for one-layer LMs of various widths / heads /
activations (isolates which dimension of the real-size OPT backward is off)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import oracle
from eavqa_amd.models.clipcap import ClipCaptionPrefix
from eavqa_amd.models.lm import FrozenCausalLM, LMConfig, random_init_state_dict

def run(arch, E, H, F, act, n_layer=1, V=512, B=2, T=32, L=10, D=64, dtype=torch.float32, perturb=True):
    cfg = LMConfig(arch, n_layer, H, E, F, V, 128, 1e-5, act, V - 1, None if arch == "gpt2" else 1)
    sd = random_init_state_dict(cfg, 3, "cpu")
    if perturb:
        g = torch.Generator().manual_seed(11)
        for k in sorted(sd):
            if k.endswith(".bias") or "ln_" in k or "layer_norm" in k:
                sd[k] = sd[k] + 0.05 * torch.randn(sd[k].shape, generator=g)
    lm = FrozenCausalLM(cfg, sd, dtype, "cuda")
    torch.manual_seed(1)
    model = ClipCaptionPrefix(prefix_length=L, prefix_size=D, mapping_type="mlp", lm=lm, dtype=dtype, device="cuda").train()
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(8, T + 1, (B,), generator=g); lens[0] = T
    ids = torch.randint(2, V - 2, (B, T), generator=g)
    mask = (torch.arange(T)[None] < lens[:, None]).long()
    pad = V - 1
    ids = ids * mask + pad * (1 - mask)
    labels = oracle.label_mask_cc(ids, pad)
    prefix = torch.randn(B, D, generator=g)
    out = model(question_tokens=ids, prefix=prefix, question_mask=mask, labels=labels)
    out.loss.backward()
    mapper = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in model.clip_project.state_dict().items()}
    ocfg = dict(arch=arch, n_layer=n_layer, n_head=H, act=act)
    loss, logits = oracle.clipcap_forward(sd, ocfg, mapper, dict(prefix_length=L, mapping_type="mlp"), ids, prefix, mask, labels)
    loss.backward()
    # the same oracle in float64: how far are the two fp32 implementations from the exact gradient?
    sd64 = {k: v.double() for k, v in sd.items()}
    m64 = {k: v.detach().double().clone().requires_grad_(True) for k, v in mapper.items()}
    l64, _ = oracle.clipcap_forward(sd64, ocfg, m64, dict(prefix_length=L, mapping_type="mlp"), ids, prefix.double(), mask, labels)
    l64.backward()
    errs, errs_hip64, errs_o64 = [], [], []
    for k, p in model.clip_project.named_parameters():
        w = mapper[k].grad
        errs.append((p.grad.float().cpu() - w).abs().max().item() / w.abs().max().item())
        e = m64[k].grad
        errs_hip64.append((p.grad.double().cpu() - e).abs().max().item() / e.abs().max().item())
        errs_o64.append((w.double() - e).abs().max().item() / e.abs().max().item())
    print(f"      vs float64 oracle: HIP {max(errs_hip64):.3e}   fp32 oracle {max(errs_o64):.3e}")
    att = torch.cat([torch.ones(B, L, dtype=torch.bool), mask.bool()], 1)
    print(f"{arch:5s} E={E:5d} H={H:3d} hd={E//H:4d} F={F:6d} act={act:9s} layers={n_layer} perturb={int(perturb)}: |d loss| {abs(out.loss.item()-loss.item()):.2e} "
          f"max|d logits| {(out.logits.float().cpu()-logits.detach())[att].abs().max().item():.2e}  max rel d grad {max(errs):.3e}", flush=True)

import sys
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "hd80"):
    run("opt", 640, 8, 2560, "relu")                  # hd 80, small
    run("gpt2", 640, 8, 2560, "gelu_new")
    run("opt", 768, 8, 3072, "relu")                  # hd 96
    run("opt", 640, 8, 2560, "relu", dtype=torch.bfloat16)
    run("opt", 640, 4, 2560, "relu")                  # hd 160
if which in ("all", "cfg3"):
    run("opt", 2048, 32, 8192, "relu", n_layer=4, V=50272, D=768)
    run("opt", 2048, 32, 8192, "relu", n_layer=4, V=50272, D=768, B=2, T=32, L=10)
    run("opt", 2048, 32, 8192, "relu", n_layer=24, V=512, D=64)
    run("opt", 2048, 32, 8192, "relu", n_layer=24, V=50272, D=768)
    run("gpt2", 2048, 32, 8192, "gelu_new", n_layer=24, V=50272, D=768)
