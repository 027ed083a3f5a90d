"""Which 8-bit format should carry the activation GRADIENTS of the fp8 LM (BASELINE configs[4])?  A CPU study on the numerics
model of oracle/fp8_sim.py (no GPU): the frozen LM's Linear layers with e4m3 per-tensor weights, row-wise quantised forward
activations, and the backward operand (dY rows) in e4m3 / e5m2 / bf16; the judge is the fp32 oracle on the same (round-tripped)
weights.  Findings this test pins (numbers printed; toy OPT / GPT-2, 2 layers; the same study at 8 layers, E = 512, V = 8192 gave
OPT cos 0.9675 / 0.9623 / 0.9696 and GPT-2 0.9984 / 0.9938 / 1.0000 for e4m3 / e5m2 / bf16 gradients):
  * e5m2 gradients are WORSE than e4m3 gradients (2 mantissa bits against 3; the row scale already keeps dY inside e4m3's range);
  * bf16 gradients (a dgrad at half the MFMA rate) buy < 0.005 of cosine over e4m3: with ReLU the deviation comes from the FORWARD
    activations' quantisation flipping ReLU derivatives, not from the gradient operand;
  * the softmax tail that e4m3's range flushes in d logits (V = 50 272 entries of ~1/V beside the label's ~1) moves the lm_head dgrad
    by < 1 % (second test).
So the product path keeps e4m3 for both operands of every fp8 GEMM (csrc/gemm_fp8.hip), forward and backward."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import oracle  # noqa: E402
from _metrics import grad_stats  # noqa: E402
from eavqa_amd.models.lm import LMConfig, random_init_state_dict  # noqa: E402


def _q_rows(x, fmt):
    xb = x.bfloat16().float()                      # the operand lives in HBM as bf16
    if fmt == "bf16":
        return xb
    amax = xb.abs().amax(-1, keepdim=True)
    mx, dt = (448.0, torch.float8_e4m3fn) if fmt == "e4m3" else (57344.0, torch.float8_e5m2)
    scale = torch.where(amax > 0, amax / mx, torch.ones_like(amax))
    return (xb / scale).to(dt).float() * scale


def _linear(fwd_fmt, bwd_fmt):
    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(w)
            return _q_rows(x, fwd_fmt) @ w.T

        @staticmethod
        def backward(ctx, dy):
            (w,) = ctx.saved_tensors
            d2 = dy.reshape(-1, dy.shape[-1])
            return (_q_rows(d2, bwd_fmt) @ w).reshape(*dy.shape[:-1], w.shape[1]), None

    return lambda x, w: Fn.apply(x, w)


def _roundtrip(w):
    s = w.abs().max() / 448.0
    return (w / s).to(torch.float8_e4m3fn).float() * s


@pytest.mark.parametrize("arch", ["opt", "gpt2"])
def test_gradient_operand_format_study(arch):
    NL, E, H, F, V, B, T, L, D = 2, 256, 4, 512, 640, 6, 24, 4, 32
    cfg = (LMConfig("opt", NL, H, E, F, V, 64, 1e-5, "relu", 2, 1) if arch == "opt" else LMConfig("gpt2", NL, H, E, F, V, 64, 1e-5, "gelu_new", V - 1, None))
    sd = random_init_state_dict(cfg, 7, "cpu")
    wq = {k: ((_roundtrip(v.T).T.contiguous() if arch == "gpt2" else _roundtrip(v)) if v.dim() == 2 and any(t in k for t in ("proj", "fc", "c_attn")) else v)
          for k, v in sd.items()}
    wq["lm_head.weight"] = _roundtrip(sd["model.decoder.embed_tokens.weight" if arch == "opt" else "transformer.wte.weight"])
    g = torch.Generator().manual_seed(2)
    lens = torch.randint(8, T + 1, (B,), generator=g)
    lens[0] = T
    pad = V - 1
    ids = torch.randint(3, V - 2, (B, T), generator=g)
    mask = (torch.arange(T)[None] < lens[:, None]).long()
    ids = ids * mask + pad * (1 - mask)
    labels = oracle.label_mask_cc(ids, pad)
    prefix = torch.randn(B, D, generator=g)
    torch.manual_seed(1)
    l0, l2 = torch.nn.Linear(D, E * L // 2), torch.nn.Linear(E * L // 2, E * L)
    base = {"model.0.weight": l0.weight.detach(), "model.0.bias": l0.bias.detach(), "model.2.weight": l2.weight.detach(), "model.2.bias": l2.bias.detach()}
    ocfg = dict(arch=arch, n_layer=NL, n_head=H, act=cfg.act)

    def grads(linear_fn):
        mp = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        c = dict(ocfg, **({"linear_fn": linear_fn} if linear_fn is not None else {}))
        loss, _ = oracle.clipcap_forward(wq, c, mp, dict(prefix_length=L, mapping_type="mlp"), ids, prefix, mask, labels)
        loss.backward()
        return loss.item(), {k: p.grad for k, p in mp.items()}

    l_ref, g_ref = grads(None)
    cos = {}
    for bwd in ("e4m3", "e5m2", "bf16"):
        l, gg = grads(_linear("e4m3", bwd))
        cos[bwd], ratio, maxrel = grad_stats(gg, g_ref)
        print(f"[{arch}] forward e4m3, gradient operand {bwd}: |d loss| {abs(l - l_ref):.2e}  cosine {cos[bwd]:.4f}  norm ratio {ratio:.4f}  max rel {maxrel:.3f}")
        assert abs(ratio - 1.0) <= 0.02 and abs(l - l_ref) <= 5e-3
    assert cos["e4m3"] >= cos["e5m2"] - 1e-3, cos                    # e5m2 does not beat e4m3 for the gradient operand
    assert cos["bf16"] - cos["e4m3"] <= 5e-3, cos                     # and a bf16 dgrad buys next to nothing
    assert cos["e4m3"] >= (0.97 if arch == "opt" else 0.998), cos


def test_e4m3_flush_of_the_softmax_tail_in_dlogits():
    """d logits = softmax - onehot at V = 50 272: beside the label's entry (~1) the tail entries (~1/V) sit below e4m3's range under
    the row scale amax / 448 and flush to zero (91 % of them for a peaked distribution).  What that costs the lm_head dgrad
    (d logits @ W): relative error < 2 %, cosine > 0.9999 - the label term carries the gradient."""
    torch.manual_seed(0)
    V, E, R = 50272, 256, 32
    W = torch.randn(V, E) * 0.02
    for sharp, bound in ((1.0, 2e-3), (4.0, 2e-2)):
        p = torch.softmax(torch.randn(R, V) * sharp, -1)
        d = p.clone()
        d[torch.arange(R), torch.randint(0, V, (R,))] -= 1.0
        d /= R
        ref = d.double() @ W.double()
        q = _q_rows(d, "e4m3")
        out = q.double() @ W.double()
        rel = ((out - ref).norm() / ref.norm()).item()
        cosv = ((out * ref).sum() / (out.norm() * ref.norm())).item()
        print(f"sharpness {sharp}: {100 * (q == 0).float().mean().item():.1f} % of d logits flushed, dgrad rel err {rel:.4f}, cosine {cosv:.6f}")
        assert rel <= bound and cosv >= 0.9999
