"""GPU: the fp8 path of BASELINE configs[4] - row-wise e4m3 quantisation (eavqa_quantize_rows_fp8) and the block-scaled-MFMA GEMM
(eavqa_gemm_fp8) - against torch's OCP float8_e4m3fn on the CPU.

The products of two e4m3 values are exact in fp32 and the kernel accumulates in fp32, so against a float64 reference computed from
the SAME quantised operands the GEMM is held to fp32-accumulation tolerance; small-integer operands are exact."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle
from _metrics import grad_stats

DEV = "cuda"
TILES = {1: "128x80", 2: "256x128", 3: "256x160", 4: "128x128", 5: "128x256"}


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def deq(q_u8, scale):
    """uint8 e4m3 bytes (+ row scales) -> float64 values."""
    v = q_u8.cpu().view(torch.float8_e4m3fn).double()
    return v * scale.cpu().double()[:, None] if scale is not None else v


def quant_ref(x):
    """The kernel's arithmetic restated in torch fp32: scale = amax * (1/448), q = e4m3(x * (1/scale))."""
    xf = x.float()
    amax = xf.abs().amax(1)
    scale = torch.where(amax > 0, amax * torch.tensor(1.0 / 448.0, dtype=torch.float32), torch.ones_like(amax))
    inv = 1.0 / scale
    q = (xf * inv[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(1, 128), (7, 4096), (33, 16384), (70, 1280), (3, 12), (5, 8192), (4, 20480), (9, 2056)])
def test_quantize_rows_matches_torch_e4m3fn(dtype, rows, cols):
    from eavqa_amd import ops
    x = rnd(rows, cols, seed=rows, scale=3.0, dtype=dtype)
    x[0, : min(cols, 8)] = 0
    if rows > 2:
        x[2] = 0                                    # an all-zero row: scale 1, bytes 0
    q, sc = ops.quantize_rows_fp8(x.to(DEV))
    q_ref, sc_ref = quant_ref(x)
    assert torch.equal(sc.cpu(), sc_ref)
    assert torch.equal(q.cpu(), q_ref), (q.cpu() != q_ref).sum().item()
    back = deq(q, sc)
    assert (back - x.double()).abs().max().item() <= x.float().abs().amax().item() * 2 ** -4 * 1.01   # e4m3: 3 mantissa bits


@pytest.mark.parametrize("tile", sorted(TILES), ids=[TILES[t] for t in sorted(TILES)])
def test_gemm_fp8_exact_on_small_integers(tile):
    """Integer-valued e4m3 operands: every product and partial sum is exact, so any mismatch is a layout error (the k-groups of
    A and B must pair up, rows / columns must not be transposed).  A carries a different pattern per row, B per column."""
    from eavqa_amd import ops
    M, N, K = 150, 200, 384
    g = torch.Generator().manual_seed(4)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    b = torch.randint(-2, 3, (N, K), generator=g).float()
    a[:, ::7] = 2 * (torch.arange(M)[:, None] % 3).float() - 1        # structure along m and along k
    b[:, 5::11] = (torch.arange(N)[:, None] % 4).float() - 2
    aq = a.to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)
    bq = b.to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)
    ones = torch.ones(M, device=DEV)
    out = ops.gemm_fp8(aq, ones, bq, 1.0, out_f32=True, tile=tile)
    assert torch.equal(out.cpu(), a @ b.T)


@pytest.mark.parametrize("tile", [0] + sorted(TILES), ids=["auto"] + [TILES[t] for t in sorted(TILES)])
@pytest.mark.parametrize("M,N,K", [(1, 64, 128), (70, 2048, 2048), (257, 1000, 640), (1943, 1280, 1280), (2048, 4096, 4096)])
def test_gemm_fp8_matches_dequantised_reference(tile, M, N, K):
    from eavqa_amd import ops
    x, w = rnd(M, K, seed=1, dtype=torch.bfloat16), rnd(N, K, seed=2, scale=0.02)
    xq, xs = ops.quantize_rows_fp8(x.to(DEV))
    w_scale = w.abs().max().item() / 448.0
    wq = (w / w_scale).to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)
    ref = deq(xq, xs) @ deq(wq, None).T * w_scale
    out = ops.gemm_fp8(xq, xs, wq, w_scale, out_f32=True, tile=tile)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * math.sqrt(K) * max(1.0, ref.abs().max().item()), err
    # against the UNquantised product: what the quantisation itself costs (reported, loosely bounded)
    full = x.double() @ w.double().T
    rel = (out.cpu().double() - full).norm() / full.norm()
    assert rel < 6e-2, rel


def test_gemm_fp8_epilogue():
    """bias + gelu_new + aux_out + fp32 residual + bf16 output, and the activation-derivative form (aux_in)."""
    from eavqa_amd import ops
    M, N, K = 300, 264, 256
    x, w = rnd(M, K, seed=5, dtype=torch.bfloat16), rnd(N, K, seed=6, scale=0.05)
    bias, res = rnd(N, seed=7), rnd(M, N, seed=8)
    xq, xs = ops.quantize_rows_fp8(x.to(DEV))
    w_scale = w.abs().max().item() / 448.0
    wq = (w / w_scale).to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)
    pre = (deq(xq, xs) @ deq(wq, None).T * w_scale).float() * 0.5 + bias
    aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    out = ops.gemm_fp8(xq, xs, wq, w_scale, bias=bias.to(DEV), act="gelu_new", aux_out=aux, residual=res.to(DEV), out_f32=True, alpha=0.5)
    assert (aux.float().cpu() - pre).abs().max().item() <= 2e-2 * max(1.0, pre.abs().max().item())
    assert (out.cpu() - (oracle.gelu_new(pre) + res)).abs().max().item() <= 1e-4 * max(1.0, pre.abs().max().item())
    out_bf = ops.gemm_fp8(xq, xs, wq, w_scale, bias=bias.to(DEV), act="relu")
    assert out_bf.dtype == torch.bfloat16
    want = torch.relu((deq(xq, xs) @ deq(wq, None).T * w_scale).float() + bias)
    assert (out_bf.float().cpu() - want).abs().max().item() <= 1e-2 * max(1.0, want.abs().max().item())
    u = rnd(M, N, seed=9, dtype=torch.bfloat16)
    uu = u.float().clone().requires_grad_(True)
    torch.relu(uu).sum().backward()
    got = ops.gemm_fp8(xq, xs, wq, w_scale, act="relu", aux_in=u.to(DEV), out_f32=True)
    assert (got.cpu() - (deq(xq, xs) @ deq(wq, None).T * w_scale).float() * uu.grad).abs().max().item() <= 1e-4


def test_gemm_fp8_rejects_bad_shapes():
    from eavqa_amd import ops
    from eavqa_amd._lib import EavqaError
    a = torch.zeros(8, 96, dtype=torch.uint8, device=DEV)      # K % 128 != 0
    with pytest.raises(EavqaError):
        ops.gemm_fp8(a, torch.ones(8, device=DEV), a, 1.0)


def _roundtrip_e4m3(w):
    """Per-tensor e4m3 round trip (the weight format of models/lm.py ``Fp8Weight``), on the CPU."""
    amax = w.abs().max().item()
    scale = amax / 448.0 if amax > 0 else 1.0
    return (w / scale).to(torch.float8_e4m3fn).float() * scale


def _fp8_oracle_weights(sd, arch):
    """The oracle's weights for an fp8 LM: every Linear weight of every layer and the lm_head round-tripped through e4m3 (the
    embedding gather keeps the original wte), so that a comparison with the HIP path isolates the KERNEL's error - row-wise
    activation quantisation and fp32 accumulation order - from the weight quantisation both sides share."""
    out = dict(sd)
    if arch == "opt":
        lin = [k for k in sd if k.endswith("_proj.weight") or k.endswith("fc1.weight") or k.endswith("fc2.weight")]
        # q / k / v are quantised as ONE fused [3E, E] tensor (one scale), exactly as the LM packs them
        layers = sorted({k.split(".self_attn.")[0] for k in sd if ".self_attn.q_proj.weight" in k})
        for p in layers:
            fused = _roundtrip_e4m3(torch.cat([sd[f"{p}.self_attn.{n}_proj.weight"] for n in "qkv"], 0))
            E = fused.shape[1]
            for i, n in enumerate("qkv"):
                out[f"{p}.self_attn.{n}_proj.weight"] = fused[i * E:(i + 1) * E]
            out[f"{p}.self_attn.out_proj.weight"] = _roundtrip_e4m3(sd[f"{p}.self_attn.out_proj.weight"])
        for k in lin:
            if "fc1" in k or "fc2" in k:
                out[k] = _roundtrip_e4m3(sd[k])
        out["lm_head.weight"] = _roundtrip_e4m3(sd["model.decoder.embed_tokens.weight"])
    else:
        for k in sd:
            if k.endswith("c_attn.weight") or k.endswith("c_proj.weight") or k.endswith("c_fc.weight"):
                out[k] = _roundtrip_e4m3(sd[k].T).T.contiguous()        # the LM quantises the packed [out, in] matrix: same elements
        out["lm_head.weight"] = _roundtrip_e4m3(sd["transformer.wte.weight"])
    return out


@pytest.mark.parametrize("arch", ["opt", "gpt2"])
def test_fp8_lm_training_step_against_weight_roundtripped_oracle(arch):
    """A small LM with the fp8 weight format (E and FFN multiples of 128): loss, attended logits and mapper gradients against the
    fp32 oracle whose Linear weights went through the same e4m3 round trip (activations exact there: this bounds what the fp8
    FORMAT costs - measured on MI355X: logits 0.31 at |logits| 4.4, loss 3e-3, mapper gradients 34 % of the largest entry after
    two layers of row-wise 3-mantissa-bit activations and gradients), and against the fp8 numerics model of oracle/fp8_sim.py,
    which quantises the same operands at the same places but keeps everything else in fp32.  Measured: the two comparisons
    come out alike (logits 0.39, gradients 27 %) - at e4m3's 2^-4 spacing the bf16 roundings of the surrounding kernels (which
    alone move this deliberately sensitive model's gradients by 8 %, first line printed) flip enough quantisation decisions to
    decorrelate the two, so a model-level comparison cannot separate kernel error from format noise.  What pins the KERNELS
    is above: the quantiser is bit-exact against torch's e4m3fn and the GEMM is exact on integers and within fp32 accumulation
    error on the same quantised operands, for every tile."""
    from eavqa_amd.models.clipcap import ClipCaptionPrefix
    from eavqa_amd.models.lm import FrozenCausalLM, LMConfig, random_init_state_dict
    E, H, F, NL, V, L, D, B, T = 256, 4, 512, 2, 640, 4, 32, 6, 24
    cfg = (LMConfig("opt", NL, H, E, F, V, 64, 1e-5, "relu", 2, 1) if arch == "opt" else LMConfig("gpt2", NL, H, E, F, V, 64, 1e-5, "gelu_new", V - 1, None))
    sd = random_init_state_dict(cfg, 7, "cpu")
    g = torch.Generator().manual_seed(2)
    for k in sorted(sd):
        if k.endswith("bias") or "ln_" in k or "layer_norm" in k:
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
        elif sd[k].dim() == 2:
            sd[k] = sd[k] * 3.0                                        # larger weights: logits of a useful size
    lens = torch.randint(6, T + 1, (B,), generator=g); lens[0] = T
    pad = V - 1
    ids = torch.randint(3, V - 2, (B, T), generator=g)
    mask = (torch.arange(T)[None] < lens[:, None]).long()
    ids = ids * mask + pad * (1 - mask)
    labels = oracle.label_mask_cc(ids, pad)
    prefix = torch.randn(B, D, generator=g)
    res = {}
    mapper_sd = None
    for fmt in ("native", "fp8"):
        lm = FrozenCausalLM(cfg, sd, torch.bfloat16, DEV, weight_format=fmt)
        torch.manual_seed(1)
        model = ClipCaptionPrefix(prefix_length=L, prefix_size=D, mapping_type="mlp", lm=lm, dtype=torch.bfloat16, device=DEV).train()
        if mapper_sd is None:
            mapper_sd = {k: v.detach().float().cpu().clone() for k, v in model.clip_project.state_dict().items()}
        out = model(question_tokens=ids, prefix=prefix, question_mask=mask, labels=labels)
        out.loss.backward()
        res[fmt] = (out.loss.item(), out.logits.float().cpu(), {k: p.grad.float().cpu() for k, p in model.clip_project.named_parameters()})
    from oracle import fp8_sim
    att = torch.cat([torch.ones(B, L, dtype=torch.bool), mask.bool()], 1)
    base = dict(arch=arch, n_layer=NL, n_head=H, act=cfg.act)
    wq = _fp8_oracle_weights(sd, arch)
    # bounds: (|d loss|, max |d logits|, cosine of the whole mapper gradient >=, |norm ratio - 1| <=).  The direction / length bounds
    # are what a wrong dgrad cannot pass (zero gradient: ratio 0; sign error: cosine -1; a wrong scale or a transposed weight: cosine
    # near 0); max-rel of the largest entry is printed only.  CPU numerics model of the same dataflow (tests/test_fp8_numerics_model.py):
    # cosine 0.988 (OPT: e4m3 forward activations flip ReLU derivatives) / 0.999 (GPT-2) against the round-tripped fp32 oracle.
    # Measured on MI355X (round 3): bf16 0.9976 / 0.9999; fp8 vs the format-cost oracle 0.9784 / 0.9961; fp8 vs the numerics model
    # 0.9767 / 0.9949; norm ratios 0.988 .. 1.000.
    relu = arch == "opt"
    cases = (("native", "bf16 LM vs fp32 oracle", sd, base, dict(loss=5e-3, logits=8e-2, cos=0.995 if relu else 0.9995, ratio=0.02)),
             ("fp8", "fp8 LM vs fp32 oracle with e4m3-round-tripped weights (what the FORMAT costs)", wq, base,
              dict(loss=2e-2, logits=0.6, cos=0.96 if relu else 0.99, ratio=0.03)),
             ("fp8", "fp8 LM vs the fp8 numerics model (oracle/fp8_sim.py: same quantisation points, fp32 elsewhere)", wq,
              dict(base, linear_fn=fp8_sim.fp8_linear), dict(loss=2e-2, logits=0.6, cos=0.96 if relu else 0.99, ratio=0.03)))
    for fmt, what, wsd, ocfg, tol in cases:
        mp = {k: v.clone().requires_grad_(True) for k, v in mapper_sd.items()}
        loss, logits = oracle.clipcap_forward(wsd, ocfg, mp, dict(prefix_length=L, mapping_type="mlp"), ids, prefix, mask, labels)
        loss.backward()
        l, lg, gr = res[fmt]
        e_log = (lg - logits.detach())[att].abs().max().item()
        cos, ratio, e_grad = grad_stats(gr, {k: p.grad for k, p in mp.items()})
        print(f"[{arch}] {what}: |d loss| {abs(l - loss.item()):.3e}  max|d logits| {e_log:.3e} (|logits| max {logits.abs().max().item():.2f})  "
              f"gradient cosine {cos:.4f}  norm ratio {ratio:.4f}  max rel {e_grad:.3e}")
        assert abs(l - loss.item()) <= tol["loss"] and e_log <= tol["logits"], what
        assert cos >= tol["cos"] and abs(ratio - 1.0) <= tol["ratio"], (what, cos, ratio)


@pytest.mark.parametrize("arch", ["opt", "gpt2"])
def test_fp8_and_bf16_lm_train_the_mapper_alike(arch):
    """The same small mapper trained 30 steps through a bf16 frozen LM and through the fp8 (e4m3 weights + activations + activation
    gradients) frozen LM, from the same initialisation on the same batch: the two loss curves must fall together.  The band is
    relative to the loss DROP of the bf16 run, so a gradient that does not train (loss flat) or trains elsewhere fails."""
    from eavqa_amd.models.clipcap import ClipCaptionPrefix
    from eavqa_amd.models.lm import FrozenCausalLM, LMConfig, random_init_state_dict
    from eavqa_amd.trainers.optim import FusedAdamW
    E, H, F, NL, V, L, D, B, T, STEPS = 256, 4, 512, 4, 640, 4, 32, 16, 24, 30
    cfg = (LMConfig("opt", NL, H, E, F, V, 64, 1e-5, "relu", 2, 1) if arch == "opt" else LMConfig("gpt2", NL, H, E, F, V, 64, 1e-5, "gelu_new", V - 1, None))
    sd = random_init_state_dict(cfg, 7, "cpu")
    for k in sorted(sd):
        if sd[k].dim() == 2:
            sd[k] = sd[k] * 3.0
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(6, T + 1, (B,), generator=g); lens[0] = T
    pad = V - 1
    ids = torch.randint(3, V - 2, (B, T), generator=g)
    mask = (torch.arange(T)[None] < lens[:, None]).long()
    ids = ids * mask + pad * (1 - mask)
    labels = oracle.label_mask_cc(ids, pad)
    prefix = torch.randn(B, D, generator=g)
    curves, init = {}, None
    for fmt in ("native", "fp8"):
        lm = FrozenCausalLM(cfg, sd, torch.bfloat16, DEV, weight_format=fmt)
        torch.manual_seed(1)
        model = ClipCaptionPrefix(prefix_length=L, prefix_size=D, mapping_type="mlp", lm=lm, dtype=torch.bfloat16, device=DEV).train()
        if init is None:
            init = {k: v.detach().clone() for k, v in model.clip_project.state_dict().items()}
        else:
            model.clip_project.load_state_dict(init)
        opt = FusedAdamW(model.clip_project.flat, lr=2e-3)
        losses = []
        for _ in range(STEPS):
            out = model(question_tokens=ids, prefix=prefix, question_mask=mask, labels=labels)
            out.loss.backward()
            opt.step()
            opt.zero_grad()
            losses.append(out.loss.item())
        curves[fmt] = losses
    a, b = torch.tensor(curves["native"]), torch.tensor(curves["fp8"])
    drop = (a[0] - a[-1]).item()
    gap = (a - b).abs().max().item()
    print(f"[{arch}] 30 steps: bf16 loss {a[0]:.4f} -> {a[-1]:.4f}, fp8 loss {b[0]:.4f} -> {b[-1]:.4f}, max |gap| {gap:.4f} = {100 * gap / drop:.1f} % of the bf16 drop")
    assert drop > 0.3, "the bf16 run must actually train for the comparison to mean anything"
    # measured on MI355X: max gap 1.2 % (OPT) / 0.7 % (GPT-2) of the bf16 drop
    assert (b[0] - b[-1]).item() >= 0.95 * drop and gap <= 0.05 * drop, (curves, gap, drop)


@pytest.mark.parametrize("M,N,K,ks", [(32, 384, 256, 1), (32, 12288, 4096, 2), (5, 200, 1024, 4), (64, 2048, 512, 2), (17, 640, 2048, None), (32, 4096, 16384, 32)])
def test_gemm_fp8_splitk_partial_sums(M, N, K, ks):
    """eavqa_gemm_fp8_splitk: sum_s partials[s] == a_row_scale[m] * b_scale * (A_q B_q^T) against a float64 product of the SAME bytes, and
    against eavqa_gemm_fp8 (the kernel the re-forward loop uses).  Bound 1e-4 of the largest entry: v_mfma_f32_16x16x32_fp8_fp8 adds its 32
    products with fewer guard bits than the block-scaled 16x16x128 form does (measured 2e-5 relative against float64, where the block-scaled
    kernel is exact on integers) - two orders of magnitude below the bf16 rounding of every tensor that follows."""
    from eavqa_amd import ops
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, K, generator=g) * 2).to(torch.bfloat16).to(DEV)
    w = torch.randn(N, K, generator=g) * 0.05
    aq, asc = ops.quantize_rows_fp8(a)
    wmax = w.abs().max().item()
    wq = (w / (wmax / 448.0)).to(torch.float8_e4m3fn).view(torch.uint8).contiguous().to(DEV)
    bsc = wmax / 448.0
    part = ops.gemm_fp8_splitk(aq, asc, wq, bsc, ks=ks)
    got = part.double().sum(0).cpu()
    ref = (aq.cpu().view(torch.float8_e4m3fn).double() @ wq.cpu().view(torch.float8_e4m3fn).double().T) * asc.cpu().double()[:, None] * bsc
    tol = 2e-6 * math.sqrt(K) * max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() <= tol
    full = ops.gemm_fp8(aq, asc, wq, bsc, out_f32=True)
    assert (got - full.double().cpu()).abs().max().item() <= tol


@pytest.mark.parametrize("rows,cols,ks", [(32, 4096, 8), (5, 256, 0), (64, 2560, 3), (1, 128, 1)])
def test_layernorm_splitk_fp8_is_layernorm_then_row_quantiser(rows, cols, ks):
    """eavqa_layernorm_splitk_fp8 == eavqa_layernorm_splitk (bf16 out) followed by eavqa_quantize_rows_fp8, byte for byte, and the same x_out."""
    from eavqa_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    x_in = (torch.randn(rows, cols, generator=g) + 0.3).to(DEV)
    part = (torch.randn(max(ks, 1), rows, cols, generator=g) * 0.5).to(DEV) if ks else None
    bias = torch.randn(cols, generator=g).to(DEV) if ks else None
    gamma, beta = (1 + 0.2 * torch.randn(cols, generator=g)).to(DEV), (0.1 * torch.randn(cols, generator=g)).to(DEV)
    x1, x2 = torch.empty_like(x_in), torch.empty_like(x_in)
    y = ops.layernorm_splitk(x_in, gamma, beta, 1e-5, torch.bfloat16, part=part, bias=bias, x_out=x1)
    q_ref, s_ref = ops.quantize_rows_fp8(y)
    q, sc = ops.layernorm_splitk_fp8(x_in, gamma, beta, 1e-5, part=part, bias=bias, x_out=x2)
    torch.cuda.synchronize()
    assert torch.equal(x1, x2) and torch.equal(sc, s_ref) and torch.equal(q, q_ref)


@pytest.mark.parametrize("rows,cols", [(1943, 1280), (2048, 4096), (7, 128), (130, 2048), (33, 520)])
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
def test_layernorm_fwd_bwd_fp8_are_the_kernel_pairs_they_replace(rows, cols, x_dtype):
    """eavqa_layernorm_fwd_fp8 == eavqa_layernorm_fwd (bf16 out) + eavqa_quantize_rows_fp8 and eavqa_layernorm_bwd_fp8 ==
    eavqa_layernorm_bwd (bf16 copy of dx) + eavqa_quantize_rows_fp8: bytes, row scales, statistics and the fp32 dx, bit for bit."""
    from eavqa_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 1.5 + 0.3).to(x_dtype).to(DEV)
    x[rows // 2] = 0                                                       # a constant row: xhat = 0, the output is beta
    gamma, beta = (1 + 0.2 * torch.randn(cols, generator=g)).to(DEV), (0.1 * torch.randn(cols, generator=g)).to(DEV)
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5, torch.bfloat16, save_stats=True)
    q_ref, s_ref = ops.quantize_rows_fp8(y)
    q, sc, mean2, rstd2 = ops.layernorm_fwd_fp8(x, gamma, beta, 1e-5, save_stats=True)
    q3, sc3 = ops.layernorm_fwd_fp8(x, gamma, beta, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(q, q_ref) and torch.equal(sc, s_ref) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    assert torch.equal(q3, q_ref) and torch.equal(sc3, s_ref)
    if cols > 4096:
        return
    dy = (torch.randn(rows, cols, generator=g) * 0.01).to(torch.bfloat16).to(DEV)
    dy[0] = 0                                                              # an all-zero gradient row: scale 1, zero bytes
    dres = torch.randn(rows, cols, generator=g).to(DEV) * 0.01
    dres[0] = 0
    lowp = torch.empty((rows, cols), device=DEV, dtype=torch.bfloat16)
    dx_ref = ops.layernorm_bwd(x, dy, gamma, mean, rstd, dres=dres, lowp_out=lowp)
    dq_ref, ds_ref = ops.quantize_rows_fp8(lowp)
    dq, ds = torch.empty((rows, cols), device=DEV, dtype=torch.uint8), torch.empty(rows, device=DEV)
    dx = ops.layernorm_bwd_fp8(x, dy, gamma, mean, rstd, dq, ds, dres=dres)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref) and torch.equal(dq, dq_ref) and torch.equal(ds, ds_ref)
    assert ds[0].item() == 1.0 and int(dq[0].max().item()) == 0


@pytest.mark.parametrize("arch", ["opt", "gpt2"])
def test_fp8_training_step_is_bit_equal_with_and_without_the_fused_quantiser(arch):
    """``FrozenCausalLM.fuse_quantizer``: the fp8 training step with LayerNorm (forward and backward) handing over quantised rows against the
    same step with the separate eavqa_quantize_rows_fp8 launches - same bytes into every GEMM, so loss, logits and gradients are identical."""
    from eavqa_amd.models.clipcap import ClipCaptionPrefix
    from eavqa_amd.models.lm import FrozenCausalLM, LMConfig, random_init_state_dict
    E, H, F, NL, V, L, D, B, T = 256, 4, 512, 3, 640, 4, 32, 6, 24
    cfg = (LMConfig("opt", NL, H, E, F, V, 64, 1e-5, "relu", 2, 1) if arch == "opt" else LMConfig("gpt2", NL, H, E, F, V, 64, 1e-5, "gelu_new", V - 1, None))
    sd = random_init_state_dict(cfg, 7, "cpu")
    g = torch.Generator().manual_seed(2)
    lens = torch.randint(6, T + 1, (B,), generator=g); lens[0] = T
    pad = V - 1
    ids = torch.randint(3, V - 2, (B, T), generator=g)
    mask = (torch.arange(T)[None] < lens[:, None]).long()
    ids = ids * mask + pad * (1 - mask)
    labels = oracle.label_mask_cc(ids, pad)
    prefix = torch.randn(B, D, generator=g)
    res = []
    for fuse in (True, False):
        lm = FrozenCausalLM(cfg, sd, torch.bfloat16, DEV, weight_format="fp8")
        assert lm.fuse_quantizer                                           # the default
        lm.fuse_quantizer = fuse
        torch.manual_seed(1)
        model = ClipCaptionPrefix(prefix_length=L, prefix_size=D, mapping_type="mlp", lm=lm, dtype=torch.bfloat16, device=DEV).train()
        out = model(question_tokens=ids, prefix=prefix, question_mask=mask, labels=labels)
        out.loss.backward()
        torch.cuda.synchronize()
        res.append((out.loss.item(), out.logits.float().cpu(), [p.grad.float().cpu().clone() for p in model.clip_project.parameters()]))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
    assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2]))


@pytest.mark.parametrize("arch", ["opt", "gpt2"])
def test_fp8_lm_cached_generation_returns_the_ids_of_the_reforward_loop(arch):
    """``generate()`` (use_cache=True by default) on an LM held in e4m3: prefill + cached decode steps through ``eavqa_lm_block_forward_fp8``
    (e4m3 weights streamed once per step, rows quantised exactly as the re-forward path quantises them) returns the ids of
    ``use_cache=False`` - the reference's own loop (src/models/clipcap.py:414-419) through the fp8 GEMMs."""
    from eavqa_amd.models.clipcap import ClipCaptionPrefix
    from eavqa_amd.models.lm import FrozenCausalLM, LMConfig, random_init_state_dict
    cfg = (LMConfig("opt", 3, 4, 256, 512, 640, 64, 1e-5, "relu", 2, 1) if arch == "opt" else LMConfig("gpt2", 3, 4, 256, 512, 640, 64, 1e-5, "gelu_new", 639, None))
    sd = random_init_state_dict(cfg, 3, "cpu")
    for k in sorted(sd):
        if sd[k].dim() == 2:
            sd[k] = sd[k] * 3.0                                        # logits of a useful size: top-2 gaps far above the rounding noise
    lm = FrozenCausalLM(cfg, sd, torch.bfloat16, DEV, weight_format="fp8")
    torch.manual_seed(1)
    model = ClipCaptionPrefix(prefix_length=4, prefix_size=32, mapping_type="mlp", lm=lm, dtype=torch.bfloat16, device=DEV).eval()
    g = torch.Generator().manual_seed(4)
    B, T = 5, 9
    ids = torch.randint(3, 600, (B, T), generator=g)
    mask = torch.ones(B, T, dtype=torch.long)
    mask[1, 6:] = 0
    mask[3, 4:] = 0
    prefix = torch.randn(B, 32, generator=g)
    import warnings
    kw = dict(question_tokens=ids, prefix=prefix, question_mask=mask, max_length=6, pad_token_id=1, eos_token_id=None, output_scores=True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                # no fallback warning any more
        a, lpa = model.generate(**kw)
    b, lpb = model.generate(use_cache=False, **kw)
    assert len(a) == B and len(a[0]) == 6
    # Two correct evaluations of the same e4m3 arithmetic differ: the fp8 matrix instructions add the products of one instruction with
    # limited alignment (test_gemm_fp8_splitk_partial_sums: ~1e-5 between kernels that cut K differently), a bf16 rounding in between flips
    # on such a difference now and then, and the e4m3 quantisation of the NEXT activation turns a flipped bf16 bit into a 6 % step of that
    # element.  Measured on this 3-layer model: per-step log-probabilities of the two runs 0.02-0.18 nats apart with equal ids.  So: a row
    # must agree up to its first step whose two winners are within that band (a near-tie broken differently), at least 85 % of all
    # (row, step) pairs must agree, and the first step - the prefill, the same kernels in both runs - must agree outright.
    agree = 0
    for r in range(B):
        assert a[r][0] == b[r][0] and abs(float(lpa[r, 0]) - float(lpb[r, 0])) <= 1e-3
        for t in range(6):
            d = abs(float(lpa[r, t]) - float(lpb[r, t]))
            assert d <= 0.4, (r, t, a[r], b[r], float(lpa[r, t]), float(lpb[r, t]))
            if a[r][t] != b[r][t]:
                break
            agree += 1
    print(f"[{arch}] cached vs re-forward, fp8 LM: {agree} of {B * 6} (row, step) pairs agree; max |d logp| "
          f"{(lpa - lpb).abs().max().item():.3f}")
    assert agree >= 0.85 * B * 6


def test_fp8_lm_rejects_unsupported_uses():
    from eavqa_amd.models.lm import FrozenCausalLM, LMConfig, random_init_state_dict
    cfg = LMConfig("opt", 1, 4, 192, 384, 64, 32, 1e-5, "relu", 2, 1)            # E not a multiple of 128
    with pytest.raises(ValueError, match="multiples of 128"):
        FrozenCausalLM(cfg, random_init_state_dict(cfg, 1, "cpu"), torch.bfloat16, DEV, weight_format="fp8")
    cfg = LMConfig("opt", 1, 2, 128, 256, 64, 32, 1e-5, "relu", 2, 1)
    with pytest.raises(ValueError, match="bfloat16"):
        FrozenCausalLM(cfg, random_init_state_dict(cfg, 1, "cpu"), torch.float32, DEV, weight_format="fp8")
