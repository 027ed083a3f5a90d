"""GPU: eavqa_gemm_ln - the LayerNorm of a frozen pre-LN decoder layer folded into the two products around it (include/eavqa.h;
HF modeling_gpt2.py:246-309 `ln_1 -> c_attn`, `ln_2 -> c_fc`, modeling_opt.py:184-254).

Producer side (second copy of the result + per-row (sum, sum of squares) spread over 64-column slots) and consumer side
(`rstd (x W'^T - mean c) + d`) are checked separately against float64 computations of the same bf16 / fp32 operands, on every kernel
family that carries the shared epilogue, then chained and compared with the call sequence they replace (eavqa_layernorm_fwd + eavqa_gemm).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle

DEV = "cuda"
ACTS = {"none": lambda x: x, "relu": torch.relu, "gelu_new": oracle.gelu_new}

# kernel selectors of include/eavqa_test.h: [13:8] forces a full-line (BK = 64) tile, [15:14] = 2 the 256 x 256 kernel, bit 7 the general
# kernel
K64 = {f"k64_{i}": i << 8 for i in (5, 6, 7, 11, 12)}        # knob ids of K64_SHAPES (gemm_k64.hip): the tiles the dispatcher picks from carry the LN form
PRODUCER_KNOBS = {"auto": 0, "big": 2 << 14, "general": 1 << 7, **K64}
CONSUMER_KNOBS = {"auto": 0, "big": 2 << 14, "general": 1 << 7, **K64}


@pytest.fixture(scope="module")
def ops():
    from eavqa_amd import ops as _ops, _lib
    assert _lib.load().eavqa_check_device() == 0, "not a gfx950 device"
    return _ops


@pytest.fixture()
def knob(ops):
    def set_(v):
        ops.KernelSelect.gemm = v
    yield set_
    ops.KernelSelect.gemm = 0


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def slots(n):
    return (n + 63) // 64


@pytest.mark.parametrize("M,N,K", [(300, 1280, 256), (129, 200, 128), (1943, 1280, 320), (70, 64, 64), (520, 2560, 192)])
@pytest.mark.parametrize("kn", list(PRODUCER_KNOBS))
def test_producer_copy_and_row_sums(ops, knob, M, N, K, kn):
    """x1 = ctx Wo^T + b + x (fp32 out) with the bf16 copy and the row statistics; two extra slots must come back as zeros."""
    a, w = rnd(M, K, seed=1, dtype=torch.bfloat16), rnd(N, K, seed=2, scale=0.05, dtype=torch.bfloat16)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4) + 0.25
    out = torch.empty((M, N), device=DEV, dtype=torch.float32)
    copy = torch.full((M + 1, N + 8), float("nan"), device=DEV, dtype=torch.bfloat16)
    stats = torch.full((M, slots(N) + 2, 2), float("nan"), device=DEV, dtype=torch.float32)
    knob(PRODUCER_KNOBS[kn])
    ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), residual=res.to(DEV), out=out, copy_out=copy[:M, :N], stats_out=stats)
    torch.cuda.synchronize()
    ref = a.double() @ w.double().T + bias.double() + res.double()
    got = out.cpu()
    assert (got.double() - ref).abs().max().item() <= 2e-4 * math.sqrt(K)
    c = copy.cpu()
    assert torch.equal(c[:M, :N], got.to(torch.bfloat16))                 # the copy is the stored value rounded once
    assert torch.isnan(c[M]).all() and torch.isnan(c[:, N:]).all()
    st = stats.cpu().double()
    assert not torch.isnan(st).any()
    s, ss = st[:, :, 0].sum(1), st[:, :, 1].sum(1)
    g64 = got.double()
    assert (s - g64.sum(1)).abs().max().item() <= 1e-5 * g64.abs().sum(1).max().item()
    assert ((ss - (g64 ** 2).sum(1)).abs() / (g64 ** 2).sum(1)).max().item() <= 1e-5
    assert (st[:, slots(N):] == 0).all()                                  # slots behind the last tile column: zeros


@pytest.mark.parametrize("M,N,K", [(300, 768, 256), (129, 200, 128), (1943, 3840, 1280), (70, 64, 64), (515, 5120, 1280)])
@pytest.mark.parametrize("kn", list(CONSUMER_KNOBS))
@pytest.mark.parametrize("act", ["none", "gelu_new"])
def test_consumer_is_layernorm_then_linear(ops, knob, M, N, K, kn, act):
    """rstd (x W'^T - mean c) + d against float64 LayerNorm(x) W^T + b of the same rounded operands; the statistics arrive in three
    unequal slots (any split must give the same mean / rstd)."""
    x = rnd(M, K, seed=1) * 1.7 + 0.4
    gamma, beta = 1.0 + rnd(K, seed=2, scale=0.2), rnd(K, seed=3, scale=0.1)
    w, bias = rnd(N, K, seed=4, scale=0.05), rnd(N, seed=5, scale=0.1)
    wf = (w * gamma).to(torch.bfloat16)                                    # W' as stored
    c = wf.float().sum(1)
    d = (w.double() @ beta.double() + bias.double()).float()
    xb = x.to(torch.bfloat16)
    x64 = x.double()
    cut = [0, K // 3, K // 2, K]
    st = torch.zeros((M, 3, 2), dtype=torch.float32)
    for i in range(3):
        st[:, i, 0] = x64[:, cut[i]:cut[i + 1]].sum(1).float()
        st[:, i, 1] = (x64[:, cut[i]:cut[i + 1]] ** 2).sum(1).float()
    mean_o = torch.full((M,), float("nan"), device=DEV)
    rstd_o = torch.full((M,), float("nan"), device=DEV)
    aux = torch.empty((M, N), device=DEV, dtype=torch.bfloat16) if act != "none" else None
    knob(CONSUMER_KNOBS[kn])
    y = ops.gemm(xb.to(DEV), wf.to(DEV), bias=d.to(DEV), act=act, aux_out=aux, ln_stats=st.to(DEV), ln_c=c.to(DEV), ln_eps=1e-5,
                 ln_save=(mean_o, rstd_o))
    torch.cuda.synchronize()
    mean = x64.mean(1)
    rstd = 1.0 / torch.sqrt(x64.var(1, unbiased=False) + 1e-5)
    assert (mean_o.cpu().double() - mean).abs().max().item() <= 1e-5
    assert ((rstd_o.cpu().double() - rstd).abs() / rstd).max().item() <= 1e-4
    # exact arithmetic of the folded form on the operands as stored
    pre = rstd[:, None] * (xb.double() @ wf.double().T - mean[:, None] * c.double()) + d.double()
    ref = ACTS[act](pre.float()).double()
    got = y.cpu().double()
    tol = 2 ** -8 * ref.abs().max().item() + 1e-3
    assert (got - ref).abs().max().item() <= tol
    if aux is not None:
        assert (aux.cpu().double() - pre).abs().max().item() <= 2 ** -8 * pre.abs().max().item() + 1e-3
    # and it IS LayerNorm -> Linear up to the bf16 roundings of x and W' (2^-8 relative each, summed over K)
    true = ACTS[act]((torch.nn.functional.layer_norm(x64, (K,), gamma.double(), beta.double(), 1e-5) @ w.double().T + bias.double()).float()).double()
    assert (got - true).abs().max().item() <= 2 ** -7 * math.sqrt(K) * 0.05 * 3 + tol


@pytest.mark.parametrize("M,E,F", [(1943, 1280, 5120), (300, 256, 1024), (77, 128, 512)])
def test_chain_matches_the_layernorm_kernel_route(ops, M, E, F):
    """out-projection (+ residual, statistics, copy) -> folded FFN-up, against out-projection -> eavqa_layernorm_fwd -> eavqa_gemm."""
    ctx, wo = rnd(M, E, seed=1, dtype=torch.bfloat16), rnd(E, E, seed=2, scale=0.03, dtype=torch.bfloat16)
    x, bo = rnd(M, E, seed=3), rnd(E, seed=4, scale=0.1)
    w1, b1 = rnd(F, E, seed=5, scale=0.03), rnd(F, seed=6, scale=0.1)
    gamma, beta = 1.0 + rnd(E, seed=7, scale=0.2), rnd(E, seed=8, scale=0.1)
    d = lambda t: t.to(DEV)
    # the route it replaces
    x1 = ops.gemm(d(ctx), d(wo), bias=d(bo), residual=d(x), out_f32=True)
    a2, mean, rstd = ops.layernorm_fwd(x1, d(gamma), d(beta), 1e-5, torch.bfloat16, save_stats=True)
    u_ref = torch.empty((M, F), device=DEV, dtype=torch.bfloat16)
    f_ref = ops.gemm(a2, d(w1.to(torch.bfloat16)), bias=d(b1), act="gelu_new", aux_out=u_ref)
    # folded
    wf = (w1 * gamma).to(torch.bfloat16)
    c, dd = wf.float().sum(1), (w1.double() @ beta.double() + b1.double()).float()
    x1T = torch.empty((M, E), device=DEV, dtype=torch.bfloat16)
    st = torch.empty((M, slots(E), 2), device=DEV, dtype=torch.float32)
    x1b = ops.gemm(d(ctx), d(wo), bias=d(bo), residual=d(x), out_f32=True, copy_out=x1T, stats_out=st)
    mean2, rstd2 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    u = torch.empty((M, F), device=DEV, dtype=torch.bfloat16)
    f = ops.gemm(x1T, d(wf), bias=d(dd), act="gelu_new", aux_out=u, ln_stats=st, ln_c=d(c), ln_eps=1e-5, ln_save=(mean2, rstd2))
    torch.cuda.synchronize()
    assert torch.equal(x1, x1b)                                            # the stream itself is bit-identical
    assert (mean - mean2).abs().max().item() <= 1e-5 and ((rstd - rstd2).abs() / rstd).max().item() <= 1e-4
    scale = u_ref.float().abs().max().item()
    assert (u.float() - u_ref.float()).abs().max().item() <= 2 ** -6 * scale
    assert (f.float() - f_ref.float()).abs().max().item() <= 2 ** -6 * scale
    assert (u.float() - u_ref.float()).abs().mean().item() <= 2 ** -9 * scale


def test_fp32_operands(ops):
    """The exact-fp32 kernel carries the same epilogue (the fp32 parity mode may fold too)."""
    M, N, K = 200, 192, 96
    x = rnd(M, K, seed=1) + 0.3
    gamma, beta, w, bias = 1.0 + rnd(K, seed=2, scale=0.2), rnd(K, seed=3, scale=0.1), rnd(N, K, seed=4, scale=0.1), rnd(N, seed=5)
    wf = w * gamma
    st = torch.zeros((M, 2, 2))
    st[:, 0, 0], st[:, 0, 1] = x.double().sum(1).float(), (x.double() ** 2).sum(1).float()
    y = ops.gemm(x.to(DEV), wf.to(DEV), bias=(w @ beta + bias).to(DEV), ln_stats=st.to(DEV), ln_c=wf.sum(1).to(DEV), ln_eps=1e-5)
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5) @ w.double().T + bias.double()
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-5 * math.sqrt(K)


def test_rejects_bad_arguments(ops, knob):
    from eavqa_amd import _lib
    a, b = torch.zeros((128, 64), device=DEV, dtype=torch.bfloat16), torch.zeros((256, 64), device=DEV, dtype=torch.bfloat16)
    knob(2 << 8)                                                                            # a knob-only tile has no LN form
    with pytest.raises(_lib.EavqaError, match="shape"):
        ops.gemm(a, b, stats_out=torch.zeros((128, 4, 2), device=DEV))
    knob(0)
    with pytest.raises(_lib.EavqaError):
        ops.gemm(a, b, stats_out=torch.zeros((128, 3, 2), device=DEV))                     # fewer than ceil(N / 64) slots
    with pytest.raises(_lib.EavqaError):
        ops.gemm(a, b, ln_stats=torch.zeros((128, 1, 2), device=DEV))                      # no ln_c
    with pytest.raises(_lib.EavqaError):
        ops.gemm(a, b.T.contiguous(), b_kc=False, ln_stats=torch.zeros((128, 1, 2), device=DEV), ln_c=torch.zeros(256, device=DEV))
