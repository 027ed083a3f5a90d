"""GPU: the decode-step GEMM without partial sums (eavqa_gemm_decode, csrc/decode_direct.hip) against a plain torch fp32 / fp64
CPU computation of the same op: bf16 operands (exact in fp64), fp32 accumulation, so the product must agree with the float64 product
of the same bf16 inputs to ~1e-3 sqrt(K) of the operand scale; a bf16 output adds one rounding (2^-9 relative).  The statistics
partials are compared through what their consumer derives from them (mean, variance of every output row).  LayerNorm / RMSNorm on load:
against torch's layer_norm / the T5 formula on the fp32 stream rounded to bf16 the way the standalone LayerNorm kernel rounds it.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle

DEV = "cuda"
ACTS = {"none": lambda x: x, "tanh": torch.tanh, "relu": torch.relu, "gelu_new": oracle.gelu_new, "quick_gelu": oracle.quick_gelu}


@pytest.fixture(scope="module")
def ops():
    from eavqa_amd import ops as _ops, _lib
    assert _lib.load().eavqa_check_device() == 0, "not a gfx950 device"
    return _ops


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def combine(stats: torch.Tensor, N: int, cols: int):
    """(mean, biased variance) of every row from the [M, n, 2] partials, in float64."""
    s = stats.double().cpu()
    n_i = torch.tensor([min(cols, N - i * cols) for i in range(s.shape[1])], dtype=torch.float64)
    mean = s[:, :, 0].sum(1) / N
    m2 = (s[:, :, 1] + n_i * (s[:, :, 0] / n_i - mean[:, None]) ** 2).sum(1)
    return mean, m2 / N


SHAPES = [(32, 2560, 2560), (32, 7680, 2560), (32, 2560, 10240), (1, 192, 64), (5, 200, 128), (17, 1000, 192), (33, 520, 320),
          (64, 2048, 2048), (16, 10240, 2560), (32, 50, 64), (64, 6144, 512), (32, 10240, 2560)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("sel", [0, 0x10, 0x1, 0x2, 0x3, 0x4, 0x201, 0x80, 0x82])
def test_bf16_product_every_tile(ops, M, N, K, sel):
    """Plain product, every fragment count per workgroup (sel bits [3:0]), plain instead of non-temporal weight loads (bit 4), the rows
    split over two workgroups (0x201)."""
    if (sel & 0xF) >= 3 and M > 32:
        pytest.skip("64 rows x 3 / 4 fragments are not instantiated")
    a, b = rnd(M, K, seed=1, dtype=torch.bfloat16), rnd(N, K, seed=2, dtype=torch.bfloat16)
    out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float32)
    ops.gemm_decode(a.to(DEV), b.to(DEV), [out], sel=sel)
    torch.cuda.synchronize()
    ref = (a.double() @ b.double().T).float()
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 1e-3 * math.sqrt(K), err


@pytest.mark.parametrize("M,N,K", [(32, 2560, 2560), (7, 264, 128), (64, 1024, 256), (32, 3 * 80, 192)])
@pytest.mark.parametrize("act", ["none", "relu", "gelu_new", "tanh"])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_epilogue_bias_act_residual_stats(ops, M, N, K, act, out_dtype):
    a, b = rnd(M, K, seed=3, dtype=torch.bfloat16), rnd(N, K, seed=4, scale=0.05, dtype=torch.bfloat16)
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    out = torch.full((M + 1, N + 8), float("nan"), device=DEV, dtype=out_dtype)     # strided rows, guard row / columns stay NaN
    stats = ops.gemm_decode(a.to(DEV), b.to(DEV), [out[:M, :N]], bias=bias.to(DEV), act=act, residual=res.to(DEV), want_stats=True)
    torch.cuda.synchronize()
    ref = ACTS[act]((a.double() @ b.double().T + bias.double()).float()).double() + res.double()
    got = out.cpu()
    assert torch.isnan(got[M]).all() and torch.isnan(got[:, N:]).all()
    tol = 2e-4 * math.sqrt(K) + (2 ** -8 * ref.abs().max().item() if out_dtype == torch.bfloat16 else 0.0)
    assert (got[:M, :N].double() - ref).abs().max().item() <= tol
    # the statistics describe the fp32 values before the output rounding
    cols = ops.gemm_decode_cols(M, N, K)
    assert stats.shape == (M, (N + cols - 1) // cols, 2)
    mean, var = combine(stats, N, cols)
    assert (mean - ref.mean(1)).abs().max().item() <= 1e-4 * math.sqrt(K)
    assert ((var - ref.var(1, unbiased=False)).abs() / ref.var(1, unbiased=False)).max().item() <= 1e-3


def test_three_segments_write_q_and_the_cache_rows(ops):
    """n_seg = 3: q to its buffer, k and v straight into row `pos` of every sample's cache [B, S_max, E] (row stride S_max * E)."""
    B, E, K, S_max, pos = 6, 192, 128, 5, 3
    a, w, bias = rnd(B, K, seed=1, dtype=torch.bfloat16), rnd(3 * E, K, seed=2, dtype=torch.bfloat16), rnd(3 * E, seed=3)
    q = torch.zeros((B, E), device=DEV, dtype=torch.bfloat16)
    kc = torch.zeros((B * S_max, E), device=DEV, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    ops.gemm_decode(a.to(DEV), w.to(DEV), [q, kc.view(B, S_max, E)[:, pos], vc.view(B, S_max, E)[:, pos]], bias=bias.to(DEV))
    torch.cuda.synchronize()
    ref = (a.double() @ w.double().T + bias.double()).float()
    for got, lo in ((q.cpu(), 0), (kc.view(B, S_max, E)[:, pos].cpu(), E), (vc.view(B, S_max, E)[:, pos].cpu(), 2 * E)):
        assert (got.float() - ref[:, lo:lo + E]).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-3
    touched = kc.view(B, S_max, E).abs().sum(-1).cpu() > 0
    assert touched[:, pos].all() and touched.sum().item() == B           # nothing but the new position was written


@pytest.mark.parametrize("M,E,F", [(32, 2560, 10240), (32, 2048, 5120), (9, 192, 320), (64, 512, 1024), (1, 64, 64)])
@pytest.mark.parametrize("norm", ["layer", "rms"])
def test_norm_on_load_from_the_producers_statistics(ops, M, E, F, norm):
    """A chain as in a decoder layer: x1 = x + ctx Wo^T + b (statistics out) -> act(norm(x1) W1^T + b1).  Reference: the norm in
    float64 on the fp32 x1 the kernel produced, rounded to bf16 (what the standalone LayerNorm / RMSNorm kernels hand the GEMM)."""
    ctx, wo = rnd(M, E, seed=1, dtype=torch.bfloat16), rnd(E, E, seed=2, scale=0.03, dtype=torch.bfloat16)
    x, bo = rnd(M, E, seed=3) + 0.3, rnd(E, seed=4, scale=0.1)                   # a mean that is not small against the deviation
    w1, b1 = rnd(F, E, seed=5, scale=0.03, dtype=torch.bfloat16), rnd(F, seed=6, scale=0.1)
    gamma, beta = 1.0 + rnd(E, seed=7, scale=0.2), rnd(E, seed=8, scale=0.1)
    x1 = torch.empty((M, E), device=DEV, dtype=torch.float32)
    st = ops.gemm_decode(ctx.to(DEV), wo.to(DEV), [x1], bias=bo.to(DEV), residual=x.to(DEV), want_stats=True)
    cols = ops.gemm_decode_cols(M, E, E)
    f = torch.empty((M, F), device=DEV, dtype=torch.bfloat16)
    ops.gemm_decode(x1, w1.to(DEV), [f], norm=norm, gamma=gamma.to(DEV), beta=beta.to(DEV) if norm == "layer" else None, eps=1e-5,
                    stats_in=st, stats_in_cols=cols, bias=b1.to(DEV), act="relu")
    torch.cuda.synchronize()
    x1c = x1.cpu().double()
    if norm == "layer":
        a = torch.nn.functional.layer_norm(x1c, (E,), gamma.double(), beta.double(), 1e-5)
    else:
        a = x1c * torch.rsqrt((x1c ** 2).mean(-1, keepdim=True) + 1e-5) * gamma.double()
    a = a.float().to(torch.bfloat16)
    ref = torch.relu(a.double() @ w1.double().T + b1.double())
    err = (f.cpu().double() - ref).abs().max().item()
    # a bf16 rounding of `a` may flip where the two statistics differ in the last bit: one ulp of a (2^-8 |a|) times |w| per flip
    assert err <= 2 ** -7 * max(1.0, ref.abs().max().item()) + 2e-4 * math.sqrt(E), err


@pytest.mark.parametrize("M,E,F", [(32, 2048, 5120), (5, 128, 200), (64, 256, 512)])
@pytest.mark.parametrize("act", ["gelu_new", "relu"])
def test_gated_pairs_t5(ops, M, E, F, act):
    """T5DenseGatedActDense: h = act(x wi_0^T) * (x wi_1^T) from one [2F, E] weight."""
    a, wi = rnd(M, E, seed=1, dtype=torch.bfloat16), rnd(2 * F, E, seed=2, scale=0.05, dtype=torch.bfloat16)
    h = torch.full((M, F), float("nan"), device=DEV, dtype=torch.bfloat16)
    ops.gemm_decode(a.to(DEV), wi.to(DEV), [h], gated=True, act=act)
    torch.cuda.synchronize()
    u = (a.double() @ wi.double().T).float()
    ref = ACTS[act](u[:, :F]).double() * u[:, F:].double()
    assert (h.cpu().double() - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-3


def test_rejects_bad_arguments(ops):
    from eavqa_amd import _lib
    a, b = torch.zeros((4, 96), device=DEV, dtype=torch.bfloat16), torch.zeros((64, 96), device=DEV, dtype=torch.bfloat16)
    out = torch.zeros((4, 64), device=DEV, dtype=torch.float32)
    with pytest.raises(_lib.EavqaError, match="shape"):
        ops.gemm_decode(a, b, [out])                                        # K % 64
    a, b = torch.zeros((65, 64), device=DEV, dtype=torch.bfloat16), torch.zeros((64, 64), device=DEV, dtype=torch.bfloat16)
    with pytest.raises(_lib.EavqaError):
        ops.gemm_decode(a, b, [torch.zeros((65, 64), device=DEV)])          # more than 64 rows
    assert ops.gemm_decode_cols(32, 2560, 96) == 0 and ops.gemm_decode_cols(32, 2560, 2560) == 16
    assert ops.gemm_decode_cols(32, 7680, 2560) == 32 and ops.gemm_decode_cols(32, 10240, 2560, 1) == 48
