"""GPU: the fp8 path of BASELINE configs[4] - row-wise e4m3 quantisation (eavqa_quantize_rows_fp8) and the block-scaled-MFMA GEMM
(eavqa_gemm_fp8) - against torch's OCP float8_e4m3fn on the CPU.

The products of two e4m3 values are exact in fp32 and the kernel accumulates in fp32, so against a float64 reference computed from
the SAME quantised operands the GEMM is held to fp32-accumulation tolerance; small-integer operands are exact."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle

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
@pytest.mark.parametrize("rows,cols", [(1, 128), (7, 4096), (33, 16384), (70, 1280), (3, 12)])
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
