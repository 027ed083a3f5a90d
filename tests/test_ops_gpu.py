"""GPU: every C-ABI kernel against a plain torch fp32/fp64 CPU computation of the same op.

Tolerances (stated per test): the float32 path uses exact-fp32 MFMA / VALU arithmetic and must
agree with an fp64-accumulated reference to ~1e-5 relative; the bfloat16 path stores operands and
outputs in bf16 (8 significant bits) with fp32 accumulation, so it is compared against the same
reference computed from the bf16-rounded inputs with a tolerance of a few bf16 ulps of the output
scale.  Integer / index kernels are bit-exact.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle
from conftest import load_golden


@pytest.fixture(scope="module")
def ops():
    from eavqa_amd import ops as _ops, _lib
    assert _lib.load().eavqa_check_device() == 0, "not a gfx950 device"
    return _ops


DEV = "cuda"


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def tol(dtype, scale=1.0):
    return (2e-5 if dtype == torch.float32 else 2e-2) * scale


ACTS = {
    "none": lambda x: x, "tanh": torch.tanh, "relu": torch.relu, "gelu_new": oracle.gelu_new, "quick_gelu": oracle.quick_gelu,
}


# --------------------------------------------------------------------------- GEMM
@pytest.fixture(params=["fast", "general", "big", "128x80", "128x96", "256x128", "256x160", "256x192"])
def gemm_path(request):
    """bf16 k-contiguous GEMMs take the 128x128 LDS-DMA kernel (K % 32 == 0) or the 256x256 one (K % 64 == 0, chosen by
    shape) or a shaped tile (128x80 ... 256x192, chosen by shape); run every case through the general kernel, the 128x128
    kernel and (forced) the 256x256 kernel and every shaped tile."""
    from eavqa_amd import ops
    big = 2 if request.param == "big" else 1
    shape = {"128x80": 2, "128x96": 3, "256x128": 4, "256x160": 5, "256x192": 6}.get(request.param, 1)
    # the selector travels with every call (eavqa_gemm_ex, include/eavqa_test.h): the library holds no state
    ops.KernelSelect.gemm = (big << 14) | (shape << 18) | (int(request.param == "general") << 7)
    yield request.param
    ops.KernelSelect.gemm = 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (1, 8, 8), (64, 520, 1032), (300, 50257 // 64, 128),
                                   (2688, 1280, 1280), (130, 3000, 32), (257, 129, 96), (300, 700, 192), (1943, 5120, 128),
                                   (32, 2560, 10240), (64, 6400, 512), (17, 100, 64), (1, 7680, 2560)])
def test_gemm_layouts_and_edges(ops, gemm_path, dtype, a_kc, b_kc, M, N, K):
    vec = 8 if dtype == torch.bfloat16 else 4
    if (not a_kc and M % vec) or (not b_kc and N % vec):
        pytest.skip("contiguous dim must be a multiple of the vector width (checked in test_gemm_rejects)")
    a = rnd(M, K, dtype=dtype, seed=1)
    b = rnd(N, K, dtype=dtype, seed=2)   # asymmetric random operands
    ref = (a.double() @ b.double().T).float()
    A = (a if a_kc else a.T.contiguous()).to(DEV)
    Bm = (b if b_kc else b.T.contiguous()).to(DEV)
    out = ops.gemm(A, Bm, a_kc=a_kc, b_kc=b_kc, out_f32=True)
    torch.cuda.synchronize()
    err = (out.cpu() - ref).abs().max().item()
    assert err <= (1e-4 if dtype == torch.float32 else 1e-3) * math.sqrt(K), err  # fp32 accumulate of exact products


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_identity_asymmetric(ops, dtype):
    """A = I with an asymmetric B catches a transposed C write (guide section 3)."""
    n = 128
    a = torch.eye(n, dtype=dtype)
    b = (torch.arange(n)[:, None] * 3 + torch.arange(n)[None, :] * 0.5).to(dtype)  # B[n][k]
    out = ops.gemm(a.to(DEV), b.to(DEV), out_f32=True).cpu()
    assert torch.equal(out, b.float().T)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["none", "tanh", "relu", "gelu_new", "quick_gelu"])
@pytest.mark.parametrize("M", [150, 40])     # 40: the skinny (weight-streaming) kernel in bf16
def test_gemm_epilogue_forward(ops, gemm_path, dtype, act, M):
    N, K = 264, 96
    a, b = rnd(M, K, dtype=dtype, seed=3, scale=0.3), rnd(N, K, dtype=dtype, seed=4, scale=0.3)
    bias = rnd(N, seed=5)
    res = rnd(M, N, seed=6)
    pre = (a.double() @ b.double().T).float() * 0.5 + bias
    ref = ACTS[act](pre) + res
    aux = torch.empty(M, N, dtype=dtype, device=DEV)
    out = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), act=act, aux_out=aux, residual=res.to(DEV), out_f32=True, alpha=0.5)
    assert (out.cpu() - ref).abs().max().item() <= tol(dtype, 0.1 if dtype == torch.float32 else 1.0)
    assert (aux.float().cpu() - pre).abs().max().item() <= tol(dtype, 1.0)
    # storage-dtype output, no residual
    out2 = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), act=act)
    assert out2.dtype == dtype
    assert (out2.float().cpu() - ACTS[act]((a.double() @ b.double().T).float() + bias)).abs().max().item() <= tol(dtype, 2.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["tanh", "relu", "gelu_new", "quick_gelu"])
@pytest.mark.parametrize("M", [70, 33])
def test_gemm_epilogue_activation_backward(ops, dtype, act, M):
    N, K = 136, 64
    a, b = rnd(M, K, dtype=dtype, seed=7, scale=0.3), rnd(N, K, dtype=dtype, seed=8, scale=0.3)
    u = rnd(M, N, dtype=dtype, seed=9)
    uu = u.float().clone().requires_grad_(True)
    ACTS[act](uu).sum().backward()
    ref = (a.double() @ b.double().T).float() * uu.grad
    out = ops.gemm(a.to(DEV), b.to(DEV), act=act, aux_in=u.to(DEV), out_f32=True)
    assert (out.cpu() - ref).abs().max().item() <= tol(dtype, 0.2)


@pytest.mark.parametrize("M,N,K,epi", [(16448, 1024, 128, "res16"), (16448, 3072, 64, "bias"), (41120, 4096, 64, "quick"), (16576, 1024, 128, "aux")])
def test_gemm_ragged_last_tile_row_goes_to_a_second_launch(ops, M, N, K, epi):
    """256 x 256 kernel (forced): M % 256 <= 192 ragged rows are handed to a second, small-tile launch when the problem without them needs
    one round of workgroups less (``big_split_rows``: 65 x 4 / 65 x 12 / 161 x 16 tiles here) - pointers of every row-indexed operand
    (A, C, residual in place, pre-activation out) must be offset alike.  Exact integer data; the knob that keeps the single launch must
    give the same bits."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-2, 3, (M, K), generator=g).to(torch.bfloat16).to(DEV)
    b = torch.randint(-2, 3, (N, K), generator=g).to(torch.bfloat16).to(DEV)
    bias = torch.randint(-3, 4, (N,), generator=g).float().to(DEV)
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 700, M)])             # the first tiles, the boundary of the split, the remainder
    pre = (a[rows].double() @ b.double().T + bias.double()).cpu()
    res0 = torch.randint(-8, 9, (M, N), generator=g).to(torch.float16).to(DEV) if epi == "res16" else None
    outs = []
    for knob in ((2 << 14), (2 << 14) | (1 << 25)):                                  # forced 256 x 256: split allowed / single launch
        ops.KernelSelect.gemm = knob
        try:
            if epi == "res16":
                res = res0.clone()
                want = pre + res[rows].double().cpu()
                out = ops.gemm(a, b, bias=bias, residual=res, out=res)
            elif epi == "aux":
                u = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
                out = ops.gemm(a, b, bias=bias, act="relu", aux_out=u)
                want = pre.clamp(min=0)
                assert torch.equal(u[rows].double().cpu(), pre)
            elif epi == "quick":
                out = ops.gemm(a, b, bias=bias, act="quick_gelu")
                want = None
            else:
                out = ops.gemm(a, b, bias=bias)
                want = pre
        finally:
            ops.KernelSelect.gemm = 0
        if want is not None:
            assert torch.equal(out[rows].double().cpu(), want), epi
        else:
            ref = pre * torch.sigmoid(1.702 * pre)
            assert (out[rows].double().cpu() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
        outs.append(out[rows].clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K", [(70, 264, 96), (300, 1024, 256), (2600, 768, 128), (33, 100, 64)])
def test_gemm_low_precision_residual_stream(ops, gemm_path, M, N, K):
    """EAVQA_GEMM_RESIDUAL_LOWP: the residual (and here the output, aliasing it) in bf16 - the frozen CLIP tower's stream.  Exact
    data: small integers, so bf16 holds every product sum and the sum with the residual without rounding."""
    g = torch.Generator().manual_seed(M + N)
    a = torch.randint(-2, 3, (M, K), generator=g).to(torch.bfloat16)
    b = torch.randint(-2, 3, (N, K), generator=g).to(torch.bfloat16)
    bias = torch.randint(-3, 4, (N,), generator=g).float()
    res = torch.randint(-8, 9, (M, N), generator=g).to(torch.bfloat16)
    ref = a.double() @ b.double().T + bias.double() + res.double()
    assert ref.abs().max().item() <= 256                                     # exactly representable in bf16
    buf = res.to(DEV).clone()
    out = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), residual=buf, out=buf)      # in place: x += a @ w^T + b
    assert out.dtype == torch.bfloat16 and torch.equal(out.double().cpu(), ref)
    # the same stream in IEEE half (EAVQA_GEMM_STREAM_F16; what the CLIP tower uses by default): exact on these integers too
    buf16 = res.to(torch.float16).to(DEV)
    out16 = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), residual=buf16, out=buf16)
    assert out16.dtype == torch.float16 and torch.equal(out16.double().cpu(), ref)
    y = ops.layernorm_fwd(out16, None, None, 1e-5, torch.bfloat16)                       # a half stream normalised into a bf16 operand
    want = torch.nn.functional.layer_norm(ref.float(), (N,))
    assert (y.float().cpu() - want).abs().max().item() <= 3e-2
    y16 = ops.layernorm_fwd(out.float(), None, None, 1e-5, torch.float16)                # float32 -> half (the tower's pre-LayerNorm)
    assert y16.dtype == torch.float16 and (y16.float().cpu() - want).abs().max().item() <= 4e-3
    # a strided (column-sliced) bf16 residual with a separate fp32 output, and random data against float64
    wide = rnd(M, N + 24, dtype=torch.bfloat16, seed=7).to(DEV)
    a2, b2 = rnd(M, K, dtype=torch.bfloat16, seed=8, scale=0.3), rnd(N, K, dtype=torch.bfloat16, seed=9, scale=0.3)
    out2 = ops.gemm(a2.to(DEV), b2.to(DEV), residual=wide[:, 8:8 + N], out_f32=True, act="quick_gelu")
    pre = a2.double() @ b2.double().T
    ref2 = pre * torch.sigmoid(1.702 * pre) + wide[:, 8:8 + N].double().cpu()
    assert (out2.double().cpu() - ref2).abs().max().item() <= 2e-2


def test_gemm_residual_alias_accumulates(ops):
    """wgrad accumulation: residual aliases the float32 output."""
    a, b = rnd(64, 32, seed=1), rnd(40, 32, seed=2)
    acc = rnd(64, 40, seed=3).to(DEV)
    ref = acc.cpu() + a @ b.T
    ops.gemm(a.to(DEV), b.to(DEV), residual=acc, out=acc)
    assert torch.allclose(acc.cpu(), ref, atol=1e-5)


def test_gemm_rejects_bad_arguments(ops):
    from eavqa_amd._lib import EavqaError
    a = torch.zeros(16, 12, device=DEV, dtype=torch.bfloat16)   # K = 12 not a multiple of 8
    b = torch.zeros(16, 12, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(EavqaError):
        ops.gemm(a, b)
    with pytest.raises(EavqaError):
        ops.gemm(torch.zeros(4, 8), torch.zeros(4, 8))          # CPU tensors: no fallback


def test_gemm_large_bf16_statistical(ops, gemm_path):
    """A real-shape bf16 GEMM (GPT-2-large c_fc on one batch): relative Frobenius error vs fp64."""
    M, N, K = 2688, 5120, 1280
    a, b = rnd(M, K, dtype=torch.bfloat16, seed=11), rnd(N, K, dtype=torch.bfloat16, seed=12, scale=0.02)
    out = ops.gemm(a.to(DEV), b.to(DEV), out_f32=True).cpu()
    ref = (a.float() @ b.float().T)
    rel = (out - ref).norm() / ref.norm()
    assert rel < 1e-5, rel   # products exact in fp32, only summation order differs


K64_NAMES = ["s128x80l2", "s128x80l4", "s128x128l4", "s256x128l4", "s256x160l4", "s128x80n4", "s128x96", "s256x192", "s256x256", "s128x128n3",
             "s128x256"]      # the loader / consumer specialised tiles of csrc/gemm_k64.hip (knob ids 2..12)


@pytest.mark.parametrize("shape_id", range(len(K64_NAMES)), ids=K64_NAMES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1, 64, 64), (333, 1000, 320), (1943, 1280, 1280), (257, 5120, 128), (70, 200, 3840),
                                   (2050, 90, 192), (512, 512, 64 * 37)])
def test_gemm_full_line_family(ops, shape_id, M, N, K):
    """Every tile shape of the BK = 64 full-cache-line family (csrc/gemm_k64.hip), forced through eavqa_gemm_ex, on ragged M / N,
    one-step and many-step K: raw products against fp64, then the full epilogue (bias, activation, aux_out, residual, bf16 out)."""
    a = rnd(M, K, dtype=torch.bfloat16, seed=21)
    b = rnd(N, K, dtype=torch.bfloat16, seed=22)
    ref = (a.double() @ b.double().T).float()
    ops.KernelSelect.gemm = (2 + shape_id) << 8
    try:
        out = ops.gemm(a.to(DEV), b.to(DEV), out_f32=True)
        bias, res = rnd(N, seed=23), rnd(M, N, seed=24)
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        out2 = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), act="gelu_new", aux_out=aux, residual=res.to(DEV), out_f32=True, alpha=0.25)
        out3 = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), act="relu")
        # activation backward at aux_in (the dgrad of an FFN-up) and a half residual stream written in place
        u = rnd(M, N, dtype=torch.bfloat16, seed=25)
        out4 = ops.gemm(a.to(DEV), b.to(DEV), act="quick_gelu", aux_in=u.to(DEV))
        stream = rnd(M, N, seed=26).to(torch.float16).to(DEV)
        want5 = ref + bias + stream.float().cpu()
        out5 = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), residual=stream, out=stream)
        torch.cuda.synchronize()
    finally:
        ops.KernelSelect.gemm = 0
    uf = u.float()
    sg = torch.sigmoid(1.702 * uf)
    want4 = ref * (sg * (1 + 1.702 * uf * (1 - sg)))
    assert (out4.float().cpu() - want4).abs().max().item() <= 1.5e-2 * max(1.0, want4.abs().max().item())
    assert out5.dtype == torch.float16 and (out5.float().cpu() - want5).abs().max().item() <= 2e-3 * max(1.0, want5.abs().max().item()) + 1e-3 * math.sqrt(K)
    assert (out.cpu() - ref).abs().max().item() <= 1e-3 * math.sqrt(K)
    pre = ref * 0.25 + bias
    assert (aux.float().cpu() - pre).abs().max().item() <= 2e-2 * max(1.0, pre.abs().max().item())
    assert (out2.cpu() - (oracle.gelu_new(pre) + res)).abs().max().item() <= 1e-3 * math.sqrt(K)
    want3 = torch.relu(ref + bias)
    assert out3.dtype == torch.bfloat16 and (out3.float().cpu() - want3).abs().max().item() <= 1e-2 * max(1.0, want3.abs().max().item())


def test_gemm_full_line_family_is_the_default_dispatch(ops):
    """With no selector a k-contiguous bf16 GEMM with K % 64 == 0 takes the full-line family; results equal the forced tile's
    bit for bit (same kernel), whichever tile the cost model picks."""
    a, b = rnd(1943, 1280, dtype=torch.bfloat16, seed=31).to(DEV), rnd(1280, 1280, dtype=torch.bfloat16, seed=32).to(DEV)
    auto = ops.gemm(a, b, out_f32=True)
    same = []
    for i in range(len(K64_NAMES)):
        ops.KernelSelect.gemm = (2 + i) << 8
        same.append(bool(torch.equal(auto, ops.gemm(a, b, out_f32=True))))
    ops.KernelSelect.gemm = 0
    assert any(same), same


# --------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(5, 64), (37, 768), (130, 1280), (3, 4096), (2, 8192)])
def test_layernorm_forward_backward(ops, dtype, rows, cols):
    x = rnd(rows, cols, seed=1) * 2 + 0.5
    g, b = rnd(cols, seed=2) * 0.2 + 1, rnd(cols, seed=3) * 0.1
    dy = rnd(rows, cols, dtype=dtype, seed=4)
    xx = x.clone().requires_grad_(True)
    gg = g.clone().requires_grad_(True)
    bb = b.clone().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(xx, (cols,), gg, bb, 1e-5)
    y_ref.backward(dy.float())
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, dtype, save_stats=True)
    assert (y.float().cpu() - y_ref.detach()).abs().max().item() <= (1e-5 if dtype == torch.float32 else 3e-2)
    assert torch.allclose(mean.cpu(), x.mean(-1), atol=1e-5)
    if cols > 4096:
        return  # backward supports cols <= 4096 (documented)
    dres = rnd(rows, cols, seed=5)
    dgamma = torch.zeros(cols, device=DEV)
    dbeta = torch.zeros(cols, device=DEV)
    dx = ops.layernorm_bwd(x.to(DEV), dy.to(DEV), g.to(DEV), mean, rstd, dres=dres.to(DEV), dgamma=dgamma, dbeta=dbeta)
    assert (dx.cpu() - (dres + xx.grad)).abs().max().item() <= 2e-5 * max(1.0, xx.grad.abs().max().item())
    assert torch.allclose(dgamma.cpu(), gg.grad, atol=1e-4, rtol=1e-4)
    assert torch.allclose(dbeta.cpu(), bb.grad, atol=1e-4, rtol=1e-4)


def test_layernorm_bf16_input_rows_with_stride(ops):
    x = rnd(9, 256, dtype=torch.bfloat16, seed=1).to(DEV)
    view = x[:, :128]    # ld = 256, cols = 128
    y = ops.layernorm_fwd(view, None, None, 1e-5, torch.bfloat16)
    ref = torch.nn.functional.layer_norm(view.float().cpu(), (128,))
    assert (y.float().cpu() - ref).abs().max().item() <= 3e-2


# --------------------------------------------------------------------------- attention
@pytest.fixture(params=["mfma", "mfma_split", "valu"])
def attn_path(request):
    """bf16 attention with hd in {64,80,96,128} runs on the matrix cores (one-tile problems take the fused backward,
    "mfma_split" forces the dQ + dK/dV pair); run every case through all of them and the vector-ALU kernels."""
    from eavqa_amd import ops
    ops.KernelSelect.attention = {"mfma": 0, "mfma_split": 2, "valu": 1}[request.param]
    yield request.param
    ops.KernelSelect.attention = 0


@pytest.mark.parametrize("B,H,hd,Sk,S_max", [(32, 32, 80, 151, 160), (5, 12, 64, 1, 8), (33, 20, 64, 97, 120), (64, 32, 128, 200, 256), (3, 6, 96, 40, 40)])
def test_attention_decode_appends_and_attends(ops, B, H, hd, Sk, S_max):
    """eavqa_attention_decode: the new K / V rows come from the QKV projection's output (strided views of one [B, 3E] buffer), are
    written to position Sk-1 of the cache and attended with the positions already there; ragged key masks; other cache rows untouched."""
    E = H * hd
    g = torch.Generator().manual_seed(B + Sk)
    kc = torch.randn(B, S_max, E, generator=g).to(torch.bfloat16)
    vc = torch.randn(B, S_max, E, generator=g).to(torch.bfloat16)
    qkv = torch.randn(B, 3 * E, generator=g).to(torch.bfloat16)
    mask = (torch.rand(B, S_max, generator=g) > 0.2).int()
    mask[:, Sk - 1] = 1
    scale = hd ** -0.5
    kc_d, vc_d, qkv_d = kc.to(DEV), vc.to(DEV), qkv.to(DEV)
    out = ops.attention_decode(qkv_d[:, :E], kc_d.view(B * S_max, E), vc_d.view(B * S_max, E), qkv_d[:, E:2 * E], qkv_d[:, 2 * E:], B, H, Sk, hd,
                               kv_batch_rows=S_max, key_mask=mask.to(DEV), ld_mask=S_max, scale=scale)
    k_full, v_full = kc.clone(), vc.clone()
    k_full[:, Sk - 1], v_full[:, Sk - 1] = qkv[:, E:2 * E], qkv[:, 2 * E:]
    assert torch.equal(kc_d.cpu(), k_full) and torch.equal(vc_d.cpu(), v_full)          # appended, nothing else touched
    q = qkv[:, :E].float().view(B, H, 1, hd)
    k = k_full[:, :Sk].float().view(B, Sk, H, hd).transpose(1, 2)
    v = v_full[:, :Sk].float().view(B, Sk, H, hd).transpose(1, 2)
    sc = (q @ k.transpose(-1, -2)) * scale
    sc = sc.masked_fill(mask[:, None, None, :Sk] == 0, torch.finfo(torch.float32).min)
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B, E)
    assert (out.float().cpu() - ref).abs().max().item() <= 3e-2


def attn_ref(q, k, v, key_mask, causal, scale):
    """q [B,Sq,H,hd] etc. float64 reference with the oracle's masking (finfo.min add)."""
    B, Sq, H, hd = q.shape
    Sk = k.shape[1]
    s = torch.einsum("bihd,bjhd->bhij", q, k) * scale
    keep = torch.ones(B, 1, Sq, Sk, dtype=torch.bool)
    if key_mask is not None:
        keep = keep & (key_mask != 0)[:, None, None, :]
    if causal:
        i = torch.arange(Sq)[:, None]
        j = torch.arange(Sk)[None, :]
        keep = keep & (j <= i + (Sk - Sq))[None, None]
    s = torch.where(keep, s, torch.full_like(s, torch.finfo(torch.float32).min))
    p = torch.softmax(s, dim=-1)
    return torch.einsum("bhij,bjhd->bihd", p, v)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Sq,Sk,hd,causal,masked", [
    (2, 3, 42, 42, 64, True, True),      # LM shape, ragged right padding
    (1, 2, 50, 50, 64, False, False),    # ViT-B/32
    (2, 2, 150, 150, 80, True, True),    # few-shot prompt, OPT-2.7B head dim
    (1, 1, 130, 130, 128, True, False),  # > 2 key tiles
    (2, 8, 7, 7, 8, False, False),       # mapper transformer (tiny)
    (1, 2, 20, 20, 96, False, False),    # mapper on GPT-2 (E/8 = 96)
    (2, 2, 1, 33, 64, True, True),       # decode step against a cache
    (1, 1, 9, 9, 160, False, False),
    (1, 4, 257, 257, 64, False, False),  # ViT-L/14
    (1, 2, 577, 577, 64, False, False),  # ViT-L/14@336px: what the reference's stored embeddings use (base_env.jsonnet:39-40)
    (1, 2, 577, 577, 64, False, True),   # ... with a ragged key mask (10 query / key tiles of 64, the last one 1 row deep)
    (2, 2, 70, 70, 96, True, True),      # mfma path, 3 k-steps
    (1, 2, 3, 200, 128, True, False),    # few queries against a long cache, two query-tile-free key tiles
    (2, 1, 100, 100, 80, False, True),   # non-causal: 8 waves per workgroup (one 128-query tile)
    (1, 2, 72, 72, 64, False, True),     # non-causal: 5 waves per workgroup
    (2, 8, 64, 64, 512, False, False),   # transformer mapper on OPT-6.7B (cfg5): 4096 / 8 heads, clip_length + prefix_length = 64
    (2, 8, 20, 20, 160, False, False),   # transformer mapper on GPT-2-large: 1280 / 8 heads (wide-head one-tile kernels, 2 chunks)
    (1, 2, 37, 37, 256, True, True),     # wide heads, causal + key mask
    (1, 2, 33, 64, 320, True, False),    # wide heads, Sq < Sk
])
def test_attention_forward_backward(ops, attn_path, dtype, B, H, Sq, Sk, hd, causal, masked):
    E = H * hd
    q, k, v = (rnd(B, Sq, H, hd, dtype=dtype, seed=1), rnd(B, Sk, H, hd, dtype=dtype, seed=2), rnd(B, Sk, H, hd, dtype=dtype, seed=3))
    do = rnd(B, Sq, H, hd, dtype=dtype, seed=4)
    km = None
    if masked:
        lens = torch.tensor([Sk - 3 * i - 1 for i in range(B)]).clamp(min=1)
        km = (torch.arange(Sk)[None] < lens[:, None]).int()
    scale = hd ** -0.5
    qq, kk, vv = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    ref = attn_ref(qq, kk, vv, km, causal, scale)
    ref.backward(do.double())
    # packed qkv rows like the LM's c_attn output: [B*S, 3E] when Sq == Sk
    if Sq == Sk:
        qkv = torch.cat([q.reshape(B * Sq, E), k.reshape(B * Sk, E), v.reshape(B * Sk, E)], dim=1).to(DEV)
        Q, K, V = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    else:
        Q, K, V = q.reshape(B * Sq, E).to(DEV), k.reshape(B * Sk, E).to(DEV), v.reshape(B * Sk, E).to(DEV)
    kmd = km.to(DEV) if km is not None else None
    o, lse = ops.attention_fwd(Q, K, V, B, H, Sq, Sk, hd, key_mask=kmd, causal=causal, scale=scale, save_lse=True)
    t = 1e-5 if dtype == torch.float32 else 2e-2
    assert (o.float().cpu().reshape(B, Sq, H, hd) - ref.detach().float()).abs().max().item() <= t
    dq, dk, dv = ops.attention_bwd(Q, K, V, o, do.reshape(B * Sq, E).to(DEV), lse, B, H, Sq, Sk, hd, key_mask=kmd, causal=causal, scale=scale)
    tb = 5e-5 if dtype == torch.float32 else 6e-2
    for got, want in ((dq, qq.grad), (dk, kk.grad), (dv, vv.grad)):
        assert (got.float().cpu().reshape(want.shape) - want.float()).abs().max().item() <= tb * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("B,H,N", [(3, 4, 257), (2, 3, 577), (2, 2, 50), (1, 2, 65), (1, 3, 97), (1, 2, 592), (2, 1, 33), (1, 1, 1), (5, 2, 288), (1, 2, 321)])
def test_attention_resident_kv_kernel_bf16(ops, B, H, N):
    """The K / V-resident forward (eavqa_attn_mfma::fwd_resident64_kernel: the CLIP tower's attention, hd 64, no mask) against the
    float64 reference and against the streamed-tile kernel on the same fused [B*N, 3E] qkv rows: every tail shape (N = 32 k + 1,
    32 k, one tile, two workgroup-size classes, a lone last query block riding on wave 0), output and log-sum-exp."""
    hd = 64
    E = H * hd
    q, k, v = (rnd(B, N, H, hd, dtype=torch.bfloat16, seed=s) for s in (1, 2, 3))
    ref = attn_ref(q.double(), k.double(), v.double(), None, False, hd ** -0.5)
    sc = torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) * hd ** -0.5
    lse_ref = torch.logsumexp(sc, -1)                                           # [B, H, N]
    qkv = torch.cat([q.reshape(B * N, E), k.reshape(B * N, E), v.reshape(B * N, E)], dim=1).to(DEV)
    Q, K, V = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    outs = {}
    for name, path in (("resident", 8), ("streamed", 4)):
        ops.KernelSelect.attention = path
        try:
            o, lse = ops.attention_fwd(Q, K, V, B, H, N, N, hd, causal=False, scale=hd ** -0.5, save_lse=True)
        finally:
            ops.KernelSelect.attention = 0
        outs[name] = o.float().cpu()
        assert (outs[name].reshape(B, N, H, hd) - ref.float()).abs().max().item() <= 2e-2, name
        assert (lse.cpu().double() - lse_ref).abs().max().item() <= 2e-2, name
    assert (outs["resident"] - outs["streamed"]).abs().max().item() <= 2e-2
    # exact data: one-hot attention (a huge score on one key per query) must copy that key's V row bit for bit
    sel = torch.randint(0, N, (B, N), generator=torch.Generator().manual_seed(9))
    kk = torch.zeros(B, N, H, hd)
    kk[..., 0] = torch.arange(N)[None, :, None].float() / 4.0                   # key j carries j / 4 in feature 0 (exact in bf16 up to 1024 / 4)
    # query i asks for key sel[i]: score = 64 * (2 t k - k^2) / 4 ... peaked at k = t; built directly as a two-feature parabola
    qq = torch.zeros(B, N, H, hd)
    t = sel[..., None].float() / 4.0
    qq[..., 0] = 2.0 * t * 64.0
    qq[..., 1] = -64.0
    kk[..., 1] = kk[..., 0] ** 2
    if N <= 512:                                                                 # (j / 4)^2 stays exact in bf16 only for small j
        vv = rnd(B, N, H, hd, dtype=torch.bfloat16, seed=5)
        qkv2 = torch.cat([qq.bfloat16().reshape(B * N, E), kk.bfloat16().reshape(B * N, E), vv.reshape(B * N, E)], dim=1).to(DEV)
        ops.KernelSelect.attention = 8
        try:
            o2 = ops.attention_fwd(qkv2[:, :E], qkv2[:, E:2 * E], qkv2[:, 2 * E:], B, H, N, N, hd, causal=False, scale=1.0)
        finally:
            ops.KernelSelect.attention = 0
        refq = attn_ref(qq.bfloat16().double(), kk.bfloat16().double(), vv.double(), None, False, 1.0)
        # (P is rounded to bf16 before P.V and the output to bf16: relative to the largest |V| a peaked row copies)
        assert (o2.float().cpu().reshape(B, N, H, hd) - refq.float()).abs().max().item() <= 1e-2 * max(1.0, refq.abs().max().item())


@pytest.mark.parametrize("B,H,lens,causal", [(4, 3, [42, 18, 33, 1], True), (3, 2, [64, 49, 16], True), (2, 2, [17, 31], False), (5, 1, [48, 47, 32, 15, 2], True),
                                              (1, 20, [42], True)])
def test_attention_one_tile_backward_swizzled_hd64(ops, B, H, lens, causal):
    """eavqa_attn_mfma::bwd_fused64_kernel (swizzled images of 16 ceil(S / 16) rows, the training step's attention backward) against the
    float64 reference and the round-2 padded-pitch kernel (path bit 2) on packed sequences of every fragment count (1 .. 64 items)."""
    hd = 64
    E = H * hd
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    M, S = int(cu[-1]), max(lens)
    qkv = rnd(M, 3 * E, dtype=torch.bfloat16, seed=11)
    do = rnd(M, E, dtype=torch.bfloat16, seed=12)
    scale = hd ** -0.5
    want = []
    for b in range(B):
        sl = slice(int(cu[b]), int(cu[b + 1]))
        q, k, v = (qkv[sl, i * E:(i + 1) * E].double().reshape(1, -1, H, hd).clone().requires_grad_(True) for i in range(3))
        o = attn_ref(q, k, v, None, causal, scale)
        o.backward(do[sl].double().reshape(1, -1, H, hd))
        want.append(tuple(t.grad.reshape(-1, E) for t in (q, k, v)))
    want = [torch.cat([w[i] for w in want]) for i in range(3)]
    Q, K, V = (qkv[:, i * E:(i + 1) * E].to(DEV) for i in range(3))
    o, lse = ops.attention_fwd(Q, K, V, B, H, S, S, hd, causal=causal, scale=scale, save_lse=True, cu_seqlens=cu.to(DEV))
    got = {}
    for name, path in (("swizzled", 0), ("padded", 4)):
        ops.KernelSelect.attention = path
        try:
            got[name] = ops.attention_bwd(Q, K, V, o, do.to(DEV), lse, B, H, S, S, hd, causal=causal, scale=scale, cu_seqlens=cu.to(DEV))
        finally:
            ops.KernelSelect.attention = 0
        for g_, w in zip(got[name], want):
            assert (g_.float().cpu() - w.float()).abs().max().item() <= 6e-2 * max(1.0, w.abs().max().item()), name
    for a, b_ in zip(got["swizzled"], got["padded"]):
        assert (a.float() - b_.float()).abs().max().item() <= 2e-2 * max(1.0, b_.float().abs().max().item())


@pytest.mark.parametrize("B,H,S,hd", [(2, 2, 42, 200), (1, 3, 64, 640), (2, 1, 5, 136)])
def test_attention_wide_heads_ragged_chunk_bf16(ops, B, H, S, hd):
    """Head dims the vector-ALU kernels do not cover (GPT-2-xl mapper 1600 / 8 = 200, OPT-13B 5120 / 8 = 640): bf16 only, the last
    128-wide chunk of the head dim partly empty."""
    E = H * hd
    q, k, v, do = (rnd(B, S, H, hd, dtype=torch.bfloat16, seed=i) for i in (1, 2, 3, 4))
    qq, kk, vv = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    ref = attn_ref(qq, kk, vv, None, False, hd ** -0.5)
    ref.backward(do.double())
    Q, K, V = (t.reshape(B * S, E).to(DEV) for t in (q, k, v))
    o, lse = ops.attention_fwd(Q, K, V, B, H, S, S, hd, causal=False, scale=hd ** -0.5, save_lse=True)
    assert (o.float().cpu().reshape(B, S, H, hd) - ref.detach().float()).abs().max().item() <= 2e-2
    dq, dk, dv = ops.attention_bwd(Q, K, V, o, do.reshape(B * S, E).to(DEV), lse, B, H, S, S, hd, causal=False, scale=hd ** -0.5)
    for got, want in ((dq, qq.grad), (dk, kk.grad), (dv, vv.grad)):
        assert (got.float().cpu().reshape(want.shape) - want.float()).abs().max().item() <= 6e-2 * max(1.0, want.abs().max().item())


def test_attention_fully_masked_row_is_finite(ops):
    B, H, S, hd = 1, 1, 5, 16
    q = rnd(B * S, H * hd, seed=1).to(DEV)
    km = torch.zeros(B, S, dtype=torch.int32, device=DEV)
    o = ops.attention_fwd(q, q, q, B, H, S, S, hd, key_mask=km, causal=False, scale=1.0)
    assert torch.isfinite(o).all()
    assert torch.allclose(o.cpu(), q.cpu().mean(0, keepdim=True).expand(S, -1), atol=1e-5)  # uniform average


def test_attention_kv_cache_batch_stride(ops):
    B, H, hd, Smax, t = 2, 2, 64, 16, 11
    E = H * hd
    cache_k, cache_v = rnd(B, Smax, E, seed=1).to(DEV), rnd(B, Smax, E, seed=2).to(DEV)
    q = rnd(B, 1, E, seed=3).to(DEV)
    o = ops.attention_fwd(q.view(B, E), cache_k.view(B * Smax, E), cache_v.view(B * Smax, E), B, H, 1, t, hd,
                          causal=True, scale=hd ** -0.5, kv_batch_rows=Smax)
    ref = attn_ref(q.cpu().double().view(B, 1, H, hd), cache_k.cpu().double()[:, :t].reshape(B, t, H, hd),
                   cache_v.cpu().double()[:, :t].reshape(B, t, H, hd), None, True, hd ** -0.5)
    assert (o.cpu().view(B, 1, H, hd) - ref.float()).abs().max().item() <= 1e-5


@pytest.mark.parametrize("B,H,hd,Sk,S_max,ks", [(32, 32, 80, 151, 160, 4), (5, 12, 64, 1, 8, 1), (33, 20, 64, 97, 120, 3), (64, 32, 128, 200, 256, 2),
                                                 (3, 6, 96, 161, 170, 5)])
def test_attention_decode_from_qkv_partial_sums(ops, B, H, hd, Sk, S_max, ks):
    """eavqa_attention_decode_splitk == eavqa_splitk_finish (q | K row | V row) followed by eavqa_attention_decode, bit for bit: output,
    and both caches."""
    E = H * hd
    g = torch.Generator().manual_seed(B + Sk)
    kc = torch.randn(B, S_max, E, generator=g).to(torch.bfloat16)
    vc = torch.randn(B, S_max, E, generator=g).to(torch.bfloat16)
    part = torch.randn(ks, B, 3 * E, generator=g).to(DEV)
    bias = torch.randn(3 * E, generator=g).to(DEV)
    mask = (torch.rand(B, S_max, generator=g) > 0.2).int()
    mask[:, Sk - 1] = 1
    scale = hd ** -0.5
    # the three-kernel route
    qkv = torch.empty(B, 3 * E, dtype=torch.bfloat16, device=DEV)
    ops.splitk_finish(part, [qkv], bias=bias)
    k1, v1 = kc.to(DEV), vc.to(DEV)
    want = ops.attention_decode(qkv[:, :E], k1.view(B * S_max, E), v1.view(B * S_max, E), qkv[:, E:2 * E], qkv[:, 2 * E:], B, H, Sk, hd,
                                kv_batch_rows=S_max, key_mask=mask.to(DEV), ld_mask=S_max, scale=scale)
    k2, v2 = kc.to(DEV), vc.to(DEV)
    got = ops.attention_decode_splitk(part, bias, k2.view(B * S_max, E), v2.view(B * S_max, E), B, H, Sk, hd, kv_batch_rows=S_max,
                                      key_mask=mask.to(DEV), ld_mask=S_max, scale=scale)
    assert torch.equal(k1, k2) and torch.equal(v1, v2)
    assert torch.equal(k2[:, Sk - 1], qkv[:, E:2 * E]) and torch.equal(v2[:, Sk - 1], qkv[:, 2 * E:])
    assert torch.equal(got, want)


@pytest.mark.parametrize("B,H,Sk,hd,masked", [(3, 5, 157, 80, True), (2, 8, 300, 128, False), (2, 4, 40, 64, True), (1, 3, 7, 32, False),
                                              (32, 32, 160, 80, True), (1, 4, 3584, 64, False), (1, 4, 3600, 64, False)])
@pytest.mark.parametrize("path", [0, 16, 32])
def test_attention_decode_step(ops, B, H, Sk, hd, masked, path):
    """Sq = 1 in bf16 takes the decode kernel (4 heads per workgroup, K/V streamed once): cache with a batch stride, ragged
    masks, H not a multiple of 4.  path 16 keeps the round-3 kernels (V image in LDS / V behind the scores) where round 4's
    all-in-registers kernel would be chosen."""
    ops.KernelSelect.attention = path
    try:
        _attention_decode_step(ops, B, H, Sk, hd, masked)
    finally:
        ops.KernelSelect.attention = 0


def _attention_decode_step(ops, B, H, Sk, hd, masked):
    E, Smax = H * hd, Sk + 9
    ck, cv = rnd(B, Smax, E, dtype=torch.bfloat16, seed=1), rnd(B, Smax, E, dtype=torch.bfloat16, seed=2)
    q = rnd(B, E, dtype=torch.bfloat16, seed=3)
    km = None
    if masked:
        lens = torch.tensor([max(1, Sk - 5 * i) for i in range(B)])
        km = torch.zeros(B, Smax, dtype=torch.int32)
        km[:, :Sk] = (torch.arange(Sk)[None] < lens[:, None]).int()
        km[:, Sk - 1] = 1            # the new token itself is always visible
    o, lse = ops.attention_fwd(q.to(DEV), ck.view(B * Smax, E).to(DEV), cv.view(B * Smax, E).to(DEV), B, H, 1, Sk, hd,
                               key_mask=km.to(DEV) if masked else None, ld_mask=Smax if masked else 0, causal=True, scale=hd ** -0.5,
                               kv_batch_rows=Smax, save_lse=True)
    kk, vv = ck.double()[:, :Sk].reshape(B, Sk, H, hd), cv.double()[:, :Sk].reshape(B, Sk, H, hd)
    ref = attn_ref(q.double().view(B, 1, H, hd), kk, vv, km[:, :Sk] if masked else None, True, hd ** -0.5)
    assert (o.float().cpu().view(B, 1, H, hd) - ref.float()).abs().max().item() <= 2e-2
    s = torch.einsum("bhd,bjhd->bhj", q.double().view(B, H, hd), kk) * hd ** -0.5
    if masked:
        s = torch.where(km[:, None, :Sk] != 0, s, torch.full_like(s, torch.finfo(torch.float32).min))
    assert (lse.cpu().view(B, H) - torch.logsumexp(s, -1).float()).abs().max().item() <= 1e-3


# --------------------------------------------------------------------------- sequence assembly (bit-exact)
@pytest.mark.parametrize("pos_mode", [0, 1])
def test_build_prefix_rows_exact(ops, pos_mode):
    g = torch.Generator().manual_seed(3)
    B, T, L = 5, 70, 10
    tok = torch.randint(0, 50257, (B, T), generator=g)
    lens = torch.tensor([70, 1, 33, 64, 65])
    qm = (torch.arange(T)[None] < lens[:, None]).long()
    src, msk, pos = ops.build_prefix_rows(tok.to(DEV), qm.to(DEV), L, pos_mode)
    S = L + T
    exp_src = torch.cat([-(1 + torch.arange(B * L).view(B, L)), tok], dim=1).int()
    exp_msk = torch.cat([torch.ones(B, L, dtype=torch.long), qm], dim=1)
    assert torch.equal(src.cpu(), exp_src)
    assert torch.equal(msk.cpu().long(), exp_msk)
    if pos_mode == 0:   # HF:gpt2 :571-574
        exp_pos = torch.arange(S).expand(B, S)
    else:               # HF:opt :45-70
        exp_pos = (torch.cumsum(exp_msk, 1) * exp_msk - 1) + 2
    assert torch.equal(pos.cpu().long(), exp_pos)


@pytest.mark.parametrize("case", ["z", "f", "s"])
def test_build_fewshot_rows_and_embed_match_reference(ops, case):
    """insert_prefix_into_input: golden vectors from vct0_test.py and the reference's own output."""
    z = load_golden("insert_prefix.npz")
    if case == "s":
        L, E, shots = [int(v) for v in z["s_cfg"]]
    else:
        L, E, shots = 2, 3, (0 if case == "z" else 2)
    tok = torch.from_numpy(z[f"{case}_tok"])
    qm = torch.from_numpy(z[f"{case}_mask"])
    text = torch.from_numpy(z[f"{case}_text"])
    pp = torch.from_numpy(z[f"{case}_pp"])
    B, T = tok.shape
    src, msk, pos, status = ops.build_fewshot_rows(tok.to(DEV), qm.to(DEV), L, shots + 1, 32099, 0)
    assert status.cpu().tolist() == [shots + 1] * B
    assert torch.equal(msk.cpu().long(), torch.from_numpy(z[f"{case}_out_mask"]))
    # the embedding gather itself: fake vocabulary = per-position text embeddings (E padded to 4)
    Ep = 8
    wte = torch.zeros(B * T, Ep)
    wte[:, :E] = text.reshape(B * T, E)
    # remap token ids to their flat (b,t) index so wte[id] is that position's embedding
    flat_ids = torch.arange(B * T).view(B, T)
    is_sent = (tok <= 32099) & (tok > 32099 - (shots + 1))
    tok2 = torch.where(is_sent, tok, flat_ids + 40000)   # keep sentinels, move text ids out of the way
    src2, _, _, _ = ops.build_fewshot_rows(tok2.to(DEV), qm.to(DEV), L, shots + 1, 32099, 0)
    src2 = torch.where(src2 >= 0, src2 - 40000, src2)
    prefix = torch.zeros(B * (shots + 1) * L, Ep)
    prefix[:, :E] = pp.reshape(-1, E)
    x = ops.embed_assemble(src2.contiguous(), None, wte.to(DEV), prefix.to(DEV), None)
    got = x.cpu().view(B, -1, Ep)[:, :, :E]
    assert torch.equal(got, torch.from_numpy(z[f"{case}_emb"]))


def test_build_fewshot_rows_reports_wrong_sentinel_count(ops):
    tok = torch.tensor([[32099, 5, 6, 7], [5, 6, 7, 8]])
    qm = torch.ones_like(tok)
    _, _, _, status = ops.build_fewshot_rows(tok.to(DEV), qm.to(DEV), 3, 1, 32099, 0)
    assert status.cpu().tolist() == [1, 0]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_assemble_and_backward(ops, dtype):
    g = torch.Generator().manual_seed(1)
    B, L, T, E, V = 3, 4, 9, 64, 100
    wte, wpe = rnd(V, E, dtype=dtype, seed=1), rnd(32, E, dtype=dtype, seed=2)
    prefix = rnd(B * L, E, dtype=dtype, seed=3)
    tok = torch.randint(0, V, (B, T), generator=g)
    qm = torch.ones(B, T, dtype=torch.long)
    src, msk, pos = ops.build_prefix_rows(tok.to(DEV), qm.to(DEV), L, 0)
    x = ops.embed_assemble(src, pos, wte.to(DEV), prefix.to(DEV), wpe.to(DEV))
    ref = torch.cat([prefix.view(B, L, E).float(), wte.float()[tok]], dim=1) + wpe.float()[: L + T][None]
    assert torch.equal(x.cpu().view(B, L + T, E), ref)
    dx = rnd(B * (L + T), E, seed=4)
    dp = ops.embed_assemble_bwd(src, dx.to(DEV), B * L, dtype)
    assert torch.equal(dp.float().cpu().view(B, L, E), dx.view(B, L + T, E)[:, :L].to(dtype).float())


def test_build_labels_exact(ops):
    z = load_golden("label_mask.npz")
    ids = torch.from_numpy(z["input_ids"])
    pad, bos = int(z["pad_id"]), int(z["bos_id"])
    out = ops.build_labels(ids.to(DEV), 3, pad, bos, mode=0).cpu()
    assert torch.equal(out[:, :3], torch.full((ids.shape[0], 3), -100))
    assert out[:, 3:].tolist() == z["labels"].tolist()
    # random rows (longer than one wave) against the oracle's loop-for-loop restatement
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 12, (16, 150), generator=g)
    ids[3, 100:] = pad
    ids[4] = pad
    ids[5, :] = 3
    for mode, want in ((0, oracle.label_mask_vqa(ids, pad, bos)), (1, oracle.label_mask_cc(ids, pad))):
        got = ops.build_labels(ids.to(DEV), 0, pad, bos, mode=mode).cpu()
        assert torch.equal(got, want)


# --------------------------------------------------------------------------- loss / pick
@pytest.mark.parametrize("V,ld", [(50257, 50264), (320, 320), (7, 8)])
def test_cross_entropy_forward_backward(ops, V, ld):
    B, S = 3, 11
    logits = torch.zeros(B * S, ld)
    logits[:, :V] = rnd(B * S, V, seed=1) * 3
    g = torch.Generator().manual_seed(2)
    labels = torch.randint(0, V, (B, S), generator=g)
    labels[0, :4] = -100
    labels[2, 7:] = -100
    lg = logits[:, :V].view(B, S, V).clone().requires_grad_(True)
    ref = oracle.causal_lm_loss(lg, labels)
    ref.backward()
    loss, count, row_lse = ops.ce_fwd(logits.to(DEV), labels.to(DEV), V)
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    gs = torch.ones(1, device=DEV)
    d = ops.ce_bwd(logits.to(DEV), labels.to(DEV), V, row_lse, count, gs, torch.float32, ld)
    assert torch.allclose(d.cpu()[:, :V].view(B, S, V), lg.grad, atol=1e-7)
    assert (d.cpu()[:, V:] == 0).all()
    db = ops.ce_bwd(logits.to(DEV), labels.to(DEV), V, row_lse, count, gs, torch.bfloat16, ld)
    assert torch.allclose(db.float().cpu()[:, :V].view(B, S, V), lg.grad, atol=1e-3, rtol=1e-2)


def test_cross_entropy_all_ignored_is_nan_like_torch(ops):
    logits = rnd(4, 8, seed=1).to(DEV)
    labels = torch.full((2, 2), -100).to(DEV)
    loss, count, _ = ops.ce_fwd(logits, labels, 8)
    assert count.item() == 0 and math.isnan(loss.item())


def test_greedy_pick_first_max_and_finished_rows(ops):
    V, ld, B = 1000, 1000, 4
    logits = rnd(B, ld, seed=1)
    logits[0, 17] = 50.0
    logits[0, 900] = 50.0          # tie: first index wins (torch.argmax)
    logits[1, 999] = 60.0
    logits[2, 5] = 70.0
    logits[3, 0] = 80.0
    raw = torch.empty(B, dtype=torch.int32, device=DEV)
    toks = torch.zeros(B, 6, dtype=torch.int64, device=DEV)
    unf = torch.tensor([1, 1, 0, 1], dtype=torch.int32, device=DEV)
    alive = torch.zeros(3, dtype=torch.int32, device=DEV)
    ops.greedy_pick(logits.to(DEV), V, 42, 999, raw, toks[:, 2], unf, any_unfinished=alive[0:1])
    assert raw.cpu().tolist() == [17, 999, 5, 0]
    assert toks[:, 2].cpu().tolist() == [17, 999, 42, 0]     # finished row emits pad (clipcap.py:431-434)
    assert unf.cpu().tolist() == [1, 0, 0, 1]                # row 1 just produced eos (clipcap.py:458-461)
    # the early-stop flag (clipcap.py:463): some row unfinished -> 1; every row finished (eos = what rows 0 and 3 emit) -> stays 0
    unf2 = torch.tensor([1, 0, 0, 0], dtype=torch.int32, device=DEV)
    ops.greedy_pick(logits.to(DEV), V, 42, 17, raw, toks[:, 4], unf2, any_unfinished=alive[1:2])
    assert alive.cpu().tolist() == [1, 0, 0] and unf2.cpu().tolist() == [0, 0, 0, 0]
    ops.greedy_pick(logits.to(DEV), V, 42, None, raw, toks[:, 3], unf)   # eos None: raw tokens, flags untouched
    lp = torch.empty(logits.shape[0], device=DEV)
    ops.greedy_pick(logits.to(DEV), V, 42, None, raw, toks[:, 3], unf, lp)
    want = torch.log_softmax(logits.double(), -1).max(-1).values.float()
    assert (lp.cpu() - want).abs().max().item() <= 1e-5
    assert toks[:, 3].cpu().tolist() == [17, 999, 5, 0] and unf.cpu().tolist() == [1, 0, 0, 1]


# --------------------------------------------------------------------------- optimiser
@pytest.mark.parametrize("n", [4096, 1003])
def test_adamw_matches_torch(ops, n):
    p = rnd(n, seed=1)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-3)
    P, M, Vv = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for step in range(1, 5):
        g = rnd(n, seed=10 + step)
        ref.grad = g.clone() * 0.5
        opt.step()
        ops.adamw(P, g.to(DEV), M, Vv, step, 1e-3, grad_scale=0.5, shadow=shadow)
    assert torch.allclose(P.cpu(), ref.detach(), atol=2e-7, rtol=1e-6)
    assert torch.equal(shadow.cpu(), P.cpu().to(torch.bfloat16))


# --------------------------------------------------------------------------- ViT front end
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("img,ps", [(64, 32), (42, 14)])
def test_patchify_and_vit_assemble(ops, dtype, img, ps):
    B, W = 2, 64
    px = rnd(B, 3, img, img, seed=1)
    g = img // ps
    K = 3 * ps * ps
    ldp = (K + 7) // 8 * 8
    pt = ops.patchify(px.to(DEV), ps, dtype, ldp).cpu()
    ref = px.reshape(B, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, K)
    assert torch.equal(pt[:, :K].float(), ref.to(dtype).float())
    assert (pt[:, K:] == 0).all()
    pe = rnd(B * g * g, W, dtype=dtype, seed=2)
    cls, pos = rnd(W, seed=3), rnd(g * g + 1, W, seed=4)
    x = ops.vit_assemble(pe.to(DEV), cls.to(DEV), pos.to(DEV), B, g * g).cpu().view(B, g * g + 1, W)
    ref = torch.cat([cls.expand(B, 1, W), pe.float().view(B, g * g, W)], dim=1) + pos[None]
    assert torch.equal(x, ref)


# --------------------------------------------------------------------------- packed rows
@pytest.mark.parametrize("dtype,hd", [(torch.float32, 16), (torch.bfloat16, 64), (torch.bfloat16, 80)])
def test_row_plan_and_packed_attention_match_padded(ops, dtype, hd):
    g = torch.Generator().manual_seed(7)
    B, S, H = 4, 37, 2
    E = H * hd
    lens = torch.tensor([37, 5, 20, 1])
    mask = (torch.arange(S)[None] < lens[:, None]).int()
    src = torch.randint(0, 100, (B, S), generator=g).int()
    pos = torch.arange(S).expand(B, S).int().contiguous()
    labels = torch.randint(0, 50, (B, S), generator=g)
    cu, src_r, pos_r, lab_r, flat = ops.build_row_plan(mask.to(DEV), labels.to(DEV), src.to(DEV), pos.to(DEV), True)
    M = int(lens.sum())
    assert cu.cpu().tolist() == [0, 37, 42, 62, 63]
    keep = mask.bool().flatten()
    exp_flat = torch.arange(B * S)[keep]
    assert torch.equal(flat.cpu()[:M].long(), exp_flat)
    assert torch.equal(src_r.cpu()[:M], src.flatten()[keep]) and torch.equal(pos_r.cpu()[:M], pos.flatten()[keep])
    shifted = torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:]
    assert torch.equal(lab_r.cpu()[:M], shifted.flatten()[keep])
    # identity plan
    cu0, src0, _, lab0, flat0 = ops.build_row_plan(mask.to(DEV), labels.to(DEV), src.to(DEV), pos.to(DEV), False)
    assert cu0.cpu().tolist() == [0, 37, 74, 111, 148] and torch.equal(flat0.cpu().long(), torch.arange(B * S))
    assert torch.equal(lab0.cpu(), shifted.flatten())
    # packed attention == padded attention on the kept rows (forward and backward)
    qkv = rnd(B * S, 3 * E, dtype=dtype, seed=3).to(DEV)
    do = rnd(B * S, E, dtype=dtype, seed=4).to(DEV)
    t = 1e-5 if dtype == torch.float32 else 2e-2
    o, lse = ops.attention_fwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], B, H, S, S, hd, key_mask=mask.to(DEV), causal=True,
                               scale=0.25, save_lse=True)
    dq, dk, dv = ops.attention_bwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], o, do, lse, B, H, S, S, hd,
                                   key_mask=mask.to(DEV), causal=True, scale=0.25)
    idx = flat[:M].long()
    qp = qkv[idx].contiguous()
    dop = do[idx].contiguous()
    op, lsep = ops.attention_fwd(qp[:, :E], qp[:, E:2 * E], qp[:, 2 * E:], B, H, S, S, hd, causal=True, scale=0.25,
                                 save_lse=True, cu_seqlens=cu)
    assert torch.allclose(op.float(), o[idx].float(), atol=t * 0.1)
    dqp, dkp, dvp = ops.attention_bwd(qp[:, :E], qp[:, E:2 * E], qp[:, 2 * E:], op, dop, lsep, B, H, S, S, hd, causal=True,
                                      scale=0.25, cu_seqlens=cu)
    # gradients flowing from padded QUERY rows do not exist in the packed run: zero them in the padded reference
    do_masked = do.clone()
    do_masked[~keep.to(DEV)] = 0
    dq2, dk2, dv2 = ops.attention_bwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], o, do_masked, lse, B, H, S, S, hd,
                                      key_mask=mask.to(DEV), causal=True, scale=0.25)
    for a, b_ in ((dqp, dq2), (dkp, dk2), (dvp, dv2)):
        assert torch.allclose(a.float(), b_[idx].float(), atol=t, rtol=t)


def test_cross_entropy_with_row_labels(ops):
    V, rows = 50, 9
    logits = (rnd(rows, 56, seed=1) * 2).to(DEV)
    lab = torch.tensor([3, -100, 7, 49, -100, 0, 1, 2, -100]).to(DEV)
    loss, count, row_lse = ops.ce_fwd(logits, lab, V)
    ref = torch.nn.functional.cross_entropy(logits.cpu()[:, :V], lab.cpu(), ignore_index=-100)
    assert abs(loss.item() - ref.item()) <= 1e-5 and count.item() == 6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(64, 64), (130, 70), (1, 5), (12800, 640), (132, 68), (4, 4), (2048, 4096), (100, 8)])
def test_transpose(ops, dtype, rows, cols):
    x = rnd(rows, cols, dtype=dtype, seed=1).to(DEV)
    assert torch.equal(ops.transpose(x).cpu(), x.cpu().T)
    # a column window of a wider buffer (row stride > cols), into a wider destination
    wide = rnd(rows, cols + 12, dtype=dtype, seed=2).to(DEV)
    out = torch.zeros(cols, rows + 8, dtype=dtype, device=DEV)
    ops.transpose(wide[:, 4:4 + cols], out=out[:, :rows])
    assert torch.equal(out[:, :rows].cpu(), wide[:, 4:4 + cols].cpu().T) and bool((out[:, rows:] == 0).all())


# --------------------------------------------------------------------------- decode-step primitives
@pytest.mark.parametrize("M,N,K", [(32, 7680, 2560), (32, 2560, 10240), (1, 64, 32), (17, 200, 96), (64, 12800, 512), (5, 50272 // 8, 768)])
def test_gemm_splitk_and_finish(ops, M, N, K):
    a = rnd(M, K, dtype=torch.bfloat16, seed=1, scale=0.5)
    b = rnd(N, K, dtype=torch.bfloat16, seed=2, scale=0.5)
    ref = a.double() @ b.double().T
    part = ops.gemm_splitk(a.to(DEV), b.to(DEV))
    assert part.shape[1:] == (M, N)
    got = part.double().sum(0).cpu()
    assert (got - ref).abs().max().item() <= 1e-3 * math.sqrt(K)
    if N % 12 == 0:
        bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
        # three bf16 segments with different row strides (q buffer + two "cache" destinations)
        o0 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)[:, :N // 3]
        o1 = torch.zeros(M, 2 * (N // 3), dtype=torch.bfloat16, device=DEV)[:, :N // 3]
        o2 = torch.zeros(M, N // 3, dtype=torch.bfloat16, device=DEV)
        ops.splitk_finish(part, [o0, o1, o2], bias=bias.to(DEV), act="relu")
        want = torch.relu(ref.float() + bias)
        cat = torch.cat([o0, o1, o2], 1).float().cpu()
        assert (cat - want).abs().max().item() <= 2e-2 * max(1.0, want.abs().max().item())
        x = torch.empty(M, N, device=DEV)
        ops.splitk_finish(part, [x], bias=bias.to(DEV), residual=res.to(DEV))
        assert (x.cpu() - (ref.float() + bias + res)).abs().max().item() <= 1e-3 * math.sqrt(K)


@pytest.mark.parametrize("sel", [8, 16, 0x0808, 0x0810, 0x2408, 0x2410, 0x2808, 0x2810])
@pytest.mark.parametrize("M,N,K", [(32, 7680, 2560), (9, 200, 96), (64, 2560, 1024)])
def test_gemm_splitk_every_kernel_variant(ops, sel, M, N, K):
    """eavqa_gemm_splitk_ex: load-window depth 8 / 16 x 4 / 8 waves per workgroup x 16 / 32 columns per wave (bits of `sel`)."""
    a = rnd(M, K, dtype=torch.bfloat16, seed=1, scale=0.5)
    b = rnd(N, K, dtype=torch.bfloat16, seed=2, scale=0.5)
    ref = a.double() @ b.double().T
    part = ops.gemm_splitk(a.to(DEV), b.to(DEV), unroll=sel)
    assert (part.double().sum(0).cpu() - ref).abs().max().item() <= 1e-3 * math.sqrt(K)


def test_gemm_splitk_is_deterministic(ops):
    a = rnd(32, 2560, dtype=torch.bfloat16, seed=1).to(DEV)
    b = rnd(2560, 2560, dtype=torch.bfloat16, seed=2).to(DEV)
    p0 = ops.gemm_splitk(a, b)
    for _ in range(3):
        assert torch.equal(ops.gemm_splitk(a, b), p0)


@pytest.mark.parametrize("rows,cols,ks", [(32, 2560, 8), (3, 768, 0), (64, 4096, 2), (7, 136, 3)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_splitk(ops, dtype, rows, cols, ks):
    x = rnd(rows, cols, seed=1)
    g, b, bias = rnd(cols, seed=2) + 1.0, rnd(cols, seed=3), rnd(cols, seed=4)
    part = rnd(ks, rows, cols, seed=5) if ks else None
    xs = x + (bias + part.sum(0) if ks else 0.0)
    want = torch.nn.functional.layer_norm(xs, (cols,), g, b, 1e-5)
    xo = torch.empty(rows, cols, device=DEV)
    y = ops.layernorm_splitk(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, dtype, part=part.to(DEV) if ks else None,
                             bias=bias.to(DEV) if ks else None, x_out=xo)
    assert (xo.cpu() - xs).abs().max().item() <= 1e-5
    assert (y.float().cpu() - want).abs().max().item() <= tol(dtype, 1.0) * max(1.0, want.abs().max().item())


def test_select_gather_scatter_rows(ops):
    g = torch.Generator().manual_seed(4)
    M, E = 3001, 64
    labels = torch.where(torch.rand(M, generator=g) < 0.6, torch.randint(0, 500, (M,), generator=g), torch.full((M,), -100))
    want = torch.nonzero(labels >= 0).flatten()
    idx, lab, cnt = ops.select_rows(labels.to(DEV), len(want) + 5)          # over-estimated capacity: the tail stays (0, -100)
    assert cnt.item() == len(want)
    assert torch.equal(idx.cpu()[:len(want)].long(), want) and torch.equal(lab.cpu()[:len(want)], labels[want])
    assert (idx.cpu()[len(want):] == 0).all() and (lab.cpu()[len(want):] == -100).all()
    idx2, _, cnt2 = ops.select_rows(labels.to(DEV), 10)                     # capacity below the count: first 10 only, count is still exact
    assert torch.equal(idx2.cpu().long(), want[:10]) and cnt2.item() == len(want)
    for dtype in (torch.float32, torch.bfloat16):
        x = rnd(M, E, dtype=dtype, seed=5).to(DEV)
        sel = idx[:len(want)].contiguous()
        y = ops.gather_rows(x, sel)
        assert torch.equal(y.cpu(), x.cpu()[want])
        back = ops.scatter_rows(y, sel, M)
        ref = torch.zeros(M, E, dtype=dtype); ref[want] = x.cpu()[want]
        assert torch.equal(back.cpu(), ref)
