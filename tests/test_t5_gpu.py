"""GPU: the T5 / T0 path (VCT0Model, src/models/vct0.py:301-549) - kernels against float64 torch, the model against the fixtures the
REFERENCE's own VCT0Prefix produced on tiny local T5 checkpoints (tests/golden/vct0_*.npz) and against the CPU oracle."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle
from _metrics import grad_stats
from conftest import load_golden

DEV = "cuda"


def rnd(*shape, seed=0, dtype=torch.float32, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(dtype)


@pytest.fixture(scope="module")
def ops():
    from eavqa_amd import ops as o
    return o


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(5, 64), (37, 512), (130, 2048), (3, 4096)])
def test_rmsnorm_forward_backward(ops, dtype, rows, cols):
    x = rnd(rows, cols, seed=1) * 2 + 0.3
    g = rnd(cols, seed=2) * 0.2 + 1
    dy = rnd(rows, cols, seed=3, dtype=dtype)
    res = rnd(rows, cols, seed=4)
    xx = x.double().clone().requires_grad_(True)
    y_ref = oracle.t5_rms_norm(xx, g.double(), 1e-6)
    y_ref.backward(dy.double())
    y, rstd = ops.rmsnorm_fwd(x.to(DEV), g.to(DEV), 1e-6, dtype, save_stats=True)
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    assert (y.double().cpu() - y_ref.detach()).abs().max().item() <= tol * max(1.0, y_ref.abs().max().item())
    lowp = torch.empty(rows, cols, dtype=dtype, device=DEV)
    dx = ops.rmsnorm_bwd(x.to(DEV), dy.to(DEV), g.to(DEV), rstd, dres=res.to(DEV), lowp_out=lowp)
    want = xx.grad + res.double()
    assert (dx.double().cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    assert (lowp.double().cpu() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["gelu_new", "relu"])
def test_gated_activation_forward_backward(ops, dtype, act):
    rows, F = 37, 136
    u = rnd(rows, 2 * F, seed=1, dtype=dtype)
    dh = rnd(rows, F, seed=2, dtype=dtype)
    uu = u.double().clone().requires_grad_(True)
    f = oracle.gelu_new if act == "gelu_new" else torch.relu
    h_ref = f(uu[:, :F]) * uu[:, F:]
    h_ref.backward(dh.double())
    h = ops.gated_act_fwd(u.to(DEV), act)
    du = ops.gated_act_bwd(u.to(DEV), dh.to(DEV), act)
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    assert (h.double().cpu() - h_ref.detach()).abs().max().item() <= tol * max(1.0, h_ref.abs().max().item())
    assert (du.double().cpu() - uu.grad).abs().max().item() <= tol * max(1.0, uu.grad.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Sq,Sk,hd,causal,bidir,masked", [(2, 4, 10, 10, 16, False, True, False), (3, 2, 37, 37, 64, False, True, True),
                                                               (2, 3, 9, 9, 64, True, False, False), (2, 2, 1, 14, 16, True, False, False),
                                                               (2, 2, 7, 150, 64, False, None, True), (1, 2, 130, 130, 32, False, True, True),
                                                               (2, 4, 150, 150, 64, False, True, True), (2, 4, 1, 14, 64, True, False, False),
                                                               (2, 2, 5, 5, 80, True, False, False), (1, 2, 70, 70, 128, True, False, True)])
def test_attention_with_relative_position_bias(ops, dtype, B, H, Sq, Sk, hd, causal, bidir, masked):
    """eavqa_attention_fwd_rel / _bwd_rel against float64: encoder (bidirectional buckets), decoder (causal, one-sided buckets, incl. a cached
    step Sq = 1 < Sk), cross-attention (no bias, key mask)."""
    from eavqa_amd.models.t5 import relative_bucket
    E = H * hd
    q, k, v, do = rnd(B, Sq, H, hd, seed=1, dtype=dtype), rnd(B, Sk, H, hd, seed=2, dtype=dtype), rnd(B, Sk, H, hd, seed=3, dtype=dtype), rnd(B, Sq, H, hd, seed=4, dtype=dtype)
    km = None
    if masked:
        lens = torch.tensor([Sk - 2 * i - 1 for i in range(B)]).clamp(min=1)
        km = (torch.arange(Sk)[None] < lens[:, None]).int()
    table = rnd(32, H, seed=5) * 0.7
    rel = zero = None
    bias = torch.zeros(1, H, Sq, Sk, dtype=torch.float64)
    if bidir is not None:
        bias = oracle.t5_position_bias(table, Sq, Sk, bidir, q_offset=Sk - Sq).double()
        off = torch.arange(-(Sk - 1), Sk)
        rel = table[relative_bucket(off, bidir, 32, 128)].T.contiguous().to(DEV)      # [H, 2 Sk - 1]
        zero = Sk - 1
    keep = torch.ones(B, 1, Sq, Sk, dtype=torch.bool)
    if km is not None:
        keep = keep & (km != 0)[:, None, None, :]
    if causal:
        keep = keep & (torch.arange(Sk)[None, :] <= torch.arange(Sq)[:, None] + (Sk - Sq))[None, None]
    qq, kk, vv = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bihd,bjhd->bhij", qq, kk) + bias
    s = torch.where(keep, s, torch.full_like(s, torch.finfo(torch.float32).min))
    ref = torch.einsum("bhij,bjhd->bihd", torch.softmax(s, -1), vv)
    ref.backward(do.double())
    Q, K, V = q.reshape(B * Sq, E).to(DEV), k.reshape(B * Sk, E).to(DEV), v.reshape(B * Sk, E).to(DEV)
    kmd = km.to(DEV) if km is not None else None
    o, lse = ops.attention_fwd_rel(Q, K, V, B, H, Sq, Sk, hd, rel_bias=rel, rel_zero=zero or 0, key_mask=kmd, causal=causal, scale=1.0, save_lse=True)
    t = 2e-5 if dtype == torch.float32 else 2e-2
    assert (o.double().cpu().reshape(B, Sq, H, hd) - ref.detach()).abs().max().item() <= t
    dq, dk, dv = ops.attention_bwd_rel(Q, K, V, o, do.reshape(B * Sq, E).to(DEV), lse, B, H, Sq, Sk, hd, rel_bias=rel, rel_zero=zero or 0, key_mask=kmd,
                                       causal=causal, scale=1.0)
    tb = 1e-4 if dtype == torch.float32 else 6e-2
    for got, want in ((dq, qq.grad), (dk, kk.grad), (dv, vv.grad)):
        assert (got.double().cpu().reshape(want.shape) - want).abs().max().item() <= tb * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols,ks", [(32, 2048, 8), (5, 64, 1), (64, 512, 10), (1, 4096, 0), (17, 1024, 3)])
def test_rmsnorm_splitk_adds_the_partial_sums_first(ops, dtype, rows, cols, ks):
    """eavqa_rmsnorm_splitk: x = x_in + sum_s P[s] (the new residual stream, float32); y = T5LayerNorm(x)."""
    x_in, gamma = rnd(rows, cols, seed=1) + 0.2, 1.0 + rnd(cols, seed=2, scale=0.2)
    part = rnd(max(ks, 1), rows, cols, seed=3, scale=0.5)
    x_ref = x_in.double() + (part.double().sum(0) if ks else 0.0)
    y_ref = x_ref * torch.rsqrt((x_ref ** 2).mean(-1, keepdim=True) + 1e-6) * gamma.double()
    x_out = torch.full((rows, cols), float("nan"), device=DEV)
    y = ops.rmsnorm_splitk(x_in.to(DEV), gamma.to(DEV), 1e-6, dtype, part=part.to(DEV) if ks else None, x_out=x_out)
    assert (x_out.double().cpu() - x_ref).abs().max().item() <= 1e-5 * max(1.0, x_ref.abs().max().item())
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (y.double().cpu() - y_ref).abs().max().item() <= tol * max(1.0, y_ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["gelu_new", "relu"])
def test_splitk_finish_gated(ops, dtype, act):
    M, F, ks = 9, 520, 4
    part = rnd(ks, M, 2 * F, seed=1)
    u = part.double().sum(0)
    f = oracle.gelu_new if act == "gelu_new" else torch.relu
    ref = f(u[:, :F].float()).double() * u[:, F:]
    h = ops.splitk_finish_gated(part.to(DEV), act, dtype)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (h.double().cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,H,Sk,hd,cross,masked", [(2, 4, 1, 64, False, False), (3, 2, 9, 64, False, False), (32, 32, 10, 64, False, False),
                                                    (2, 4, 150, 64, True, True), (32, 32, 170, 64, True, True), (2, 2, 14, 16, False, False),
                                                    (2, 2, 37, 80, True, False), (2, 2, 20, 128, False, False)])
def test_attention_decode_from_partial_sums_with_bias(ops, B, H, Sk, hd, cross, masked):
    """eavqa_attention_decode_splitk_rel against eavqa_attention_fwd_rel on the q / K / V the partial sums add up to: a decoder self-attention
    step (q | k | v partial sums, K / V appended at row Sk - 1, relative-position bias) and a cross-attention step (q alone, key mask)."""
    from eavqa_amd.models.t5 import relative_bucket
    bf = torch.bfloat16
    I, ks, S_max = H * hd, 3, Sk + 2
    cols = I if cross else 3 * I
    part = rnd(ks, B, cols, seed=1, scale=0.6)
    full = part.sum(0).to(bf)                                         # what a finish pass would store (fp32 sum in slice order, one rounding)
    kc, vc = rnd(B * S_max, I, seed=2, dtype=bf).to(DEV), rnd(B * S_max, I, seed=3, dtype=bf).to(DEV)
    km = None
    if masked:
        lens = torch.tensor([max(1, Sk - 3 * i - 1) for i in range(B)])
        km = (torch.arange(Sk)[None] < lens[:, None]).int().to(DEV)
    rel = zero = None
    if not cross:
        table = rnd(32, H, seed=5) * 0.7
        off = torch.arange(-(Sk + 4), Sk + 5)                         # a table wider than this step needs (one table per generation)
        rel, zero = table[relative_bucket(off, False, 32, 128)].T.contiguous().to(DEV), Sk + 4
    kr, vr = kc.clone(), vc.clone()
    if not cross:
        kr.view(B, S_max, I)[:, Sk - 1] = full[:, I:2 * I].to(DEV)
        vr.view(B, S_max, I)[:, Sk - 1] = full[:, 2 * I:].to(DEV)
    q = full[:, :I].contiguous().to(DEV)
    ref = ops.attention_fwd_rel(q, kr, vr, B, H, 1, Sk, hd, rel_bias=rel, rel_zero=zero or 0, key_mask=km, causal=not cross, scale=1.0,
                                q_batch_rows=1, kv_batch_rows=S_max)
    got = ops.attention_decode_splitk_rel(part.to(DEV), kc, vc, B, H, Sk, hd, kv_batch_rows=S_max, key_mask=km, scale=1.0, rel_bias=rel,
                                          rel_zero=zero or 0)
    torch.cuda.synchronize()
    assert (got.float() - ref.float()).abs().max().item() <= 2e-2
    if cross:
        assert torch.equal(kc, kr) and torch.equal(vc, vr)                       # nothing appended
    else:
        assert torch.equal(kc.view(B, S_max, I)[:, Sk - 1], kr.view(B, S_max, I)[:, Sk - 1])       # the appended rows, bit for bit
        assert torch.equal(vc.view(B, S_max, I)[:, Sk - 1], vr.view(B, S_max, I)[:, Sk - 1])
        assert torch.equal(kc.view(B, S_max, I)[:, :Sk - 1], kr.view(B, S_max, I)[:, :Sk - 1])


def _model(tag, dtype):
    from eavqa_amd.models.t5 import FrozenT5, T5Config
    from eavqa_amd.models.vct0 import VCT0Prefix
    z = load_golden(f"vct0_{tag}.npz")
    T = lambda a: torch.from_numpy(a)
    V, E, DKV, H, F, NL, L, D, gated, tied = [int(v) for v in z["cfg"]]
    sd = {k[3:]: T(v) for k, v in z.items() if k.startswith("lm.")}
    cfg = T5Config(E, DKV, H, F, NL, NL, V, bool(gated), bool(tied))
    lm = FrozenT5(cfg, sd, dtype, DEV)
    model = VCT0Prefix(prefix_length=L, prefix_size=D, mapping_type="mlp", lm=lm, dtype=dtype, device=DEV).train()
    model.clip_project.load_state_dict({k[4:]: T(v) for k, v in z.items() if k.startswith("map.")})
    return z, T, model, V


@pytest.mark.parametrize("tag", ["t0", "t5v10"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2)])
def test_vct0_training_step_matches_reference(tag, dtype, tol):
    """``VCT0Prefix.forward(prefix, labels)`` + backward into the mapper (the CC training step, vct0_exector.py:143-146): loss, logits and
    mapper gradients against the reference's own outputs."""
    z, T, model, V = _model(tag, dtype)
    out = model(prefix=T(z["prefix"]), labels=T(z["labels"]))
    out.loss.backward()
    assert abs(out.loss.item() - float(z["loss"])) <= tol
    assert (out.logits.float().cpu() - T(z["logits"])).abs().max().item() <= tol * max(1.0, float(abs(z["logits"]).max()))
    got = {k: p.grad.float().cpu() for k, p in model.clip_project.named_parameters()}
    want = {k: T(z["gmap." + k]) for k in got}
    cos, ratio, maxrel = grad_stats(got, want)
    print(f"[{tag} {dtype}] |d loss| {abs(out.loss.item() - float(z['loss'])):.2e}  gradient cosine {cos:.6f}  norm ratio {ratio:.5f}  max rel {maxrel:.2e}")
    # measured on MI355X: fp32 cosine 1.000000 / max rel 4e-6; bf16 cosine 0.9992 (gated gelu) / 0.9970 (ReLU: bf16 pre-activations flip
    # derivatives at the kink, as in the OPT tests), norm ratio 0.989 / 1.008, max rel 5e-2 / 1e-1
    assert maxrel <= (1e-3 if dtype == torch.float32 else 0.15)
    assert cos >= (0.999999 if dtype == torch.float32 else 0.995) and abs(ratio - 1) <= (1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("tag", ["t0", "t5v10"])
def test_vct0_generate_paths_match_reference_ids(tag):
    """All four paths of ``VCT0Model.generate`` in fp32: ids equal to the reference's (HF greedy search) wherever the reference's own top-2
    gap exceeds 1e-3, per-step scores within 1e-3."""
    z, T, model, V = _model(tag, torch.float32)
    model.eval()
    special = V - 1
    kw = dict(max_length=9, output_scores=True, return_dict_in_generate=True)
    runs = {
        "prefix": model.generate(prefix=T(z["prefix"]), **kw),
        "fs": model.generate(prefix=T(z["fs_prefix"]), question_tokens=T(z["fs_tokens"]), question_mask=T(z["fs_mask"]), special_token_id=special, **kw),
        "one": model.generate(prefix=T(z["fs_prefix"]), question_tokens=T(z["one_tokens"]), question_mask=T(z["one_mask"]), special_token_id=special,
                              pass_examples_through_encoder_one_at_a_time=True, **kw),
        "text": model.generate(prefix=T(z["fs_prefix"]), question_tokens=T(z["fs_tokens"]), question_mask=T(z["fs_mask"]), no_prefix=True, **kw),
    }
    for name, o in runs.items():
        want_ids, want_scores = T(z[f"gen_{name}_ids"]), T(z[f"gen_{name}_scores"])
        got = torch.stack(list(o.scores))
        assert got.shape == want_scores.shape, (name, got.shape, want_scores.shape)
        assert (got - want_scores).abs().max().item() <= 1e-3, name
        top2 = want_scores.topk(2, dim=-1).values
        decisive = ((top2[..., 0] - top2[..., 1]) > 1e-3).all().item()
        if decisive:
            assert torch.equal(o.sequences, want_ids), (name, o.sequences, want_ids)
        else:
            assert o.sequences.shape == want_ids.shape
    # bf16: same shapes, scores within the bf16 tolerance of this depth
    z, T, model, V = _model(tag, torch.bfloat16)
    model.eval()
    o = model.generate(prefix=T(z["fs_prefix"]), question_tokens=T(z["fs_tokens"]), question_mask=T(z["fs_mask"]), special_token_id=V - 1, **kw)
    want = T(z["gen_fs_scores"])
    n = min(len(o.scores), want.shape[0])
    assert (torch.stack(list(o.scores))[:1] - want[:1]).abs().max().item() <= 8e-2 * max(1.0, want.abs().max().item())     # first step: same decoder prefix


@pytest.mark.parametrize("tag", ["t0", "t5v10"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vct0_native_decoder_step_is_the_python_call_sequence(tag, dtype):
    """``eavqa_t5_decoder_step`` (one C call per greedy step) enqueues exactly the kernels of ``FrozenT5.decode_step``: scores bit-equal,
    ids equal, for the gated (T0) and the ReLU (T5 v1.0) feed-forward."""
    z, T, model, V = _model(tag, dtype)
    model.eval()
    kw = dict(prefix=T(z["fs_prefix"]), question_tokens=T(z["fs_tokens"]), question_mask=T(z["fs_mask"]), special_token_id=V - 1, max_length=9,
              output_scores=True, return_dict_in_generate=True)
    assert model.lm.native_step
    a = model.generate(**kw)
    model.lm.native_step = False
    b = model.generate(**kw)
    model.lm.native_step = True
    assert torch.equal(a.sequences, b.sequences)
    assert torch.equal(torch.stack(list(a.scores)), torch.stack(list(b.scores)))


def test_vct0_splitk_step_route_is_taken_and_agrees_with_the_round3_sequence():
    """bf16: the library's step takes the split-K route (plan present), and its per-step scores stay within bf16 rounding of the round-3 call
    sequence's (different summation order inside the projections)."""
    from eavqa_amd import _lib
    z, T, model, V = _model("t0", torch.bfloat16)
    model.eval()
    assert model.lm.splitk_step_plan(2, 3, 20) is not None
    kw = dict(prefix=T(z["fs_prefix"]), question_tokens=T(z["fs_tokens"]), question_mask=T(z["fs_mask"]), special_token_id=V - 1, max_length=9,
              output_scores=True, return_dict_in_generate=True)
    a = model.generate(**kw)
    model.lm.step_route = 1
    b = model.generate(**kw)
    model.lm.step_route = 0
    sa, sb = torch.stack(list(a.scores)), torch.stack(list(b.scores))
    assert sa.shape[0] >= 1 and (sa[:1] - sb[:1]).abs().max().item() <= 6e-2 * max(1.0, sb.abs().max().item())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 6e-2)])
def test_vct0_cached_decoder_steps_equal_the_reforward_loop(dtype, tol):
    """``use_cache=True`` (one query against the self-attention K / V cache, B rows per step) against ``use_cache=False`` (the decoder re-run
    over its whole prefix every step, as the oracle and HF without a cache do): per-step scores equal within the arithmetic's rounding, ids
    equal in fp32."""
    z, T, model, V = _model("t0", dtype)
    model.eval()
    kw = dict(prefix=T(z["fs_prefix"]), question_tokens=T(z["fs_tokens"]), question_mask=T(z["fs_mask"]), special_token_id=V - 1, max_length=9,
              output_scores=True, return_dict_in_generate=True)
    a, b = model.generate(use_cache=True, **kw), model.generate(use_cache=False, **kw)
    sa, sb = torch.stack(list(a.scores)), torch.stack(list(b.scores))
    assert sa.shape == sb.shape
    if dtype == torch.float32:
        assert (sa - sb).abs().max().item() <= tol * max(1.0, sb.abs().max().item())
        assert torch.equal(a.sequences, b.sequences)
    else:
        assert (sa[:1] - sb[:1]).abs().max().item() <= tol * max(1.0, sb.abs().max().item())      # first step: same decoder prefix in both
        assert a.sequences.shape == b.sequences.shape


@pytest.mark.parametrize("tag", ["t0", "t5v10"])
def test_vct0_generate_stops_early_like_hf_greedy(tag):
    """Rows that emit eos continue with pad and the search stops as soon as every row has finished (HF greedy search; the fixtures' random-init
    models never emit T5's eos, so eos is moved onto tokens they do emit).  The HIP loop looks at the finished flags every fourth step
    only and trims afterwards: sequences, their length and the number of score steps must equal the oracle's step-by-step search."""
    z, T, model, V = _model(tag, torch.float32)
    model.eval()
    sd = {k[3:]: T(v) for k, v in z.items() if k.startswith("lm.")}
    V_, E, DKV, H, F, NL, L, D, gated, tied = [int(v) for v in z["cfg"]]
    mapper = {k[4:]: T(v) for k, v in z.items() if k.startswith("map.")}
    base = model.generate(prefix=T(z["fs_prefix"]), question_tokens=T(z["fs_tokens"]), question_mask=T(z["fs_mask"]), special_token_id=V - 1, max_length=12)
    first = base[:, 1].tolist()
    stopped_early = 0
    # (eos, rows): the whole batch with every emitted token as eos (some rows finish, the search goes on) and one never emitted; then only
    # the rows that emit token e, with eos = e: every row finishes at the first step and the search stops there
    cases = [(e, list(range(len(first)))) for e in sorted(set(first)) + [V - 2]] + [(e, [i for i, f in enumerate(first) if f == e]) for e in sorted(set(first))]
    for eos, rows in cases:
        model.lm.cfg.eos_token_id = eos
        pf, tk, mk = T(z["fs_prefix"])[rows], T(z["fs_tokens"])[rows], T(z["fs_mask"])[rows]
        got = model.generate(prefix=pf, question_tokens=tk, question_mask=mk, special_token_id=V - 1, max_length=12, output_scores=True,
                             return_dict_in_generate=True)
        ocfg = dict(n_layer=NL, n_head=H, d_kv=DKV, gated=bool(gated), tied=bool(tied), eos_token_id=eos)
        with torch.no_grad():
            want, want_scores = oracle.vct0_generate(sd, ocfg, mapper, dict(prefix_length=L, mapping_type="mlp"), pf, tk, mk, max_length=12,
                                                     special_token_id=V - 1)
        assert got.sequences.shape == want.shape and len(got.scores) == len(want_scores), (eos, rows, got.sequences.shape, want.shape)
        top2 = torch.stack(want_scores).topk(2, dim=-1).values
        if ((top2[..., 0] - top2[..., 1]) > 1e-3).all().item():
            assert torch.equal(got.sequences, want), (eos, rows, got.sequences, want)
        stopped_early += int(want.shape[1] < 12)
    model.lm.cfg.eos_token_id = 1
    assert stopped_early >= 1                            # the case this test exists for did occur


def test_vct0_prefix_trains_only_the_mapper():
    z, T, model, V = _model("t0", torch.float32)
    names = {n for n, _ in model.clip_project.named_parameters()}
    assert {id(p) for p in model.parameters()} == {id(p) for p in model.clip_project.parameters()} and names


@pytest.mark.parametrize("tag", ["t0", "t5v10"])
def test_vct0_generate_decoder_prompt_matches_reference_ids(tag):
    """``VCT0Model.generate(decoder_input_ids=..., decoder_attention_mask=...)`` (vct0.py:468-480) against the reference's own output
    (tests/golden/make_golden.py --vct0-only): (a) a prompt no row of which begins with the decoder start id - HF prepends it and the
    reference's slice by the GIVEN prompt length keeps the prompt's last token; (b) a LEFT-PADDED prompt as the reference tokenises decoder
    prompts (module_parser.py:397-399): pad id == start id, nothing prepended, the pads are masked keys of the decoder self-attention."""
    z, T, model, V = _model(tag, torch.float32)
    model.eval()
    kw = dict(prefix=T(z["fs_prefix"]), question_tokens=T(z["dp_tokens"]), question_mask=T(z["dp_mask"]), special_token_id=V - 1, max_length=9)
    a = model.generate(decoder_input_ids=T(z["dp_dec_a"]), decoder_attention_mask=torch.ones_like(T(z["dp_dec_a"])), **kw)
    assert a.tolist() == z["gen_dp_a_ids"].tolist()
    b = model.generate(decoder_input_ids=T(z["dp_dec_b"]), decoder_attention_mask=T(z["dp_dec_b_mask"]), **kw)
    assert b.tolist() == z["gen_dp_b_ids"].tolist()
    # the padded mask matters: ignoring it changes what the decoder attends to (at least the scores; the tiny model's ids may coincide)
    assert model.generate(decoder_input_ids=T(z["dp_dec_b"]), **kw).shape == b.shape


def test_vct0_executors_from_config_train_and_generate():
    """``VCT0Executor`` / ``FewShotVQAExecutor`` constructed from this build's jsonnet configs (model injected: the tiny reference-fixture
    T5): a two-step ``fit`` lowers the loss; the few-shot generative step reproduces the reference's ids on the plain path, and the
    permutation ensemble picks, per question, the member the CPU oracle's scores pick (few_shot_vqa_executor.py:293-332)."""
    import os
    import numpy as np
    from eavqa_amd.trainers.vct0_executor import FewShotVQAExecutor, VCT0Executor
    from eavqa_amd.utils import config_system as cs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z, T, model, V = _model("t0", torch.float32)
    cfg = cs.load_config(os.path.join(root, "configs", "conceptual_captions", "vct0_t0_3b.jsonnet"))
    assert cfg.model_config.ModelClass == "VCT0Prefix" and cfg.train.type == "VCT0Executor"
    ex = VCT0Executor(cfg, model=model, dtype=torch.float32, device=DEV)
    batch = {"clip_embeddings": T(z["prefix"]), "labels": T(z["labels"])}
    losses = ex.fit([batch] * 6, accumulate_grad_batches=1)
    assert abs(losses[0].item() - float(z["loss"])) <= 2e-4 and losses[-1].item() < losses[0].item()
    step = ex._generative_step({**batch, "image_urls": ["u"] * 3}, 0)
    assert len(step["predictions"]) == 3 and step["outputs"].shape[0] == 3
    # few-shot executor on a fresh copy of the fixture weights (fit above moved the mapper)
    z, T, model, V = _model("t0", torch.float32)
    model.eval()
    few = cs.load_config(os.path.join(root, "configs", "vqa2", "few_shot_vqa_t0_3b.jsonnet"), mode="test",
                         opts=[f"data_loader.additional.special_token_id={V - 1}", "data_loader.additional.max_target_length=9"])
    assert few.train.type == "FewShotVQAExecutor"
    fx = FewShotVQAExecutor(few, model=model, dtype=torch.float32, device=DEV)
    plain = fx._generative_step({"generative_input_ids": T(z["fs_tokens"]), "generative_attention_mask": T(z["fs_mask"]), "clip_embeddings": T(z["fs_prefix"]),
                                 "labels": T(z["labels"])}, 0)
    assert torch.equal(plain["outputs"], T(z["gen_fs_ids"]))
    # permutation ensemble: two "permutations" = the fixture prompt and the same prompt with images 0 and 1 swapped in the embeddings
    few.data_loader.additional.num_permutations_of_in_context_examples = 2
    toks = torch.stack([T(z["fs_tokens"]), T(z["fs_tokens"])], dim=1)               # [B, 2, T]
    msk = torch.stack([T(z["fs_mask"]), T(z["fs_mask"])], dim=1)
    pf = T(z["fs_prefix"])[:, :, 0]                                                 # [B, 3, D]
    emb = torch.stack([pf, pf[:, [1, 0, 2]]], dim=1)                                # [B, 2, 3, D]
    ens = fx._generative_step({"generative_input_ids": toks.reshape(-1, toks.shape[-1]), "generative_attention_mask": msk.reshape(-1, msk.shape[-1]),
                               "clip_embeddings": emb, "labels": T(z["labels"])}, 0)
    sd = {k[3:]: T(v) for k, v in z.items() if k.startswith("lm.")}
    V_, E, DKV, H, F, NL, L, D, gated, tied = [int(v) for v in z["cfg"]]
    ocfg = dict(n_layer=NL, n_head=H, d_kv=DKV, gated=bool(gated), tied=bool(tied))
    mapper = {k[4:]: T(v) for k, v in z.items() if k.startswith("map.")}
    want = []
    scores = np.zeros((3, 2))
    seqs = []
    with torch.no_grad():
        for i in range(2):
            seq, sc = oracle.vct0_generate(sd, ocfg, mapper, dict(prefix_length=L, mapping_type="mlp"), emb[:, i], toks[:, i], msk[:, i], max_length=9,
                                           special_token_id=V - 1)
            logp = torch.log(torch.stack(sc).softmax(-1))
            for j, s_ in enumerate(seq.tolist()):
                scores[j, i] = sum(float(logp[k - 1, j, t]) for k, t in enumerate(s_) if t not in (0, 1, 2))
            seqs.append(seq)
    for j, ind in enumerate(np.argmax(scores, axis=1)):
        gap = abs(scores[j, 0] - scores[j, 1])
        if gap > 1e-2:
            assert ens["outputs"][j].tolist() == seqs[ind][j].tolist(), (j, scores[j])
