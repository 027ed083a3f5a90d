"""CPU: the oracle (oracle/ref_cpu.py) against the fixtures generated from the reference.

These pin the oracle.  Tolerances: fp32 CPU vs fp32 CPU, different op order
only -> 2e-5 absolute on logits/loss of the tiny models; integer work bit-exact.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden

ATOL = 2e-5


def T(a):
    return torch.from_numpy(np.asarray(a))


def sub(z, prefix):
    return {k[len(prefix):]: T(v) for k, v in z.items() if k.startswith(prefix)}


def test_mlp_mapper_forward_and_grads():
    z = load_golden("mapper_mlp.npz")
    w = {k: v.clone().requires_grad_(True) for k, v in sub(z, "w.").items()}
    y = oracle.mlp_mapper(T(z["x"]), w)
    assert torch.allclose(y, T(z["y"]), atol=ATOL)
    (y * T(z["gy"])).sum().backward()
    for k, g in sub(z, "g.").items():
        assert torch.allclose(w[k].grad, g, atol=ATOL), k


def test_transformer_mapper_forward_and_grads():
    z = load_golden("mapper_transformer.npz")
    D, E, L, CL, NL = [int(v) for v in z["cfg"]]
    w = {k: v.clone().requires_grad_(True) for k, v in sub(z, "w.").items()}
    y = oracle.transformer_mapper(T(z["x"]), w, CL, NL)
    assert y.shape == (3, L, E)
    assert torch.allclose(y, T(z["y"]), atol=ATOL)
    (y * T(z["gy"])).sum().backward()
    for k, g in sub(z, "g.").items():
        assert torch.allclose(w[k].grad, g, atol=5e-5), k


def _gpt2_setup(z):
    V, E, NLAY, NH, NPOS, L, D, CL, NL = [int(v) for v in z["cfg"]]
    cfg = dict(arch="gpt2", n_layer=NLAY, n_head=NH)
    return cfg, dict(prefix_length=L, clip_length=CL, num_layers=NL)


@pytest.mark.parametrize("mapping_type", ["mlp", "transformer"])
def test_clipcap_gpt2_forward_loss_grads(mapping_type):
    z = load_golden(f"clipcap_gpt2_{mapping_type}.npz")
    cfg, mcfg = _gpt2_setup(z)
    mcfg["mapping_type"] = mapping_type
    sd = sub(z, "lm.")
    mapper = {k: v.clone().requires_grad_(True) for k, v in sub(z, "map.").items()}
    loss, logits = oracle.clipcap_forward(sd, cfg, mapper, mcfg, T(z["ids"]), T(z["prefix"]), T(z["mask"]), T(z["labels"]))
    assert torch.allclose(logits, T(z["logits"]), atol=ATOL)
    assert abs(float(loss.detach()) - float(z["loss"])) < ATOL
    loss.backward()
    for k, g in sub(z, "g.").items():
        assert torch.allclose(mapper[k].grad, g, atol=ATOL), k


@pytest.mark.parametrize("mapping_type", ["mlp", "transformer"])
def test_clipcap_gpt2_generate_ids_exact(mapping_type):
    z = load_golden(f"clipcap_gpt2_{mapping_type}.npz")
    cfg, mcfg = _gpt2_setup(z)
    mcfg["mapping_type"] = mapping_type
    sd, mapper = sub(z, "lm."), sub(z, "map.")
    pad = int(z["pad_id"])
    args = (sd, cfg, mapper, mcfg, T(z["gen_ids"]), T(z["prefix"]), T(z["gen_mask"]))
    with torch.no_grad():
        free = oracle.clipcap_generate(*args, max_length=6, pad_token_id=pad, eos_token_id=None)
        forced = oracle.clipcap_generate(*args, max_length=6, pad_token_id=pad, eos_token_id=int(z["gen_forced_eos"]))
        early = oracle.clipcap_generate(sd, cfg, mapper, mcfg, T(z["gen_ids"])[:1], T(z["prefix"])[:1], T(z["gen_mask"])[:1],
                                        max_length=6, pad_token_id=pad, eos_token_id=int(z["gen_early_eos"]))
    assert free == z["gen_free"].tolist()
    assert forced == z["gen_forced"].tolist()
    assert early == z["gen_early"].tolist()
    # the forced-EOS row really finished early and emits pad afterwards (clipcap.py:431-434)
    assert forced[1][3:] == [pad] * (len(forced[1]) - 3)
    assert len(early[0]) == 1  # all rows finished after the first token -> early break (clipcap.py:463)


def test_clipcap_opt_forward_and_generate():
    z = load_golden("clipcap_opt_mlp.npz")
    V, E, NLAY, NH, NPOS, L, D, FFN = [int(v) for v in z["cfg"]]
    cfg = dict(arch="opt", n_layer=NLAY, n_head=NH)
    mcfg = dict(prefix_length=L, mapping_type="mlp")
    sd = sub(z, "lm.")
    mapper = {k: v.clone().requires_grad_(True) for k, v in sub(z, "map.").items()}
    loss, logits = oracle.clipcap_forward(sd, cfg, mapper, mcfg, T(z["ids"]), T(z["prefix"]), T(z["mask"]), T(z["labels"]))
    assert torch.allclose(logits, T(z["logits"]), atol=ATOL)
    assert abs(float(loss.detach()) - float(z["loss"])) < ATOL
    loss.backward()
    for k, g in sub(z, "g.").items():
        assert torch.allclose(mapper[k].grad, g, atol=ATOL), k
    pad = int(z["pad_id"])
    with torch.no_grad():
        m2 = {k: v.detach() for k, v in mapper.items()}
        free = oracle.clipcap_generate(sd, cfg, m2, mcfg, T(z["gen_ids"]), T(z["prefix"]), T(z["gen_mask"]),
                                       max_length=5, pad_token_id=pad, eos_token_id=None)
        forced = oracle.clipcap_generate(sd, cfg, m2, mcfg, T(z["gen_ids"]), T(z["prefix"]), T(z["gen_mask"]),
                                         max_length=5, pad_token_id=pad, eos_token_id=int(z["gen_forced_eos"]))
    assert free == z["gen_free"].tolist()
    assert forced == z["gen_forced"].tolist()


@pytest.mark.parametrize("case", ["z", "f", "s"])
def test_insert_prefix_matches_reference(case):
    z = load_golden("insert_prefix.npz")
    if case == "s":
        L, E, shots = [int(v) for v in z["s_cfg"]]
    else:
        L, E, shots = 2, 3, (0 if case == "z" else 2)
    emb, msk = oracle.insert_prefix_into_input(L, shots, T(z[f"{case}_tok"]), T(z[f"{case}_text"]), T(z[f"{case}_pp"]),
                                               T(z[f"{case}_mask"]))
    assert torch.equal(emb, T(z[f"{case}_emb"]))
    assert torch.equal(msk, T(z[f"{case}_out_mask"]))


def test_insert_prefix_known_answers_from_reference_tests():
    """Expected values spelled out in src/models/vct0_test.py:110-139 and :178-207."""
    z = load_golden("insert_prefix.npz")
    text, pp = T(z["f_text"]), T(z["f_pp"])
    emb, msk = oracle.insert_prefix_into_input(2, 2, T(z["f_tok"]), text, pp, T(z["f_mask"]))
    exp0 = torch.stack([*pp[0, 0], text[0, 1], *pp[0, 1], text[0, 3], *pp[0, 2], text[0, 5], text[0, 6]])
    exp1 = torch.stack([text[1, 0], *pp[1, 0], text[1, 2], *pp[1, 1], text[1, 4], *pp[1, 2], text[1, 6]])
    assert torch.equal(emb, torch.stack([exp0, exp1]))
    assert msk.tolist() == [[1, 1, 1, 1, 1, 1, 1, 1, 1, 0], [1] * 10]
    emb, msk = oracle.insert_prefix_into_input(2, 0, T(z["z_tok"]), T(z["z_text"]), T(z["z_pp"]), T(z["z_mask"]))
    text, pp = T(z["z_text"]), T(z["z_pp"])
    exp0 = torch.stack([*pp[0, 0], *text[0, 1:]])
    exp1 = torch.stack([text[1, 0], *pp[1, 0], *text[1, 2:]])
    assert torch.equal(emb, torch.stack([exp0, exp1]))
    assert msk.tolist() == [[1, 1, 1, 1, 1, 1, 1, 0], [1] * 8]


def test_insert_prefix_rejects_wrong_sentinel_count():
    z = load_golden("insert_prefix.npz")
    tok = T(z["f_tok"]).clone()
    tok[0, 0] = 5  # row 0 now has 2 sentinels for 3 images
    with pytest.raises(ValueError):
        oracle.insert_prefix_into_input(2, 2, tok, T(z["f_text"]), T(z["f_pp"]), T(z["f_mask"]))


def test_label_mask_known_answers():
    z = load_golden("label_mask.npz")
    out = oracle.label_mask_vqa(T(z["input_ids"]), int(z["pad_id"]), int(z["bos_id"]))
    assert out.tolist() == z["labels"].tolist()
    cc = oracle.label_mask_cc(T(z["input_ids"]), int(z["pad_id"]))
    assert cc.tolist() == np.where(z["input_ids"] == int(z["pad_id"]), -100, z["input_ids"]).tolist()


@pytest.mark.parametrize("name", ["clip_vit.npz", "clip_vit_p14.npz"])
def test_clip_vit_encode(name):
    z = load_golden(name)
    W, MLP, NL, NH, IMG, P, D = [int(v) for v in z["cfg"]]
    cfg = dict(width=W, n_layer=NL, n_head=NH, patch=P)
    emb = oracle.clip_vit_encode(sub(z, "w."), cfg, T(z["pixels"]))
    assert emb.shape == (z["pixels"].shape[0], D)
    assert torch.allclose(emb, T(z["image_embeds"]), atol=ATOL)


def test_adamw_step_matches_torch():
    torch.manual_seed(0)
    p = torch.randn(37, 5)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-4)  # clipcap_exector.py:79-81 (defaults otherwise)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        g = torch.randn_like(p)
        ref.grad = g.clone()
        opt.step()
        oracle.adamw_step(p, g, m, v, step, 1e-4)
        assert torch.allclose(p, ref.detach(), atol=1e-7)


def test_oracle_reproduces_the_reference_after_resize_token_embeddings():
    """``resize_gpt2.npz``: the reference model after ``gpt.resize_token_embeddings(V + 1)`` (clipcap_exector.py:55-56) on a
    VQA-style batch whose answers start at the NEW token id V: the oracle on the grown matrix gives the reference's
    logits [B, L+T, V+1], loss, mapper gradients and greedy ids; the new row is the mean of the old ones (what this build's
    ``FrozenCausalLM.resize_token_embeddings`` writes) to within HF's 1e-9-covariance draw."""
    z = load_golden("resize_gpt2.npz")
    zl = load_golden("clipcap_gpt2_mlp.npz")
    V, E, NLAY, NH, NPOS, L, D = [int(v) for v in z["cfg"]]
    sd = {k[3:]: T(v) for k, v in zl.items() if k.startswith("lm.")}
    assert np.array_equal(z["wte"][:V], zl["lm.transformer.wte.weight"])
    assert np.abs(z["wte"][V] - z["wte"][:V].mean(0)).max() <= 1e-5
    sd["transformer.wte.weight"] = T(z["wte"])
    sd.pop("lm_head.weight", None)
    mapper = {k[4:]: T(v).clone().requires_grad_(True) for k, v in z.items() if k.startswith("map.")}
    cfg, mcfg = dict(arch="gpt2", n_layer=NLAY, n_head=NH), dict(prefix_length=L, mapping_type="mlp")
    assert np.array_equal(oracle.label_mask_vqa(T(z["ids"]), int(z["pad_id"]), int(z["bos_id"])).numpy(), z["labels"])
    loss, logits = oracle.clipcap_forward(sd, cfg, mapper, mcfg, T(z["ids"]), T(z["prefix"]), T(z["mask"]), T(z["labels"]))
    assert logits.shape[-1] == V + 1
    assert (logits.detach() - T(z["logits"])).abs().max().item() <= 1e-4
    assert abs(loss.item() - float(z["loss"])) <= 1e-5
    loss.backward()
    for k, p in mapper.items():
        assert (p.grad - T(z["g." + k])).abs().max().item() <= 1e-5, k
    with torch.no_grad():
        ids = oracle.clipcap_generate(sd, cfg, {k: v.detach() for k, v in mapper.items()}, mcfg, T(z["gen_ids"]), T(z["prefix"]),
                                      T(z["gen_mask"]), max_length=5, pad_token_id=int(z["pad_id"]), eos_token_id=int(z["pad_id"]))
    assert ids == z["gen"].tolist()
