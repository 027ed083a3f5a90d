"""CPU: the oracle's T5 / VCT0 restatement (oracle/ref_cpu.py) against fixtures produced by the REFERENCE's own VCT0Prefix
(src/models/vct0.py) on tiny local T5ForConditionalGeneration checkpoints (tests/golden/make_golden.py::vct0_golden): training forward
(loss, logits, mapper gradients) and greedy generation on all four paths of ``VCT0Model.generate``."""
import pytest
import torch

import oracle
from conftest import load_golden


def _case(tag):
    z = load_golden(f"vct0_{tag}.npz")
    T = lambda a: torch.from_numpy(a)
    V, E, DKV, H, F, NL, L, D, gated, tied = [int(v) for v in z["cfg"]]
    sd = {k[3:]: T(v) for k, v in z.items() if k.startswith("lm.")}
    cfg = dict(n_layer=NL, n_head=H, d_kv=DKV, gated=bool(gated), tied=bool(tied), eps=1e-6)
    mapper = {k[4:]: T(v) for k, v in z.items() if k.startswith("map.")}
    return z, T, sd, cfg, mapper, dict(prefix_length=L, mapping_type="mlp"), V


@pytest.mark.parametrize("tag", ["t0", "t5v10"])
def test_vct0_forward_matches_reference(tag):
    z, T, sd, cfg, mapper, mcfg, V = _case(tag)
    mp = {k: v.clone().requires_grad_(True) for k, v in mapper.items()}
    loss, logits = oracle.vct0_forward(sd, cfg, mp, mcfg, T(z["prefix"]), T(z["labels"]))
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) <= 1e-5
    assert (logits.detach() - T(z["logits"])).abs().max().item() <= 2e-5
    for k, p in mp.items():
        want = T(z["gmap." + k])
        assert (p.grad - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item()), k


@pytest.mark.parametrize("tag", ["t0", "t5v10"])
def test_vct0_generate_paths_match_reference(tag):
    z, T, sd, cfg, mapper, mcfg, V = _case(tag)
    special = V - 1                                  # the tiny vocabulary's sentinel ids are V - 1 - i (32099 - i in T5's)
    with torch.no_grad():
        runs = {
            "prefix": oracle.vct0_generate(sd, cfg, mapper, mcfg, T(z["prefix"]), max_length=9),
            "fs": oracle.vct0_generate(sd, cfg, mapper, mcfg, T(z["fs_prefix"]), T(z["fs_tokens"]), T(z["fs_mask"]), max_length=9, special_token_id=special),
            "one": oracle.vct0_generate(sd, cfg, mapper, mcfg, T(z["fs_prefix"]), T(z["one_tokens"]), T(z["one_mask"]), max_length=9,
                                        special_token_id=special, one_at_a_time=True),
            "text": oracle.vct0_generate(sd, cfg, mapper, mcfg, T(z["fs_prefix"]), T(z["fs_tokens"]), T(z["fs_mask"]), max_length=9, no_prefix=True),
        }
    for name, (seq, scores) in runs.items():
        want_ids, want_scores = T(z[f"gen_{name}_ids"]), T(z[f"gen_{name}_scores"])
        assert torch.equal(seq, want_ids), (name, seq, want_ids)
        got = torch.stack(scores)
        assert got.shape == want_scores.shape and (got - want_scores).abs().max().item() <= 5e-5, name


@pytest.mark.parametrize("tag", ["t0", "t5v10"])
def test_vct0_generate_decoder_prompt_matches_reference(tag):
    """The decoder-prompt branch (vct0.py:468-480) on the reference's own outputs: a prompt HF prepends the start token to, and a left-padded
    prompt with its mask (module_parser.py:397-399)."""
    z, T, sd, cfg, mapper, mcfg, V = _case(tag)
    kw = dict(max_length=9, special_token_id=V - 1)
    with torch.no_grad():
        a, _ = oracle.vct0_generate(sd, cfg, mapper, mcfg, T(z["fs_prefix"]), T(z["dp_tokens"]), T(z["dp_mask"]), decoder_input_ids=T(z["dp_dec_a"]),
                                    decoder_attention_mask=torch.ones_like(T(z["dp_dec_a"])), **kw)
        b, _ = oracle.vct0_generate(sd, cfg, mapper, mcfg, T(z["fs_prefix"]), T(z["dp_tokens"]), T(z["dp_mask"]), decoder_input_ids=T(z["dp_dec_b"]),
                                    decoder_attention_mask=T(z["dp_dec_b_mask"]), **kw)
    assert torch.equal(a, T(z["gen_dp_a_ids"])) and torch.equal(b, T(z["gen_dp_b_ids"]))


def test_t5_relative_buckets_known_values():
    """Spot values of T5's bucket function (HF:models/t5/modeling_t5.py:217-262): exact buckets below 8 (bidirectional) / 16 (causal),
    logarithmic beyond, the sign in the upper half for the bidirectional case."""
    rel = torch.tensor([-200, -128, -20, -8, -7, -1, 0, 1, 7, 8, 20, 127, 300])
    bi = oracle.t5_relative_bucket(rel, True).tolist()
    assert bi[6] == 0 and bi[5] == 1 and bi[7] == 17 and bi[4] == 7 and bi[8] == 23 and bi[0] == 15 and bi[-1] == 31 and bi[3] == 8 and bi[9] == 24
    uni = oracle.t5_relative_bucket(rel, False).tolist()
    assert uni[6:] == [0] * 7 and uni[5] == 1 and uni[3] == 8 and uni[0] == 31 and uni[2] == 17
