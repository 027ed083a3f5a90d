"""GPU: the executor mirror (eavqa_amd.trainers.clipcap_executor) end to end on synthetic batches:
config -> model by name -> training_step (reference label masking) -> backward -> fused AdamW, against the same steps
taken with the CPU oracle + oracle AdamW; and _generative_step with a duck-typed tokenizer."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle
from conftest import load_golden

DEV = "cuda"


def T(a):
    return torch.from_numpy(np.asarray(a))


def sub(z, prefix):
    return {k[len(prefix):]: T(v) for k, v in z.items() if k.startswith(prefix)}


class FakeTokenizer:
    """The attributes the executor touches (clipcap_exector.py:55-56,135-143,241-272)."""

    def __init__(self, vocab, eos, bos):
        self.vocab, self.eos_token_id, self.bos_token_id, self.pad_token_id = vocab, eos, bos, None
        self.eos_token = "<eos>"

    @property
    def pad_token(self):
        return None if self.pad_token_id is None else "<eos>"

    @pad_token.setter
    def pad_token(self, tok):
        self.pad_token_id = self.eos_token_id

    def __len__(self):
        return self.vocab

    def decode(self, ids, skip_special_tokens=True):
        ids = [int(i) for i in ids]
        if skip_special_tokens:
            ids = [i for i in ids if i not in (self.eos_token_id, self.bos_token_id)]
        return " ".join(str(i) for i in ids)


def make_executor(z, dtype):
    from eavqa_amd.models.clipcap import ClipCaptionPrefix
    from eavqa_amd.models.lm import FrozenCausalLM, LMConfig
    from eavqa_amd.trainers.clipcap_executor import ClipCapExecutor
    from eavqa_amd.utils.config_system import load_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_config(os.path.join(root, "configs", "vqa2", "clip_cap_gpt2_large.jsonnet"),
                      opts=["train.lr=0.01", "data_loader.type=DataLoaderVQA2", "data_loader.additional.max_target_length=4"])
    V, E, NLAY, NH, NPOS, L, D, CL, NL = [int(v) for v in z["cfg"]]
    lm = FrozenCausalLM(LMConfig("gpt2", NLAY, NH, E, 4 * E, V, NPOS, 1e-5, "gelu_new", V - 1, None), sub(z, "lm."), dtype, DEV)
    model = ClipCaptionPrefix(prefix_length=L, prefix_size=D, mapping_type="mlp", lm=lm, dtype=dtype, device=DEV)
    model.clip_project.load_state_dict(sub(z, "map."))
    tok = FakeTokenizer(V, eos=V - 1, bos=V - 2)
    loader = type("Loader", (), {"tokenizer": tok, "decoder_tokenizer": tok})()
    return ClipCapExecutor(cfg, loader, model=model, dtype=dtype, device=DEV), tok, (V, E, NLAY, NH, L, D)


def vqa_batch(V, D, bos, pad, seed):
    g = torch.Generator().manual_seed(seed)
    B, T_ = 4, 10
    ids = torch.randint(0, V - 3, (B, T_), generator=g)
    qlen = [3, 5, 2, 4]
    alen = [2, 1, 3, 2]
    mask = torch.zeros(B, T_, dtype=torch.long)
    for b in range(B):
        ids[b, qlen[b]] = bos
        end = qlen[b] + 1 + alen[b]
        ids[b, end:] = pad
        mask[b, :end] = 1
    return dict(input_ids=ids, attention_mask=mask, clip_embeddings=torch.randn(B, 1, 1, D, generator=g))


def test_fit_matches_oracle_training_fp32():
    """3 optimiser steps with accumulate_grad_batches=2 (6 batches): losses and final mapper weights equal the oracle's
    (reference label masking, mean CE, AdamW with torch defaults)."""
    z = load_golden("clipcap_gpt2_mlp.npz")
    ex, tok, (V, E, NLAY, NH, L, D) = make_executor(z, torch.float32)
    batches = [vqa_batch(V, D, tok.bos_token_id, tok.eos_token_id, 100 + i) for i in range(6)]
    losses = ex.fit(batches, accumulate_grad_batches=2)
    assert ex.global_step == 3 and "train/loss" in ex.logged and "train/lr[0]" in ex.logged
    # oracle replay
    sd = sub(z, "lm.")
    mapper = {k: v.clone().requires_grad_(True) for k, v in sub(z, "map.").items()}
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in mapper.items()}
    ocfg, mcfg = dict(arch="gpt2", n_layer=NLAY, n_head=NH), dict(prefix_length=L, mapping_type="mlp")
    want = []
    for i, b in enumerate(batches):
        labels = oracle.label_mask_vqa(b["input_ids"], tok.eos_token_id, tok.bos_token_id)
        loss, _ = oracle.clipcap_forward(sd, ocfg, mapper, mcfg, b["input_ids"], b["clip_embeddings"].reshape(4, D), b["attention_mask"], labels)
        loss.backward()
        want.append(loss.item())
        if i % 2 == 1:
            with torch.no_grad():
                for k, p in mapper.items():
                    oracle.adamw_step(p, p.grad / 2, state[k][0], state[k][1], i // 2 + 1, 0.01)
                    p.grad = None
    assert np.allclose([l.item() for l in losses], want, atol=2e-4), (losses, want)
    for k, p in ex.model.clip_project.state_dict().items():
        assert torch.allclose(p.cpu(), mapper[k].detach(), atol=2e-4), k
    # checkpoint round trip (mapper-only + reference-style "model." prefix)
    ck = ex.state_dict()
    assert all(k.startswith("model.clip_project.") for k in ck["state_dict"])
    ex2, _, _ = make_executor(z, torch.float32)
    ex2.configure_optimizers()
    ex2.load_state_dict(ck)
    for (k, a), (_, b) in zip(ex.model.clip_project.state_dict().items(), ex2.model.clip_project.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_generative_step_contract(dtype):
    z = load_golden("clipcap_gpt2_mlp.npz")
    ex, tok, (V, E, NLAY, NH, L, D) = make_executor(z, dtype)
    ex.model.eval()
    b = vqa_batch(V, D, tok.bos_token_id, tok.eos_token_id, 7)
    batch = dict(generative_input_ids=b["input_ids"][:, :5], generative_attention_mask=b["attention_mask"][:, :5],
                 clip_embeddings=b["clip_embeddings"], labels=b["input_ids"], question_ids=[11, 12, 13, 14], answers=[["a"]] * 4)
    out = ex.test_step(batch, 0)
    assert [p["question_id"] for p in out["predictions"]] == [11, 12, 13, 14]
    assert len(out["outputs"]) == 4 and all(1 <= len(o) <= 4 for o in out["outputs"])
    assert all(isinstance(p["answer"], str) for p in out["predictions"])


def test_fit_from_a_conceptual_captions_parquet_store(tmp_path):
    """Batches read from the reference's parquet store format (data/stores.py) drive the executor's CC branch
    (labels come masked from the collate, data_loader_conceptual_captions.py:94-95); first loss equals the oracle's."""
    import types
    from eavqa_amd.data import stores
    z = load_golden("clipcap_gpt2_mlp.npz")
    ex, tok, (V, E, NLAY, NH, L, D) = make_executor(z, torch.float32)
    ex.config.data_loader.type = "DataLoaderConceptualCaptions"
    g = torch.Generator().manual_seed(5)
    N = 12
    emb = torch.randn(N, D, generator=g)
    ids = [torch.randint(0, V - 3, (int(n),), generator=g).tolist() for n in torch.randint(3, 9, (N,), generator=g)]
    caps = [" ".join(str(t) for t in row) + " ." for row in ids]
    path = str(tmp_path / "cc.parquet")
    stores.write_cc_parquet(path, [f"u{i}" for i in range(N)], caps, emb, wrap=True, row_group_size=5)

    class IdTokenizer:     # captions are already "token id" strings; the trailing period maps to eos like a sentence end
        pad_token_id = tok.eos_token_id

        def __call__(self, texts, padding, max_length, truncation, return_tensors):
            rows = [[tok.eos_token_id if w == "." else int(w) for w in t.split()][:max_length] for t in texts]
            T_ = max(len(r) for r in rows)
            return types.SimpleNamespace(input_ids=torch.tensor([r + [self.pad_token_id] * (T_ - len(r)) for r in rows]),
                                         attention_mask=torch.tensor([[1] * len(r) + [0] * (T_ - len(r)) for r in rows]))

    batches = list(stores.ConceptualCaptionsParquet(path).iter_batches(4, IdTokenizer(), max_source_length=8))
    assert len(batches) == 3 and tuple(batches[0]["clip_embeddings"].shape) == (4, D)
    losses = ex.fit(batches, accumulate_grad_batches=1)
    assert ex.global_step == 3 and all(torch.isfinite(l) for l in losses)
    b = batches[0]
    mapper = sub(z, "map.")
    want, _ = oracle.clipcap_forward(sub(z, "lm."), dict(arch="gpt2", n_layer=NLAY, n_head=NH), mapper, dict(prefix_length=L, mapping_type="mlp"),
                                     b["input_ids"], b["clip_embeddings"], b["attention_mask"], b["labels"])
    assert abs(losses[0].item() - want.item()) <= 2e-4


def test_vqa_label_count_matches_the_masking_rule():
    """The host-side count that sizes the scored-row compaction equals the number of labels the reference rule keeps
    (oracle.label_mask_vqa), on rows with / without <BOS>, without padding, with repeated <BOS> and an all-pad row."""
    from eavqa_amd.trainers.clipcap_executor import vqa_label_count
    pad, bos = 99, 98
    ids = torch.tensor([[5, 6, bos, 7, 8, pad, pad, pad],
                        [5, bos, 7, 8, 9, 10, 11, 12],       # no padding: no restored eos
                        [5, 6, 7, 8, pad, pad, pad, pad],    # no <BOS>: only the restored pad is scored
                        [bos, 7, bos, 8, 9, pad, pad, pad],  # a second <BOS> is masked itself
                        [pad, pad, pad, pad, pad, pad, pad, pad],
                        [5, 6, 7, 8, 9, 10, 11, bos]])
    want = int((oracle.label_mask_vqa(ids, pad, bos) != -100).sum())
    assert vqa_label_count(ids, pad, bos) == want
    g = torch.Generator().manual_seed(0)
    rnd_ids = torch.randint(90, 100, (64, 12), generator=g)
    assert vqa_label_count(rnd_ids, pad, bos) == int((oracle.label_mask_vqa(rnd_ids, pad, bos) != -100).sum())
    assert vqa_label_count(ids, pad, None) == int((oracle.label_mask_vqa(ids, pad, -1) != -100).sum())
