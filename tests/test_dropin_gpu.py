"""GPU: the REAL construction path of the reference executor (src/trainers/clipcap_exector.py:52-56):

    ModelClass = globals()[config.model_config.ModelClass]
    self.model = ModelClass(**config.model_config.model_args)        # -> from_pretrained(model_version)  (clipcap.py:252)
    self.tokenizer.pad_token = self.tokenizer.eos_token
    self.model.gpt.resize_token_embeddings(len(self.tokenizer))

driven from a jsonnet config whose ``model_args.model_version`` is an HF-format directory (``tests/golden/hf_*_tiny``: config.json
+ model.safetensors written by ``save_pretrained`` in tests/golden/make_golden.py), with a tokenizer stub that is one token
longer than the LM's vocabulary (the added "<BOS>").  Expected values: fixtures produced by running the reference.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN, load_golden

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a):
    return torch.from_numpy(np.asarray(a))


def sub(z, prefix):
    return {k[len(prefix):]: T(v) for k, v in z.items() if k.startswith(prefix)}


class StubTokenizer:
    """What the executor touches of a HF tokenizer (clipcap_exector.py:55-56,135-143,241-272)."""

    def __init__(self, n, eos, bos, pad=None):
        self.n, self.eos_token_id, self.bos_token_id, self.pad_token_id = n, eos, bos, pad
        self.eos_token = "<eos>"

    @property
    def pad_token(self):
        return None if self.pad_token_id is None else "<pad>"

    @pad_token.setter
    def pad_token(self, tok):
        assert tok == self.eos_token
        self.pad_token_id = self.eos_token_id

    def __len__(self):
        return self.n

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(int(i)) for i in ids if not (skip_special_tokens and int(i) in (self.eos_token_id, self.bos_token_id)))


def executor_from_config(model_dir, tok, dtype, extra_opts=()):
    from eavqa_amd.trainers.clipcap_executor import ClipCapExecutor
    from eavqa_amd.utils.config_system import load_config
    cfg = load_config(os.path.join(ROOT, "configs", "vqa2", "clip_cap_gpt2_large.jsonnet"),
                      opts=[f"model_config.model_args.model_version='{model_dir}'", "model_config.model_args.prefix_length=4",
                            "model_config.model_args.clip_length=3", "model_config.model_args.prefix_size=24",
                            "model_config.model_args.num_layers=2", "data_loader.type=DataLoaderVQA2",
                            "data_loader.additional.max_target_length=5", *extra_opts])
    loader = type("Loader", (), {"tokenizer": tok, "decoder_tokenizer": tok})()
    return ClipCapExecutor(cfg, loader, dtype=dtype, device=DEV)          # model=None: built by name from model_args


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_executor_builds_gpt2_from_a_local_hf_directory_and_grows_the_vocabulary(dtype):
    z = load_golden("resize_gpt2.npz")
    V, E, NLAY, NH, NPOS, L, D = [int(v) for v in z["cfg"]]
    tok = StubTokenizer(V + 1, eos=V - 1, bos=V)
    ex = executor_from_config(os.path.join(GOLDEN, "hf_gpt2_tiny"), tok, dtype)
    lm = ex.model.gpt
    assert type(ex.model).__name__ == "ClipCaptionPrefix" and lm.cfg.arch == "gpt2" and lm.cfg.n_layer == NLAY
    assert lm.vocab == V + 1 and lm.head is lm.wte and tok.pad_token_id == tok.eos_token_id
    # the grown row: this build writes the mean of the old rows, HF >= 4.46 draws N(mean, 1e-9 cov): same to ~1e-6
    assert (lm.wte.float().cpu() - T(z["wte"])).abs().max().item() <= (2e-6 if dtype == torch.float32 else 1e-3)
    ex.model.clip_project.load_state_dict(sub(z, "map."))
    tol = dict(logits=2e-4, loss=2e-5, grad=2e-4) if dtype == torch.float32 else dict(logits=6e-2, loss=2e-2, grad=5e-2)
    # 1. the training step through the executor (its own label masking on the device)
    ex.model.train()
    ex.model.pack_padding = False
    batch = dict(input_ids=T(z["ids"]), attention_mask=T(z["mask"]), clip_embeddings=T(z["prefix"])[:, None, None, :])
    ex.configure_optimizers()
    loss = ex.training_step(batch, 0)["loss"]
    assert abs(loss.item() - float(z["loss"])) <= tol["loss"]
    loss.backward()
    for k, g in sub(z, "g.").items():
        p = dict(ex.model.clip_project.named_parameters())[k]
        assert (p.grad.cpu() - g).abs().max().item() <= tol["grad"] * max(1.0, g.abs().max().item()), k
    # 2. logits over the grown vocabulary, with the exact rows of a reference checkpoint loaded through load_state_dict
    ck = {"state_dict": {**{"model.clip_project." + k: v for k, v in sub(z, "map.").items()},
                         "model.gpt.transformer.wte.weight": T(z["wte"]), "model.gpt.lm_head.weight": T(z["wte"])}, "global_step": 7}
    ex.load_state_dict(ck)
    assert ex.global_step == 7 and ex.scheduler.n == 7
    if dtype == torch.float32:
        assert torch.equal(lm.wte.cpu(), T(z["wte"]))
    out = ex.model(question_tokens=T(z["ids"]), prefix=T(z["prefix"]), question_mask=T(z["mask"]), labels=T(z["labels"]))
    assert tuple(out.logits.shape) == tuple(z["logits"].shape) and out.logits.shape[-1] == V + 1
    assert (out.logits.float().cpu() - T(z["logits"])).abs().max().item() <= tol["logits"]
    # 3. generation through the executor's _generative_step (eos = pad = V - 1)
    ex.model.eval()
    res = ex.test_step(dict(generative_input_ids=T(z["gen_ids"]), generative_attention_mask=T(z["gen_mask"]),
                            clip_embeddings=T(z["prefix"])[:, None, None, :], question_ids=[1, 2, 3, 4]), 0)
    if dtype == torch.float32:
        assert res["outputs"] == z["gen"].tolist()
    assert [p["question_id"] for p in res["predictions"]] == [1, 2, 3, 4]


def test_executor_with_an_unchanged_vocabulary_reproduces_the_reference_fixture():
    """Tokenizer as long as the LM's vocabulary: resize is a no-op and the directory-loaded model gives the logits, loss and
    greedy ids of ``clipcap_gpt2_mlp.npz`` (the same weights, stored as arrays there)."""
    z = load_golden("clipcap_gpt2_mlp.npz")
    V = int(z["cfg"][0])
    ex = executor_from_config(os.path.join(GOLDEN, "hf_gpt2_tiny"), StubTokenizer(V, eos=V - 1, bos=V - 2), torch.float32)
    ex.model.clip_project.load_state_dict(sub(z, "map."))
    ex.model.train()
    ex.model.pack_padding = False
    out = ex.model(question_tokens=T(z["ids"]), prefix=T(z["prefix"]), question_mask=T(z["mask"]), labels=T(z["labels"]))
    assert (out.logits.cpu() - T(z["logits"])).abs().max().item() <= 2e-4 and abs(out.loss.item() - float(z["loss"])) <= 2e-5
    ex.model.eval()
    ids = ex.model.generate(question_tokens=T(z["gen_ids"]), prefix=T(z["prefix"]), question_mask=T(z["gen_mask"]), max_length=6,
                            pad_token_id=int(z["pad_id"]), eos_token_id=None)
    assert ids == z["gen_free"].tolist()


def test_opt_from_a_local_hf_directory_and_pad_token_override():
    """OPT directory (the reference hard-wires GPT-2; BASELINE configs 3-5 need OPT).  The tokenizer HAS a pad
    token (1) different from eos (2): the executor overrides it with eos like the reference (:55), so the VQA label rule
    restores the first pad position to the EOS id."""
    import oracle
    z = load_golden("clipcap_opt_mlp.npz")
    V, E, NLAY, NH, NPOS, L, D, FFN = [int(v) for v in z["cfg"]]
    tok = StubTokenizer(V, eos=2, bos=V - 2, pad=1)
    ex = executor_from_config(os.path.join(GOLDEN, "hf_opt_tiny"), tok, torch.float32)
    lm = ex.model.gpt
    assert lm.cfg.arch == "opt" and lm.cfg.ffn == FFN and lm.vocab == V
    assert tok.pad_token_id == 2 and ex._pad_id() == 2
    ex.model.clip_project.load_state_dict(sub(z, "map."))
    ex.model.train()
    ex.model.pack_padding = False
    out = ex.model(question_tokens=T(z["ids"]), prefix=T(z["prefix"]), question_mask=T(z["mask"]), labels=T(z["labels"]))
    assert (out.logits.cpu() - T(z["logits"])).abs().max().item() <= 2e-4 and abs(out.loss.item() - float(z["loss"])) <= 2e-5
    # executor label rule with pad == eos == 2 against the oracle loop
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(3, V - 3, (3, 9), generator=g)
    mask = torch.ones(3, 9, dtype=torch.long)
    for b, (ql, al) in enumerate(((2, 2), (4, 1), (3, 3))):
        ids[b, ql] = tok.bos_token_id
        ids[b, ql + 1 + al:] = 2
        mask[b, ql + 1 + al:] = 0
    ex.configure_optimizers()
    batch = dict(input_ids=ids, attention_mask=mask, clip_embeddings=torch.randn(3, 1, 1, D, generator=g))
    loss = ex.training_step(batch, 0)["loss"]
    labels = oracle.label_mask_vqa(ids, 2, tok.bos_token_id)
    assert int((labels == 2).sum()) == 3                       # one restored eos per row
    ocfg = dict(arch="opt", n_layer=NLAY, n_head=NH, act="relu")
    want, _ = oracle.clipcap_forward(sub(z, "lm."), ocfg, sub(z, "map."), dict(prefix_length=L, mapping_type="mlp"), ids,
                                     batch["clip_embeddings"].reshape(3, D), mask, labels)
    assert abs(loss.item() - want.item()) <= 5e-5


def test_fit_flushes_trailing_micro_batches_and_resume_continues_the_schedule():
    z = load_golden("clipcap_gpt2_mlp.npz")
    V, D = int(z["cfg"][0]), int(z["cfg"][6])
    tok = StubTokenizer(V, eos=V - 1, bos=V - 2)
    ex = executor_from_config(os.path.join(GOLDEN, "hf_gpt2_tiny"), tok, torch.float32,
                              ["train.lr=0.01", "train.scheduler='linear'", "train.additional.warmup_steps=2"])
    ex.configure_optimizers(num_training_steps=10)
    g = torch.Generator().manual_seed(1)

    def batch():
        ids = torch.randint(0, V - 3, (2, 6), generator=g)
        ids[:, 2] = tok.bos_token_id
        return dict(input_ids=ids, attention_mask=torch.ones(2, 6, dtype=torch.long), clip_embeddings=torch.randn(2, 1, 1, D, generator=g))

    ex.fit([batch() for _ in range(5)], accumulate_grad_batches=2)       # 2 + 2 + 1: the trailing batch is applied too
    assert ex.global_step == 3 and ex.scheduler.n == 3
    assert not ex.model.clip_project.flat.grad_live                       # nothing leaks into the next fit()
    lr3 = ex.scheduler.get_last_lr()[0]
    ck = ex.state_dict()
    ex2 = executor_from_config(os.path.join(GOLDEN, "hf_gpt2_tiny"), tok, torch.float32,
                               ["train.lr=0.01", "train.scheduler='linear'", "train.additional.warmup_steps=2"])
    ex2.configure_optimizers(num_training_steps=10)
    ex2.load_state_dict(ck)
    assert ex2.global_step == 3 and abs(ex2.scheduler.get_last_lr()[0] - lr3) <= 1e-12 and lr3 == pytest.approx(0.01 * 7 / 8)


def test_an_undercounted_label_hint_poisons_the_loss():
    """The scored-row compaction is sized from a host-side label count; fewer slots than labelled rows must not pass silently."""
    z = load_golden("clipcap_gpt2_mlp.npz")
    V = int(z["cfg"][0])
    ex = executor_from_config(os.path.join(GOLDEN, "hf_gpt2_tiny"), StubTokenizer(V, eos=V - 1, bos=V - 2), torch.float32)
    ex.model.train()
    n = int((T(z["labels"]) != -100).sum())
    kw = dict(question_tokens=T(z["ids"]), prefix=T(z["prefix"]), question_mask=T(z["mask"]), labels=T(z["labels"]).to(DEV))
    assert torch.isfinite(ex.model(label_count=n, **kw).loss).item()
    assert torch.isnan(ex.model(label_count=n - 2, **kw).loss).item()
    bad = T(z["labels"]).clone()
    bad[0, 0] = V + 5                                                      # a label beyond the vocabulary (torch asserts)
    assert torch.isnan(ex.model(**{**kw, "labels": bad}).loss).item()


def test_fewshot_generation_from_a_module_parser_batch():
    """The data path of BASELINE configs[3] end to end on the host side: `ModuleParser` few-shot modules (QInput + EmbeddingInput,
    module_parser.py:68-93,234-260) -> collate -> `generate_fewshot` with the sentinel id that `register_special_tokens` returns;
    ids equal the oracle's `insert_prefix_into_input` + greedy loop on the same batch (fp32, exact)."""
    import json
    from test_model_gpu import _oracle_fewshot_generate
    from test_module_parser import build_word_tokenizer
    from eavqa_amd.data.module_parser import VQA2Collator, make_sample, register_special_tokens
    from eavqa_amd.models.clipcap import ClipCaptionPrefix
    from eavqa_amd.models.lm import FrozenCausalLM, LMConfig, random_init_state_dict
    from eavqa_amd.utils.attrdict import AttrDict
    with open(os.path.join(GOLDEN, "module_parser.json")) as f:
        gold = json.load(f)
    case = next(c for c in gold["cases"] if c["name"] == "fewshot_2")
    tok = build_word_tokenizer(gold["words"])
    n_img = case["num_shots"] + 1
    special = register_special_tokens(tok, case["special_tokens"], num_sentinels=n_img)
    tok.pad_token = tok.eos_token
    cfg = AttrDict(data_loader=AttrDict(additional=AttrDict(case["additional"])), model_config=AttrDict(case["module_cfg"]))
    store = {k: torch.tensor(v) for k, v in gold["store"].items()}
    batch = VQA2Collator(cfg, tok)([make_sample(it, gold["examples"], store, case["num_shots"]) for it in gold["items"]])
    ids, mask, emb = batch["generative_input_ids"], batch["generative_attention_mask"], batch["clip_embeddings"]
    assert tuple(emb.shape) == (ids.shape[0], n_img, 1, 6)
    emb = torch.nn.functional.pad(emb, (0, 2))            # the fixture's 6-d toy embeddings -> 8 (GEMM operands need K % 4 == 0)
    B, D, L = ids.shape[0], emb.shape[-1], 3
    lcfg = LMConfig("opt", 2, 4, 64, 96, len(tok), 200, 1e-5, "relu", tok.eos_token_id, tok.pad_token_id)
    sd = random_init_state_dict(lcfg, 3, "cpu")
    lm = FrozenCausalLM(lcfg, sd, torch.float32, DEV)
    torch.manual_seed(1)
    model = ClipCaptionPrefix(prefix_length=L, prefix_size=D, mapping_type="mlp", lm=lm, dtype=torch.float32, device=DEV).eval()
    got = model.generate_fewshot(ids, emb, mask, num_shots=case["num_shots"], special_token_id=special, max_length=4,
                                 pad_token_id=tok.pad_token_id, eos_token_id=None)
    mapper = {k: v.detach().cpu() for k, v in model.clip_project.state_dict().items()}
    ocfg = dict(arch="opt", n_layer=2, n_head=4, act="relu")
    with torch.no_grad():
        want = _oracle_fewshot_generate(sd, ocfg, mapper, L, ids, emb.reshape(B, n_img, D), mask, n_img, special, 4, tok.pad_token_id, None)
    assert got == want
