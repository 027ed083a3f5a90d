"""CPU: the ModuleParser / collate mirror (eavqa_amd.data.module_parser) against batch dicts produced by RUNNING the reference's
``ModuleParser`` (tests/golden/module_parser.json, generator tests/golden/make_golden.py::module_parser_golden) with the same
offline-built HuggingFace tokenizer: training (QAInput + EmbeddingInput), few-shot and zero-shot generation (QInput +
EmbeddingInput), and permuted in-context examples."""
import json
import os

import pytest
import torch

from conftest import GOLDEN


def build_word_tokenizer(words, eos="</s>", pad=None):
    """The same offline tokenizer the golden generator builds (WordLevel, lower-cased, whitespace / punctuation split)."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"<unk>": 0, eos: 1}
    if pad:
        vocab[pad] = 2
    for w in words:
        vocab.setdefault(w, len(vocab))
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.normalizer = normalizers.Lowercase()
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    return PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", eos_token=eos, pad_token=pad)


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "module_parser.json")) as f:
        return json.load(f)


def collate(golden, case):
    from eavqa_amd.data.module_parser import VQA2Collator, make_sample, register_special_tokens
    from eavqa_amd.utils.attrdict import AttrDict
    tok = build_word_tokenizer(golden["words"])
    special = register_special_tokens(tok, case["special_tokens"], num_sentinels=case["n_sentinels"])
    tok.pad_token = tok.eos_token                                          # clipcap_exector.py:55
    cfg = AttrDict(data_loader=AttrDict(additional=AttrDict(case["additional"])), model_config=AttrDict(case["module_cfg"]))
    store = {k: torch.tensor(v) for k, v in golden["store"].items()}
    batch = [make_sample(it, golden["examples"], store, case["num_shots"]) for it in golden["items"]]
    return VQA2Collator(cfg, tok)(batch), tok, special


@pytest.mark.parametrize("name", ["train_qa", "fewshot_2", "zeroshot", "fewshot_3_perm2"])
def test_collate_matches_the_reference_module_parser(golden, name):
    case = next(c for c in golden["cases"] if c["name"] == name)
    out, tok, special = collate(golden, case)
    assert len(tok) == case["vocab_size"] and tok.bos_token_id == case["bos_token_id"] and tok.pad_token_id == case["pad_token_id"]
    if case["n_sentinels"]:
        assert special == case["sentinel_ids"][0]
        assert [tok.convert_tokens_to_ids(f"<extra_id_{i}>") for i in range(case["n_sentinels"])] == case["sentinel_ids"]
        assert case["sentinel_ids"] == [special - i for i in range(case["n_sentinels"])]      # what insert_prefix_into_input matches
    want = case["batch"]
    meta = {"question_ids", "questions", "answers", "gold_answers"}
    assert set(out) - meta == set(want)
    for k, v in want.items():
        got = out[k]
        if torch.is_tensor(got):
            assert got.tolist() == v, k
            assert got.dtype == (torch.float32 if k == "clip_embeddings" else torch.int64), (k, got.dtype)
        else:
            assert got == v, k
    assert out["question_ids"] == [it["question_id"] for it in golden["items"]]
    assert out["gold_answers"] == [it["gold_answer"] for it in golden["items"]]


def test_fewshot_batch_has_one_sentinel_per_image_in_order(golden):
    """The contract ``generate_fewshot`` / ``eavqa_build_fewshot_rows`` relies on: row b of ``generative_input_ids`` holds
    exactly n_img sentinel ids, ``special - i`` for the i-th image, and ``clip_embeddings`` is [B, n_img, 1, D]."""
    case = next(c for c in golden["cases"] if c["name"] == "fewshot_2")
    out, tok, special = collate(golden, case)
    ids = out["generative_input_ids"]
    n_img = case["num_shots"] + 1
    assert tuple(out["clip_embeddings"].shape) == (len(golden["items"]), n_img, 1, 6)
    for row in ids.tolist():
        found = [t for t in row if special - n_img < t <= special]
        assert found == [special - i for i in range(n_img)]


def test_register_special_tokens_rejects_increasing_sentinel_ids(golden):
    from eavqa_amd.data.module_parser import register_special_tokens
    tok = build_word_tokenizer(golden["words"])
    tok.add_special_tokens({"additional_special_tokens": ["<extra_id_0>", "<extra_id_1>"]})      # forward order: ids increase
    with pytest.raises(ValueError, match="decrease"):
        register_special_tokens(tok, {"additional_special_tokens": []}, num_sentinels=2)
