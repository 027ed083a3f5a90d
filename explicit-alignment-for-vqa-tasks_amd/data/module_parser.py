"""Host-side mirror of the reference's ``ModuleParser`` input / output modules and of the dataset's ``collate_fn`` for the
hot path's boundary: the code that turns samples into the batch dict the executor and
``ClipCaptionModel.forward / generate / generate_fewshot`` consume.

Reference (``src/data_loader_manager/module_parser.py``): ``QAInput`` :47-65, ``QInput`` :68-93, ``TestInput`` :95-109,
``EmbeddingInput`` :234-260, ``GenerationOutput`` :275-286, ``parse_modules`` :320-364, ``DefaultProcessing`` :366-385,
``PostProcessInputTokenization`` :387-449, ``PostProcessClipEmbeddings`` :466-478, ``PostProcessOutputTokenization`` :504-563,
``post_processing`` :582-604; collate ``src/data_loader_manager/datasets/vqa2_datasets.py:94-181``; sample assembly (in-context
examples + their CLIP embeddings, query last) ``vqa2_datasets.py:65-91``; special-token registration
``src/data_loader_manager/data_loader_wrapper.py:57-62``.

Same method names, ``module`` dict shapes (``type / option / separation_tokens``) and output keys, dispatched by name like the
reference (``getattr(self, module.type)``).  Pure host code: strings, a HuggingFace-API tokenizer supplied by the caller, and
``torch.stack`` of the stored embeddings - no kernel and no arithmetic on the hot path.  Pinned by
``tests/golden/module_parser.json`` (produced by running the reference's ``ModuleParser`` with the same tokenizer).

Sentinel tokens for causal LMs.  The prompt formatter announces image i with the string ``<extra_id_i>``
(``src/utils/in_context_examples.py:116``).  T5 vocabularies contain those strings with DECREASING ids (``<extra_id_0>`` =
32099, ``<extra_id_1>`` = 32098, ...), which is what ``insert_prefix_into_input`` matches (``special_token_id - i``,
src/models/vct0.py:503-507) and what ``eavqa_build_fewshot_rows`` implements.  GPT-2 / OPT vocabularies have no such tokens:
:func:`register_special_tokens` adds them as ``additional_special_tokens`` in REVERSE order (``<extra_id_{n-1}>`` first) so that
their ids decrease with i exactly like T5's, and returns ``special_token_id`` = id of ``<extra_id_0>``.
"""
from __future__ import annotations

import random
from typing import Any, Dict, List, Optional

import torch

from ..utils.attrdict import AttrDict
from ..utils.in_context_examples import InContextExampleFormatter


def register_special_tokens(tokenizer, special_tokens: Dict[str, Any], num_sentinels: int = 0) -> Optional[int]:
    """``data_loader_wrapper.py:57-62``: keep the tokenizer's own ``additional_special_tokens``, append the config's, call
    ``add_special_tokens`` (the caller then runs ``model.gpt.resize_token_embeddings(len(tokenizer))``, clipcap_exector.py:56).
    ``num_sentinels`` > 0 (causal LMs, see the module docstring) also registers ``<extra_id_{n-1}> ... <extra_id_0>`` unless the
    vocabulary already has them (T5) and returns the id of ``<extra_id_0>``; otherwise returns None."""
    st = dict(special_tokens)
    own = getattr(tokenizer, "additional_special_tokens", None) or getattr(tokenizer, "extra_special_tokens", None) or []   # renamed in transformers 5
    extra = list(own) + list(st.get("additional_special_tokens", []))
    if num_sentinels > 0:
        vocab = tokenizer.get_vocab()
        for i in reversed(range(num_sentinels)):
            tok = InContextExampleFormatter.image_token.format(i)
            if tok not in vocab and tok not in extra:
                extra.append(tok)
    st["additional_special_tokens"] = extra
    tokenizer.add_special_tokens(st)
    if num_sentinels > 0:
        ids = [tokenizer.convert_tokens_to_ids(InContextExampleFormatter.image_token.format(i)) for i in range(num_sentinels)]
        if any(ids[i] != ids[0] - i for i in range(num_sentinels)):
            raise ValueError(f"sentinel ids {ids} do not decrease by one per image: insert_prefix_into_input (vct0.py:503-507) "
                             "matches special_token_id - i")
        return ids[0]
    return None


class ModuleParser:
    """Mixin with the reference's module methods; needs ``self.config``, ``self.tokenizer``, ``self.decoder_tokenizer``."""

    # ---------------------------------------------------------------- sample-level sub parsers
    def _additional(self):
        return self.config.data_loader.additional

    def _formatter(self, module) -> InContextExampleFormatter:
        a = self._additional()
        return InContextExampleFormatter(format_type=module["option"],
                                         pass_examples_through_encoder_one_at_a_time=a.get("pass_examples_through_encoder_one_at_a_time", 0),
                                         sample_templates=a.get("sample_templates", 0), ensemble_one_shots=a.get("ensemble_one_shots", 0))

    def QuestionInput(self, sample, module):
        """module_parser.py:29-45."""
        sep = module["separation_tokens"]
        return AttrDict(text_sequence=" ".join([sep["start"]] + [sample["question"]] + [sep["end"]]))

    def QAInput(self, sample, module):
        """module_parser.py:47-65: ``start question end <BOS> answer <EOS>`` - the training sequence of the causal path."""
        sep = module["separation_tokens"]
        return AttrDict(text_sequence=" ".join([sep["start"]] + [sample["question"]] + [sep["end"]] + [self.tokenizer.bos_token]
                                               + [sample["gold_answer"]] + [self.tokenizer.eos_token]))

    def QInput(self, sample, module):
        """module_parser.py:68-93: the few-shot prompt (``InContextExampleFormatter``), optionally one prompt per random
        permutation of the in-context examples (``random.seed(2022)`` per sample, :80)."""
        f = self._formatter(module)
        n_perm = self._additional().get("num_permutations_of_in_context_examples", 0)
        if n_perm > 0:
            random.seed(2022)
            text = [f.format_input(random.sample(sample["in_context_examples"], k=len(sample["in_context_examples"])), sample)
                    for _ in range(n_perm)]
        else:
            text = f.format_input(sample["in_context_examples"], sample)
        return AttrDict(text_sequence=text)

    def TestInput(self, sample, module):
        """module_parser.py:95-109: the zero-shot prompt."""
        return AttrDict(text_sequence=self._formatter(module).format_input([], sample))

    def EmbeddingInput(self, sample, module):
        """module_parser.py:234-260: the sample's stored CLIP embeddings (in-context images first, query image last,
        vqa2_datasets.py:76-79), each ``[1, D]`` -> ``[n_img, 1, D]``; with permutations ``[n_perm, n_img, D]`` using the same
        seeded shuffles as ``QInput``."""
        n_perm = self._additional().get("num_permutations_of_in_context_examples", 0)
        embs = sample["clip_embedding"]
        if n_perm > 0:
            ctx = embs[:-1]
            random.seed(2022)
            perms = [[*random.sample(ctx, k=len(ctx)), embs[-1]] for _ in range(n_perm)]
            t = torch.stack([torch.as_tensor(e) for p in perms for e in p])
            return AttrDict(clip_embedding=t.view(n_perm, len(embs), t.shape[-1]))
        return AttrDict(clip_embedding=torch.stack([torch.as_tensor(e) for e in embs]))

    def GenerationOutput(self, sample, module):
        """module_parser.py:275-286."""
        return AttrDict(text_sequence=sample["gold_answer"])

    # ---------------------------------------------------------------- aggregation
    def parse_modules(self, sample, modules, type: str, process_modules=None):
        """module_parser.py:320-364."""
        if type not in ("input", "decoder_input", "output"):
            raise ValueError("Unknown type: {}".format(type))
        data_collection = [getattr(self, m["type"])(sample, m) for m in modules]
        if process_modules is None:
            return self.DefaultProcessing(data_collection)
        processed = data_collection
        for pm in process_modules:
            processed = getattr(self, pm["type"])(processed)
        return processed

    def DefaultProcessing(self, data_to_process):
        """module_parser.py:366-385: strings under one key are joined by a space; anything else may appear once."""
        out = AttrDict()
        for entry in data_to_process:
            for key, value in entry.items():
                if key not in out:
                    out[key] = value
                elif isinstance(value, str):
                    out[key] += " " + value
                else:
                    raise TypeError("Undefined processing type: {}".format(type(value)))
        return out

    # ---------------------------------------------------------------- batch-level post-processing
    def PostProcessInputTokenization(self, data_to_process, module):
        """module_parser.py:387-449: tokenise (padding ``longest``, right side; left + ``<pad>`` prefix for
        ``decoder_generation``), keys ``input_ids / attention_mask`` (option default), ``generative_*`` (generation),
        ``decoder_generative_*`` (decoder_generation)."""
        text_sequences = data_to_process.pop("text_sequence")
        a = self._additional()
        task_prefix = ""
        if module["option"] == "decoder_generation":
            self.tokenizer.padding_side = "left"
            task_prefix = "<pad>"
        nested = (a.get("pass_examples_through_encoder_one_at_a_time", 0) or a.get("num_permutations_of_in_context_examples", 0) > 0
                  or a.get("ensemble_one_shots", 0))
        texts = ([example for sequence in text_sequences for example in sequence] if nested
                 else [task_prefix + sequence for sequence in text_sequences])
        encoding = self.tokenizer(texts, padding="longest", max_length=a["max_source_length"], truncation=True, return_tensors="pt")
        self.tokenizer.padding_side = "right"
        if module["option"] == "generation":
            for key, value in encoding.items():
                data_to_process[f"generative_{key}"] = value
            data_to_process["generative_text_sequences"] = text_sequences
        elif module["option"] == "decoder_generation":
            for key, value in encoding.items():
                data_to_process[f"decoder_generative_{key}"] = value
            data_to_process["decoder_generative_text_sequences"] = text_sequences
        else:
            data_to_process.update({**encoding, "input_text_sequences": text_sequences})
        return data_to_process

    def PostProcessClipEmbeddings(self, data_to_process, module):
        """module_parser.py:466-478: ``clip_embedding`` (list of per-sample stacks) -> ``clip_embeddings`` ``[B, n_img, 1, D]``."""
        data_to_process["clip_embeddings"] = torch.stack(data_to_process.pop("clip_embedding"))
        return data_to_process

    def PostProcessOutputTokenization(self, data_to_process, module):
        """module_parser.py:504-563: targets tokenised with the decoder tokenizer; in ``labels`` the FIRST pad of a row stays
        (it is the eos when pad == eos), later pads become -100."""
        text_sequences = data_to_process.pop("text_sequence")
        tok = self.decoder_tokenizer
        enc = tok(text_sequences, padding="longest", max_length=self._additional()["max_target_length"], truncation=True)
        ids = enc["input_ids"]
        rows = []
        for row in ids:
            seen, out = False, []
            for label in row:
                if label == tok.pad_token_id:
                    if seen:
                        label = -100
                    seen = True
                out.append(label)
            rows.append(out)
        labels = torch.LongTensor(rows)
        output_sequence_ids = torch.LongTensor(ids)
        assert labels.shape == output_sequence_ids.shape
        data_to_process.update({"labels": labels, "output_sequence_ids": output_sequence_ids,
                                "output_sequence_attention_mask": torch.LongTensor(enc["attention_mask"]),
                                "output_text_sequences": text_sequences})
        return data_to_process

    def post_processing(self, processed_batch_data, postprocess_modules=None):
        """module_parser.py:582-604."""
        if postprocess_modules is None:
            return processed_batch_data
        for pm in postprocess_modules:
            processed_batch_data = getattr(self, pm["type"])(processed_batch_data, pm)
        return processed_batch_data


def make_sample(item, in_context_examples: List, clip_embeddings: Dict[str, Any], num_shots: int) -> AttrDict:
    """``VQA2Dataset.__getitem__`` vqa2_datasets.py:65-91: the LAST ``num_shots`` retrieved examples, their stored CLIP
    embeddings (``{img_key: float32[1, D]}``, extract_contrastive_image_embeddings.py:44-72) followed by the query image's."""
    ctx = [] if num_shots == 0 else list(in_context_examples)[-num_shots:]
    g = (lambda o, k: o[k] if isinstance(o, dict) else getattr(o, k))
    embs = [clip_embeddings.get(str(g(e, "img_key"))) for e in ctx] + [clip_embeddings.get(str(g(item, "img_key")))]
    return AttrDict(question_id=g(item, "question_id"), question=g(item, "question"), gold_answer=g(item, "gold_answer"),
                    answers=g(item, "answers"), clip_embedding=embs, in_context_examples=ctx)


class VQA2Collator(ModuleParser):
    """``VQA2Dataset.collate_fn`` vqa2_datasets.py:94-181 over the module lists of ``config.model_config``."""

    def __init__(self, config, tokenizer, decoder_tokenizer=None):
        self.config = config
        self.tokenizer = tokenizer
        self.decoder_tokenizer = decoder_tokenizer if decoder_tokenizer is not None else tokenizer

    def __call__(self, batch: List) -> AttrDict:
        mc = self.config.model_config
        groups = (("input", mc["input_modules"]), ("decoder_input", mc["decoder_input_modules"]), ("output", mc["output_modules"]))
        out = AttrDict(question_ids=[s["question_id"] for s in batch], questions=[s["question"] for s in batch],
                       answers=[s["answers"] for s in batch], gold_answers=[s["gold_answer"] for s in batch])
        for kind, spec in groups:
            data: Dict[str, list] = {}
            for sample in batch:
                for key, value in self.parse_modules(sample, spec["module_list"], type=kind).items():
                    data.setdefault(key, []).append(value)
            out.update(self.post_processing(AttrDict(data), spec["postprocess_module_list"] or None))
        return out
