"""``VCT0Executor`` and ``FewShotVQAExecutor``: the reference's Lightning executor surface for the T5 / T0 path
(src/trainers/vct0_exector.py, src/trainers/few_shot_vqa_executor.py) over the HIP model classes.

Kept from the reference: construction by name lookup (``ModelClass(**model_args)``, vct0_exector.py:50-51 /
few_shot_vqa_executor.py:52-53), ``tokenizer.bos_token = tokenizer.pad_token`` (vct0_exector.py:53), ``training_step`` =
``model(prefix=clip_embeddings, labels=labels).loss`` (:132-167), the CC ``_generative_step`` (loss + prefix-only generate,
:184-263), and the few-shot ``_generative_step`` with its batch reshapes for one-example-at-a-time encoding, one-shot
ensembles and permutation ensembles (few_shot_vqa_executor.py:158-210) + ``generate_from_ensembles`` (:293-332: log-softmax of
the per-step scores summed over the emitted tokens not in [0, 1, 2], best member per question).
pytorch_lightning is absent offline: plain classes with Lightning's method names; ``fit`` comes from ``ClipCapExecutor``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..models.vct0 import VCT0Model, VCT0Prefix  # noqa: F401  (looked up by name)
from .clipcap_executor import ClipCapExecutor


class VCT0Executor(ClipCapExecutor):
    """Mapper training on Conceptual-Captions-shaped batches through the frozen T5 / T0 (vct0_exector.py)."""

    def __init__(self, config, data_loader=None, *, vision_encoder=None, model=None, dtype=torch.bfloat16, device="cuda"):
        self.config = config
        self.data_loader = data_loader
        self.device = torch.device(device)
        self.tokenizer = getattr(data_loader, "tokenizer", None)
        self.decoder_tokenizer = getattr(data_loader, "decoder_tokenizer", self.tokenizer)
        if model is None:
            ModelClass = globals()[self.config.model_config.ModelClass]                     # vct0_exector.py:50
            model = ModelClass(**dict(self.config.model_config.model_args), dtype=dtype, device=device)   # :51
        self.model = model
        self.vision_encoder = vision_encoder
        if self.tokenizer is not None:
            self.tokenizer.bos_token = self.tokenizer.pad_token                             # :53
        self.global_step = 0
        self.logged = {}
        self.optimizer = None
        self.scheduler = None
        self.grad_sync = None

    def training_step(self, sample_batched, batch_idx):
        """vct0_exector.py:132-167."""
        out = self.model(prefix=self._clip_embeddings(sample_batched), labels=sample_batched["labels"].to(self.device))
        for i, lr in enumerate(self.scheduler.get_last_lr() if self.scheduler else []):
            self.log(f"train/lr[{i}]", lr, prog_bar=True, on_step=True, logger=True)
        self.log("train/loss", out.loss, on_step=True, on_epoch=True, logger=True)
        return {"loss": out.loss}

    def _generative_step(self, sample_batched, batch_idx):
        """vct0_exector.py:184-263: the test loss of the batch, then (first six batches only) captions from the prefix alone."""
        prefix = self._clip_embeddings(sample_batched)
        with torch.no_grad():
            loss = self.model(prefix=prefix, labels=sample_batched["labels"].to(self.device)).loss
        self.log("test/loss", loss, on_step=True, on_epoch=True, logger=True)
        if batch_idx > 5:
            return None
        outputs = self.model.generate(prefix=prefix, max_length=self.config.data_loader.additional.max_target_length,
                                      bos_token_id=getattr(self.tokenizer, "bos_token_id", None))
        predictions = []
        bos = getattr(self.decoder_tokenizer, "bos_token_id", None)
        for index, seq in enumerate(outputs.tolist()):
            if bos is not None and bos in seq:
                seq = seq[seq.index(bos):]
            decoded = self.decoder_tokenizer.decode(seq, skip_special_tokens=True) if self.decoder_tokenizer is not None else seq
            predictions.append({"image_url": (sample_batched.get("image_urls") or [None] * len(outputs))[index], "caption": decoded})
        return {"predictions": predictions, "outputs": outputs, "loss": loss}


class FewShotVQAExecutor(VCT0Executor):
    """Few-shot VQA2 inference through ``VCT0Model.generate`` (few_shot_vqa_executor.py)."""

    def training_step(self, sample_batched, batch_idx):
        return None                                                                         # few_shot_vqa_executor.py:139-140

    def _generative_step(self, sample_batched, batch_idx):
        """few_shot_vqa_executor.py:158-210."""
        add = self.config.data_loader.additional
        ids = sample_batched["generative_input_ids"].to(self.device)
        mask = sample_batched["generative_attention_mask"].to(self.device)
        emb = sample_batched["clip_embeddings"].to(self.device)
        max_length = add.max_target_length
        dec_ids = dec_mask = None
        if "decoder_generative_input_ids" in sample_batched:
            dec_ids = sample_batched["decoder_generative_input_ids"][:, :-1].to(self.device)
            dec_mask = sample_batched["decoder_generative_attention_mask"][:, :-1].to(self.device)
        one_at_a_time = bool(add.get("pass_examples_through_encoder_one_at_a_time", False))
        no_prefix = bool(add.get("no_prefix", False))
        sentinel = add.get("special_token_id", 32099)
        if one_at_a_time:
            ids = ids.view(-1, add.num_shots + 1, ids.shape[-1])
            mask = mask.view(-1, add.num_shots + 1, mask.shape[-1])
        if add.get("ensemble_one_shots", False):
            ids = ids.view(-1, add.num_shots, ids.shape[-1])
            mask = mask.view(-1, add.num_shots, mask.shape[-1])
            outputs = self.generate_from_ensembles(ids, mask, emb, add.num_shots, max_length, num_shots=1, one_shots=True, sentinel=sentinel,
                                                   no_prefix=no_prefix, one_at_a_time=one_at_a_time)
        elif add.get("num_permutations_of_in_context_examples", 0) > 0:
            n = add.num_permutations_of_in_context_examples
            ids = ids.view(-1, n, ids.shape[-1])
            mask = mask.view(-1, n, mask.shape[-1])
            outputs = self.generate_from_ensembles(ids, mask, emb, n, max_length, sentinel=sentinel, no_prefix=no_prefix,
                                                   one_at_a_time=one_at_a_time)
        else:
            outputs = self.model.generate(question_tokens=ids, question_mask=mask, prefix=emb, decoder_input_ids=dec_ids,
                                          decoder_attention_mask=dec_mask, no_prefix=no_prefix,
                                          pass_examples_through_encoder_one_at_a_time=one_at_a_time, max_length=max_length, special_token_id=sentinel)
        predictions = []
        for index, seq in enumerate(outputs):
            seq = [int(t) for t in (seq.tolist() if torch.is_tensor(seq) else seq)]
            decoded = self.decoder_tokenizer.decode(seq, skip_special_tokens=True) if self.decoder_tokenizer is not None else seq
            qid = sample_batched["question_ids"][index] if "question_ids" in sample_batched else index
            predictions.append({"question_id": qid, "answer": decoded})
        return {"predictions": predictions, "outputs": outputs, "question_ids": sample_batched.get("question_ids"),
                "answers": sample_batched.get("answers")}

    def generate_from_ensembles(self, ids, mask, emb, num_ensembles: int, max_length: int, num_shots: Optional[int] = None, one_shots: bool = False,
                                sentinel: int = 32099, no_prefix: bool = False, one_at_a_time: bool = False):
        """few_shot_vqa_executor.py:293-332: one greedy generation per ensemble member; a sequence's score is the sum over its emitted
        tokens not in [0, 1, 2] of log softmax(step scores)[token] (token k is scored by step k - 1: the start token has no score);
        ``np.argmax`` keeps the first best member.  ``no_prefix`` / ``pass_examples_through_encoder_one_at_a_time`` travel to
        ``model.generate`` as the reference forwards them (:304-314)."""
        B = ids.shape[0]
        batch_scores = np.zeros((B, num_ensembles))
        members = []
        for i in range(num_ensembles):
            clip = emb[:, [i, -1]] if one_shots else emb[:, i]                              # :298-302
            out = self.model.generate(question_tokens=ids[:, i].contiguous(), question_mask=mask[:, i].contiguous(), prefix=clip, num_shots=num_shots,
                                      no_prefix=no_prefix, pass_examples_through_encoder_one_at_a_time=one_at_a_time,
                                      max_length=max_length, output_scores=True, return_dict_in_generate=True, special_token_id=sentinel)
            logp = torch.log(torch.stack(list(out.scores)).softmax(dim=-1))                 # [steps, B, V] (host tensors)
            for j, seq in enumerate(out.sequences.tolist()):
                s = 0.0
                for k, tok in enumerate(seq):
                    if tok not in (0, 1, 2):
                        s += float(logp[k - 1, j, tok])
                batch_scores[j, i] = s
            members.append(out.sequences)
        best = np.argmax(batch_scores, axis=1)
        return [members[ind][j] for j, ind in enumerate(best)]
