"""``VCT0Model`` / ``VCT0Prefix`` (src/models/vct0.py:301-549): CLIP embedding -> mapping network -> frozen T5 / T0 encoder-decoder.

Same class names, constructor keywords and method signatures as the reference, so that ``ModelClass(**model_args)`` of
src/trainers/vct0_exector.py:50-51 and src/trainers/few_shot_vqa_executor.py constructs it by name.  The LM is a
:class:`~eavqa_amd.models.t5.FrozenT5`; the mapper is the same hand-written MLP / TransformerMapper as ``ClipCaptionModel``'s.
``mapping_type="perceiver"`` (flamingo_pytorch, absent from the reference's own requirements pin and from this container) is not built.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn

from .. import ops
from .clipcap import MLP, TransformerMapper
from .lm import load_local_hf, synthetic_weights_notice
from .t5 import KNOWN_T5, FrozenT5, T5Config, random_init_t5_state_dict

Tensor = torch.Tensor


def _resolve_t5(model_version: str, dtype, device, seed: int = 2021) -> FrozenT5:
    """``AutoModelForSeq2SeqLM.from_pretrained(model_version)`` (vct0.py:312) without network access: a local HF directory is loaded; a
    known architecture name gets seeded random-init weights (synthetic runs - say so where results are reported)."""
    root = os.environ.get("EAVQA_MODEL_DIR", "")
    for cand in (model_version, os.path.join(root, model_version) if root else ""):
        if cand and os.path.isdir(cand) and os.path.exists(os.path.join(cand, "config.json")):
            cfgd, sd = load_local_hf(cand)
            return FrozenT5(T5Config.from_hf_dict(cfgd), sd, dtype, device)
    if model_version in KNOWN_T5:
        synthetic_weights_notice(model_version)
        cfg = T5Config.from_hf_dict(KNOWN_T5[model_version])
        lm = FrozenT5(cfg, random_init_t5_state_dict(cfg, seed, device), dtype, device)
        lm.synthetic_weights = True
        return lm
    raise FileNotFoundError(f"{model_version!r}: not a local HF directory and not a known architecture name; no network access is attempted")


class _Seq2SeqLoss(torch.autograd.Function):
    """loss = CE(T5(encoder <- mapper rows, decoder <- shift_right(labels))) with the frozen model's dgrad as backward."""

    @staticmethod
    def forward(ctx, rows, lm: FrozenT5, B, S, labels, holder):
        out = lm.forward_train(rows, B, S, labels)
        ctx.lm, ctx.tape = lm, out["tape"]
        holder.update(out)
        return out["loss"].view(())

    @staticmethod
    def backward(ctx, gloss):
        d = ctx.lm.backward(ctx.tape, gloss.reshape(1).to(torch.float32).contiguous())
        ctx.tape = None
        return (d, None, None, None, None, None)


class _Seq2SeqOutput:
    """``.loss`` / ``.logits`` [B, T, V] like HF's ``Seq2SeqLMOutput`` (what vct0_exector.py:143-146 reads)."""

    def __init__(self, loss, rows_logits, B, T, V):
        self.loss, self._lg, self._shape = loss, rows_logits, (B, T, V)

    @property
    def logits(self):
        B, T, V = self._shape
        return self._lg.view(B, T, -1)[:, :, :V]


class _GenerateOutput:
    """``return_dict_in_generate=True``: ``.sequences`` [B, len] int64, ``.scores`` tuple of per-step [B, V] float32 logits
    (few_shot_vqa_executor.py:301-314 reads exactly these two)."""

    def __init__(self, sequences, scores):
        self.sequences, self.scores = sequences, tuple(scores) if scores is not None else None


class VCT0Model(nn.Module):
    """``VCT0Model`` vct0.py:301-533."""

    def __init__(self, prefix_length: int, clip_length: Optional[int] = None, prefix_size: int = 512, num_layers: int = 8,
                 mapping_type: str = "mlp", model_version: str = "bigscience/T0_3B", *, lm: Optional[FrozenT5] = None,
                 dtype: torch.dtype = torch.bfloat16, device="cuda"):
        super().__init__()
        self.prefix_length = prefix_length
        self.dtype, self.device_ = dtype, torch.device(device)
        self.lm = lm if lm is not None else _resolve_t5(model_version, dtype, device)
        self.lm_embedding_size = self.lm.model_dim
        E = self.lm_embedding_size
        if mapping_type == "perceiver":
            raise NotImplementedError("mapping_type='perceiver' needs flamingo_pytorch (vct0.py:333-346), which is not part of this build")
        self.mapping_type = "transformer" if mapping_type == "transformer" else "mlp"      # unrecognised -> MLP (vct0.py:347-357)
        if self.mapping_type == "mlp":
            self.clip_project = MLP((prefix_size, (E * prefix_length) // 2, E * prefix_length), device=device, dtype=dtype)
        else:
            self.clip_project = TransformerMapper(prefix_size, E, prefix_length, clip_length, num_layers, device=device, dtype=dtype)

    def get_dummy_token(self, batch_size: int, num_question_tokens: int, device) -> Tensor:
        return torch.full((batch_size, self.prefix_length + num_question_tokens), -100, dtype=torch.int64, device=device)

    def _project(self, prefix: Tensor) -> Tensor:
        """``clip_project(prefix).view(-1, L, E)`` as rows [(image, l), E] in the compute dtype."""
        L, E = self.prefix_length, self.lm_embedding_size
        prefix = prefix.to(self.device_)
        if self.mapping_type == "mlp":
            return self.clip_project(prefix.reshape(-1, prefix.shape[-1])).reshape(-1, E)
        n = prefix.numel() // self.clip_project.linear.in_features
        stream = self.clip_project(prefix.reshape(n, -1))                                   # [n, CL + L, E]
        return stream[:, self.clip_project.clip_length:].reshape(-1, E).contiguous()

    # -- training forward (vct0.py:380-394) ---------------------------------------------------
    def forward(self, prefix: Tensor, labels: Optional[Tensor] = None):
        rows = self._project(prefix)
        B = rows.shape[0] // self.prefix_length
        if labels is None:
            raise ValueError("VCT0Model.forward needs labels (the decoder is teacher-forced on them, vct0.py:390-393)")
        holder: dict = {}
        lab = labels.to(self.device_).contiguous()
        loss = _Seq2SeqLoss.apply(rows, self.lm, B, self.prefix_length, lab, holder)
        return _Seq2SeqOutput(loss, holder["logits"], B, lab.shape[1], self.lm.cfg.vocab)

    # -- generation (vct0.py:396-491) -----------------------------------------------------------
    def _encode_interleaved(self, tok: Tensor, qm: Tensor, rows: Tensor, n_img: int, special_token_id: int):
        """``insert_prefix_into_input`` (:494-533) + encoder: (encoder output rows, mask [B, S'], S')."""
        L = self.prefix_length
        B, T = tok.shape
        src, mask, _, status = ops.build_fewshot_rows(tok, qm.to(torch.int64), L, n_img, special_token_id, 0)
        if not bool((status == n_img).all().item()):
            raise ValueError("every row must hold exactly one sentinel token per image")   # the reference's .view at :512 fails
        S = T + (L - 1) * n_img
        x = ops.embed_assemble(src.reshape(-1), None, self.lm.shared, rows, None)
        enc, _ = self.lm.encode(x, mask, B, S)
        return enc, mask, S

    @torch.no_grad()
    def generate(self, prefix: Tensor, question_tokens: Optional[Tensor] = None, question_mask: Optional[Tensor] = None,
                 decoder_input_ids: Optional[Tensor] = None, decoder_attention_mask: Optional[Tensor] = None, no_prefix: Optional[bool] = False,
                 pass_examples_through_encoder_one_at_a_time: Optional[bool] = False, num_shots: Optional[int] = None,
                 special_token_id: int = 32099, max_length: int = 20, output_scores: bool = False, return_dict_in_generate: bool = False,
                 use_cache: bool = True, **generation_kwargs):
        """Greedy generation (HF defaults of ``lm.generate``; ``max_length`` counts the decoder start token).  ``special_token_id`` is an
        addition: the reference hard-codes T5's 32099; ``use_cache`` (HF's name and default): decoder steps against a self-attention K / V cache.  """
        dev, lm, L = self.device_, self.lm, self.prefix_length
        unsupported = {k: v for k, v in generation_kwargs.items() if k not in ("bos_token_id", "do_sample", "num_beams") or (k == "num_beams" and v != 1)
                       or (k == "do_sample" and v)}
        if unsupported:
            raise NotImplementedError(f"greedy search only; unsupported generation arguments: {sorted(unsupported)}")
        finish = lambda seq, scores: _GenerateOutput(seq, scores) if return_dict_in_generate else seq
        tok = question_tokens.to(dev) if question_tokens is not None else None
        qm = question_mask.to(dev) if question_mask is not None else (torch.ones_like(tok) if tok is not None else None)
        if no_prefix:
            if pass_examples_through_encoder_one_at_a_time:
                raise NotImplementedError("text-only generation one example at a time (vct0.py:411-419) is not built")
            B, T = tok.shape
            enc, _ = lm.encode(lm.embed(tok), qm.to(torch.int32).contiguous(), B, T)
            return finish(*lm.greedy(enc, qm.to(torch.int32).contiguous(), B, T, max_length, output_scores=output_scores, use_cache=use_cache))
        if tok is None:                                                    # prefix only (:485-491)
            rows = self._project(prefix)
            B = rows.shape[0] // L
            mask = torch.ones((B, L), device=dev, dtype=torch.int32)
            src = -(torch.arange(B * L, device=dev, dtype=torch.int32) + 1)
            enc, _ = lm.encode(ops.embed_assemble(src, None, lm.shared, rows, None), mask, B, L)
            return finish(*lm.greedy(enc, mask, B, L, max_length, output_scores=output_scores, use_cache=use_cache))
        B = tok.shape[0]
        prefix = prefix.to(dev).reshape(B, -1, prefix.shape[-1])
        n_img = prefix.shape[1]
        rows = self._project(prefix)                                       # [(b, n, l), E]
        if pass_examples_through_encoder_one_at_a_time:                    # :426-442: tokens [B, n, T1], example i carries sentinel special - i
            E = self.lm_embedding_size
            r4 = rows.view(B, n_img, L, E)
            encs, masks = [], []
            for i in range(n_img):
                enc_i, m_i, S_i = self._encode_interleaved(tok[:, i].contiguous(), qm[:, i].contiguous(), r4[:, i].reshape(-1, E).contiguous(), 1,
                                                           special_token_id - i)
                encs.append(enc_i.view(B, S_i, E))
                masks.append(m_i)
            enc = torch.cat(encs, dim=1)
            mask = torch.cat(masks, dim=1).contiguous()
            S = enc.shape[1]
            return finish(*lm.greedy(enc.reshape(B * S, E).contiguous(), mask, B, S, max_length, output_scores=output_scores, use_cache=use_cache))
        if decoder_input_ids is not None:                                  # :468-480: only the query image, the decoder continues a prompt
            enc, mask, S = self._encode_interleaved(tok, qm, rows.view(B, n_img, L, -1)[:, -1].reshape(B * L, -1).contiguous(), 1, special_token_id)
            seq, scores = lm.greedy(enc, mask, B, S, max_length, dec_prompt=decoder_input_ids, output_scores=output_scores, use_cache=use_cache,
                                    dec_mask=decoder_attention_mask)
            # (the reference slices by the prompt length it was GIVEN: when HF prepended the start token the prompt's last token stays in)
            return finish(seq[:, decoder_input_ids.shape[1]:], scores)
        ns = (n_img - 1) if not num_shots else num_shots
        enc, mask, S = self._encode_interleaved(tok, qm, rows, ns + 1, special_token_id)
        return finish(*lm.greedy(enc, mask, B, S, max_length, output_scores=output_scores, use_cache=use_cache))


class VCT0Prefix(VCT0Model):
    """``VCT0Prefix`` vct0.py:536-549: only the mapper trains; the LM is frozen by construction (it holds no torch parameters)."""

    def parameters(self, recurse: bool = True):
        return self.clip_project.parameters()

    def train(self, mode: bool = True):
        super().train(mode)
        return self
