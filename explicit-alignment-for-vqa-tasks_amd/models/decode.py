"""Greedy generation for the prefix LM (``_generate_from_embeddings`` src/models/clipcap.py:387-471).

Two drivers produce the same token ids:
  * ``use_cache=False`` - the reference's algorithm verbatim: every step re-runs the whole, growing
    sequence (clipcap.py:414-419);
  * ``use_cache=True``  - prefill once, keep per-layer K/V in HBM ``[B, S_max, E]`` and run one
    token per step; a decode step is weight-streaming (HBM) bound.  An LM held in e4m3 streams half the bytes
    (``eavqa_lm_block_forward_fp8``; every Linear = row-quantised activations x e4m3 weights, as in the re-forward loop).
Token bookkeeping (argmax, pad for finished rows, eos flags) is one kernel per step
(``eavqa_greedy_pick``); the host reads the ``unfinished`` flags back once per step only to
honour the reference's early exit (clipcap.py:463).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .. import _lib, ops
from .lm import FrozenCausalLM

Tensor = torch.Tensor


def greedy_decode(lm: FrozenCausalLM, prefix_rows: Tensor, src: Tensor, mask: Tensor, pos: Tensor, B: int, S0: int,
                  max_length: int, pad_token_id: Optional[int], eos_token_id: Optional[int], use_cache: bool = True,
                  output_scores: bool = False, marks: Optional[list] = None):
    """``src/mask/pos``: int32 [B, S0 + max_length] for the whole horizon (appended positions have mask 1;
    their ``src`` entries are filled in as tokens are produced).  ``output_scores``: also return the float32
    [B, produced] log-probabilities of the raw greedy tokens (what HF's ``output_scores=True`` yields after
    ``log(softmax)``, few_shot_vqa_executor.py:301-314).  ``marks`` (bench instrumentation): a list that receives
    ``("prefill", event)`` and ``("decode", event)`` - HIP events recorded on the launch stream behind the prefill and behind the
    last decode step."""
    dev = lm.device
    S_max = S0 + max_length
    tokens = torch.zeros((B, max_length), dtype=torch.int64, device=dev)
    raw = torch.empty(B, dtype=torch.int32, device=dev)
    unfinished = torch.ones(B, dtype=torch.int32, device=dev)
    logp = torch.zeros((max_length, B), dtype=torch.float32, device=dev) if output_scores else None
    produced = 0
    alive = torch.zeros(max_length, dtype=torch.int32, device=dev)     # alive[t]: some row still unfinished after step t (set by the pick kernel)
    if use_cache:
        cache = _KVCache(lm, B, S_max, B * S0)
        logits = _prefill(lm, cache, prefix_rows, src[:, :S0].contiguous(), pos[:, :S0].contiguous(), mask, B, S0, S_max)
    _mark(marks, "prefill")
    for t in range(max_length):
        if not use_cache:
            S = S0 + t
            logits = lm.forward(prefix_rows, src[:, :S].contiguous(), pos[:, :S].contiguous(), mask[:, :S].contiguous(),
                                B, S, logits="last")["logits"]
        ops.greedy_pick(logits, lm.vocab, pad_token_id, eos_token_id, raw, tokens[:, t], unfinished,
                        logp[t] if output_scores else None, alive[t:t + 1] if eos_token_id is not None else None)
        produced = t + 1
        src[:, S0 + t] = raw                                   # the RAW argmax is what gets embedded (clipcap.py:423)
        if eos_token_id is not None:
            # clipcap.py:463 stops when every row has finished.  The host reads the flag every fourth step only (a device -> host round
            # trip per step would leave the launch queue empty while the next step is being enqueued); steps run past the stop emit pad
            # and are cut off below, so the ids are those of a check after every step.
            if (t + 1) % 4 == 0 and int(alive[t].item()) == 0:
                break
        if t + 1 < max_length and use_cache:
            logits = _decode_step(lm, cache, raw, pos[:, S0 + t].contiguous(), mask, B, S0 + t, S_max)
    _mark(marks, "decode")
    if eos_token_id is not None:
        dead = (alive[:produced] == 0).nonzero()
        if dead.numel():
            produced = int(dead[0].item()) + 1
    ids = tokens[:, :produced].cpu().numpy().astype(int).tolist()   # clipcap.py:469
    if output_scores:
        return ids, logp[:produced].t().contiguous().cpu()
    return ids


def _mark(marks: Optional[list], name: str) -> None:
    if marks is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append((name, ev))


class _KVCache:
    """Per-layer K/V ``[B, S_max, E]`` plus the host-side layer table and scratch for ``eavqa_lm_block_forward`` - or, for an LM held in e4m3
    (``weight_format="fp8"``), for ``eavqa_lm_block_forward_fp8``: the table then points at the weight BYTES and a second table carries the
    per-tensor scales."""

    def __init__(self, lm: FrozenCausalLM, B: int, S_max: int, max_rows: int):
        E, F = lm.cfg.n_embd, lm.cfg.ffn
        self.fp8 = getattr(lm, "weight_format", "native") == "fp8"
        self.k = [torch.empty((B * S_max, E), device=lm.device, dtype=lm.dtype) for _ in lm.layers]
        self.v = [torch.empty((B * S_max, E), device=lm.device, dtype=lm.dtype) for _ in lm.layers]
        self.table = (_lib.LMLayer * len(lm.layers))()
        self.scales = (_lib.LMLayerScales * len(lm.layers))() if self.fp8 else None
        for i, L in enumerate(lm.layers):
            t = self.table[i]
            for name in ("ln1_g", "ln1_b", "b_qkv", "b_o", "ln2_g", "ln2_b", "b_fc1", "b_fc2"):
                setattr(t, name, getattr(L, name).data_ptr())
            for name in ("w_qkv", "w_o", "w_fc1", "w_fc2"):
                w = getattr(L, name)
                setattr(t, name, (w.q if self.fp8 else w).data_ptr())
                if self.fp8:
                    setattr(self.scales[i], "s_" + name[2:], w.scale)
            t.k_cache, t.v_cache = self.k[i].data_ptr(), self.v[i].data_ptr()
        lib = _lib.load()
        # prefill (max_rows = B * S0 rows) and decode steps (B rows; their split-K partial sums) share one workspace
        if self.fp8:
            self.ws_bytes = max(int(lib.eavqa_lm_block_fp8_workspace_bytes(max_rows, E, F)), int(lib.eavqa_lm_block_fp8_workspace_bytes(B, E, F)))
        else:
            self.ws_bytes = max(int(lib.eavqa_lm_block_workspace_bytes(ops.dtype_id(lm.dtype), max_rows, E, F)),
                                int(lib.eavqa_lm_block_workspace_bytes(ops.dtype_id(lm.dtype), B, E, F)))
        self.ws = torch.empty(self.ws_bytes, device=lm.device, dtype=torch.uint8)


def _block(lm: FrozenCausalLM, cache: _KVCache, x: Tensor, mask: Tensor, B: int, Sq: int, row0: int, S_max: int) -> Tensor:
    """All decoder layers for ``Sq`` new positions per row starting at sequence index ``row0`` (one C call): K/V of the
    new positions are appended to the cache and attention runs against rows [0, row0+Sq).  ``x`` is updated in place."""
    c = lm.cfg
    if cache.fp8:
        _lib.call("eavqa_lm_block_forward_fp8", len(lm.layers), cache.table, cache.scales, c.n_embd, c.n_head, c.ffn, _lib.ACT[c.act], float(c.eps),
                  B, Sq, row0, S_max, x.data_ptr(), mask.data_ptr(), mask.stride(0), cache.ws.data_ptr(), cache.ws_bytes, ops._stream())
        return x
    args = (ops.dtype_id(lm.dtype), len(lm.layers), cache.table, c.n_embd, c.n_head, c.ffn, _lib.ACT[c.act], float(c.eps), B, Sq, row0, S_max,
            x.data_ptr(), mask.data_ptr(), mask.stride(0), cache.ws.data_ptr(), cache.ws_bytes, ops._stream())
    if ops.KernelSelect.decode_route:
        _lib.call("eavqa_lm_block_forward_ex", *args, ops.KernelSelect.decode_route)
    else:
        _lib.call("eavqa_lm_block_forward", *args)
    return x


def _last_logits(lm: FrozenCausalLM, x: Tensor, B: int, Sq: int) -> Tensor:
    E = lm.cfg.n_embd
    xl = x.view(B, Sq, E)[:, -1]
    hf = ops.layernorm_fwd(xl, lm.lnf_g, lm.lnf_b, lm.cfg.eps, lm.dtype)
    return lm._head(hf)


def _prefill(lm, cache, prefix_rows, src, pos, mask, B, S0, S_max) -> Tensor:
    x = ops.embed_assemble(src, pos, lm.wte, prefix_rows, lm.wpe)
    x = _block(lm, cache, x, mask, B, S0, 0, S_max)
    return _last_logits(lm, x, B, S0)


def _decode_step(lm, cache, raw, pos_col, mask, B, row0, S_max) -> Tensor:
    x = ops.embed_assemble(raw, pos_col, lm.wte, None, lm.wpe)
    x = _block(lm, cache, x, mask, B, 1, row0, S_max)
    return _last_logits(lm, x, B, 1)
