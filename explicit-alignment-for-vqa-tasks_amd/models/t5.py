"""Frozen T5 / T0 encoder-decoder executed by the HIP kernels (the LM of ``VCT0Model``, src/models/vct0.py:301-491).

Arithmetic mirrored (HF ``T5ForConditionalGeneration``, transformers/models/t5/modeling_t5.py):
  * ``T5LayerNorm`` :50-72 (RMS norm, no bias)                         -> ``eavqa_rmsnorm_fwd / _bwd``
  * ``T5Attention`` :176-369: q / k / v / o without bias, scores NOT scaled, the relative-position bias of the
    stack's first block (:217-279) added to every layer's scores, additive key mask -> ``eavqa_attention_fwd_rel / _bwd_rel``
    (the bias depends on key - query only, so it travels as a per-head table over offsets)
  * ``T5DenseGatedActDense`` :97-123 (v1.1 / T0: gelu_new(wi_0 x) * wi_1 x; one GEMM against [wi_0; wi_1] + ``eavqa_gated_act``),
    ``T5DenseActDense`` :75-94 (v1.0: ReLU in the GEMM epilogue)
  * ``T5Stack`` :640-751, ``_shift_right`` :618-637, tied-embedding rescale + lm_head + CE :1040-1056.
The model is frozen (``VCT0Prefix`` trains the mapper only, vct0.py:536-549): the backward pass is dgrad only, through the decoder,
its cross-attention into the encoder output, and the encoder, down to the rows the mapper produced.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from .. import _lib, ops

Tensor = torch.Tensor


@dataclass
class T5Config:
    d_model: int
    d_kv: int
    n_head: int
    d_ff: int
    n_layer: int
    n_dec_layer: int
    vocab: int
    gated: bool = True               # feed_forward_proj == "gated-gelu" (T5 v1.1 / T0); False: "relu" (v1.0)
    tied: bool = False               # tie_word_embeddings: lm_head = shared, decoder output scaled by d_model^-0.5
    eps: float = 1e-6
    num_buckets: int = 32
    max_distance: int = 128
    decoder_start_token_id: int = 0
    pad_token_id: int = 0
    eos_token_id: int = 1

    @property
    def inner(self) -> int:
        return self.n_head * self.d_kv

    @property
    def act(self) -> str:
        return "gelu_new" if self.gated else "relu"

    @staticmethod
    def from_hf_dict(d: dict) -> "T5Config":
        ff = d.get("feed_forward_proj", "relu")
        if ff not in ("relu", "gated-gelu"):
            raise NotImplementedError(f"feed_forward_proj {ff!r}")
        return T5Config(d["d_model"], d["d_kv"], d["num_heads"], d["d_ff"], d["num_layers"], d.get("num_decoder_layers") or d["num_layers"],
                        d["vocab_size"], ff == "gated-gelu", bool(d.get("tie_word_embeddings", True)), d.get("layer_norm_epsilon", 1e-6),
                        d.get("relative_attention_num_buckets", 32), d.get("relative_attention_max_distance", 128),
                        d.get("decoder_start_token_id", 0) or 0, d.get("pad_token_id", 0) or 0, d.get("eos_token_id", 1))


# public architecture constants (no weights are fetched): bigscience/T0_3B is T5-XL v1.1 LM-adapted
KNOWN_T5 = {
    "bigscience/T0_3B": dict(d_model=2048, d_kv=64, num_heads=32, d_ff=5120, num_layers=24, vocab_size=32128, feed_forward_proj="gated-gelu",
                             tie_word_embeddings=False),
    "t5-small": dict(d_model=512, d_kv=64, num_heads=8, d_ff=2048, num_layers=6, vocab_size=32128, feed_forward_proj="relu", tie_word_embeddings=True),
}


def random_init_t5_state_dict(cfg: T5Config, seed: int = 2021, device="cpu") -> Dict[str, Tensor]:
    """Seeded random-init weights under HF key names (synthetic runs: no pretrained weights exist offline)."""
    g = torch.Generator(device=device).manual_seed(seed)
    E, I, F = cfg.d_model, cfg.inner, cfg.d_ff
    n = lambda *shape, std=0.02: torch.randn(*shape, generator=g, device=device) * std
    sd = {"shared.weight": n(cfg.vocab, E, std=1.0)}
    if not cfg.tied:
        sd["lm_head.weight"] = n(cfg.vocab, E, std=E ** -0.5)

    def attn(p):
        sd[p + "q.weight"], sd[p + "k.weight"] = n(I, E, std=(E * cfg.d_kv) ** -0.5), n(I, E, std=E ** -0.5)
        sd[p + "v.weight"], sd[p + "o.weight"] = n(I, E, std=E ** -0.5), n(E, I, std=I ** -0.5)

    def ffn(p):
        if cfg.gated:
            sd[p + "wi_0.weight"], sd[p + "wi_1.weight"] = n(F, E, std=E ** -0.5), n(F, E, std=E ** -0.5)
        else:
            sd[p + "wi.weight"] = n(F, E, std=E ** -0.5)
        sd[p + "wo.weight"] = n(E, F, std=F ** -0.5)

    for stack, nl, dec in (("encoder", cfg.n_layer, False), ("decoder", cfg.n_dec_layer, True)):
        sd[f"{stack}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] = n(cfg.num_buckets, cfg.n_head, std=0.5)
        for i in range(nl):
            p = f"{stack}.block.{i}.layer."
            attn(p + "0.SelfAttention.")
            sd[p + "0.layer_norm.weight"] = torch.ones(E, device=device)
            if dec:
                attn(p + "1.EncDecAttention.")
                sd[p + "1.layer_norm.weight"] = torch.ones(E, device=device)
            ffn(p + ("2." if dec else "1.") + "DenseReluDense.")
            sd[p + ("2." if dec else "1.") + "layer_norm.weight"] = torch.ones(E, device=device)
        sd[f"{stack}.final_layer_norm.weight"] = torch.ones(E, device=device)
    return sd


def relative_bucket(rel: Tensor, bidirectional: bool, num_buckets: int, max_distance: int) -> Tensor:
    """``T5Attention._relative_position_bucket`` (:217-262) on a small int64 host tensor (table construction, not the data path)."""
    out = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        out = out + (rel > 0).long() * num_buckets
        rel = rel.abs()
    else:
        rel = -torch.min(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).long()
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return out + torch.where(rel < max_exact, rel, large)


class _Block:
    __slots__ = ("ln_sa", "w_qkv", "w_o", "ln_ca", "w_q_ca", "w_kv_ca", "w_o_ca", "ln_ff", "w_i", "w_o_ff",
                 "w_qkv_t", "w_o_t", "w_q_ca_t", "w_kv_ca_t", "w_o_ca_t", "w_i_t", "w_o_ff_t")


class _StepDriver:
    """``eavqa_t5_decoder_step``: the calls of :meth:`FrozenT5.decode_step` issued from C++ (one ctypes call per step instead of ~340)."""

    def __init__(self, lm: "FrozenT5", cache, kv, B: int, t_max: int):
        c = lm.cfg
        self.lm, self.B, self.t_max = lm, B, t_max
        self.table = (_lib.T5DecLayer * len(lm.dec))()
        p = lambda t: t.data_ptr()
        for i, (b, (kc, vc), ckv) in enumerate(zip(lm.dec, cache, kv)):
            e = self.table[i]
            e.ln_sa, e.w_qkv, e.w_o = p(b.ln_sa), p(b.w_qkv), p(b.w_o)
            e.ln_ca, e.w_q_ca, e.w_o_ca = p(b.ln_ca), p(b.w_q_ca), p(b.w_o_ca)
            e.ln_ff, e.w_i, e.w_o_ff = p(b.ln_ff), p(b.w_i), p(b.w_o_ff)
            e.k_cache, e.v_cache, e.cross_kv = p(kc), p(vc), p(ckv)
        self.keep = (cache, kv)
        dt = ops.dtype_id(lm.dtype)
        nbytes = int(_lib.load().eavqa_t5_decoder_step_workspace_bytes(dt, B, c.d_model, c.inner, c.d_ff, int(c.gated)))
        self.ws = torch.empty(nbytes, device=lm.device, dtype=torch.uint8)
        self.out = torch.empty((B, c.d_model), device=lm.device, dtype=lm.dtype)

    def step(self, y_last: Tensor, enc_mask: Tensor, t: int, S: int, rel=None) -> Tensor:
        """``rel``: the (table, zero index) of :meth:`FrozenT5.rel_table` for ANY length >= t - one table per generation (t_max) serves
        every step: the kernel reads table[(key - query) + zero], the same values whatever the table's span."""
        lm, c = self.lm, self.lm.cfg
        rel, zero = rel if rel is not None else lm.rel_table(True, t)
        args = (ops.dtype_id(lm.dtype), len(lm.dec), self.table, lm.dec_final.data_ptr(), c.d_model, c.inner, c.n_head,
                c.d_ff, int(c.gated), _lib.ACT[c.act], float(c.eps), self.B, t, self.t_max, S, y_last.data_ptr(), self.out.data_ptr(),
                enc_mask.data_ptr() if enc_mask is not None else None, enc_mask.stride(0) if enc_mask is not None else 0, rel.data_ptr(), rel.stride(0),
                int(zero), self.ws.data_ptr(), self.ws.numel(), ops._stream())
        if lm.step_route:
            _lib.call("eavqa_t5_decoder_step_ex", *args, int(lm.step_route))          # tests / A-B measurements (include/eavqa_test.h)
        else:
            _lib.call("eavqa_t5_decoder_step", *args)
        return self.out


class FrozenT5:
    """Weights of a frozen T5 encoder-decoder packed for the HIP kernels + forward / dgrad / greedy-generation drivers."""

    def __init__(self, cfg: T5Config, state_dict: Dict[str, Tensor], dtype: torch.dtype = torch.bfloat16, device="cuda"):
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.native_step = True          # cached greedy steps through eavqa_t5_decoder_step (False: the same calls from Python, decode_step)
        self.step_route = 0              # eavqa_t5_decoder_step_ex route (0 = the library's choice; 1 = the round-3 call sequence)
        T = lambda t: t.to(device=self.device, dtype=dtype).contiguous()
        F = lambda t: t.to(device=self.device, dtype=torch.float32).contiguous()
        sd = state_dict
        self.shared = T(sd["shared.weight"])
        self.head = self.shared if cfg.tied else T(sd["lm_head.weight"])
        self.head_alpha = cfg.d_model ** -0.5 if cfg.tied else 1.0
        self.enc_rel = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].float().cpu()    # [buckets, H]
        self.dec_rel = sd["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].float().cpu()
        self._rel_cache: Dict = {}

        def pack(stack, i, dec):
            p = f"{stack}.block.{i}.layer."
            b = _Block()
            sa = p + "0.SelfAttention."
            b.ln_sa = F(sd[p + "0.layer_norm.weight"])
            b.w_qkv = T(torch.cat([sd[sa + "q.weight"], sd[sa + "k.weight"], sd[sa + "v.weight"]], 0))
            b.w_o = T(sd[sa + "o.weight"])
            b.ln_ca = b.w_q_ca = b.w_kv_ca = b.w_o_ca = None
            if dec:
                ca = p + "1.EncDecAttention."
                b.ln_ca = F(sd[p + "1.layer_norm.weight"])
                b.w_q_ca = T(sd[ca + "q.weight"])
                b.w_kv_ca = T(torch.cat([sd[ca + "k.weight"], sd[ca + "v.weight"]], 0))
                b.w_o_ca = T(sd[ca + "o.weight"])
            ff = p + ("2." if dec else "1.")
            b.ln_ff = F(sd[ff + "layer_norm.weight"])
            d = ff + "DenseReluDense."
            b.w_i = T(torch.cat([sd[d + "wi_0.weight"], sd[d + "wi_1.weight"]], 0)) if cfg.gated else T(sd[d + "wi.weight"])
            b.w_o_ff = T(sd[d + "wo.weight"])
            b.w_qkv_t = b.w_o_t = b.w_q_ca_t = b.w_kv_ca_t = b.w_o_ca_t = b.w_i_t = b.w_o_ff_t = None
            return b

        self.enc = [pack("encoder", i, False) for i in range(cfg.n_layer)]
        self.dec = [pack("decoder", i, True) for i in range(cfg.n_dec_layer)]
        self.enc_final, self.dec_final = F(sd["encoder.final_layer_norm.weight"]), F(sd["decoder.final_layer_norm.weight"])
        self.head_t = None
        self._bwd_ready = False

    @property
    def model_dim(self) -> int:                  # ``self.lm.model_dim`` vct0.py:313
        return self.cfg.d_model

    @property
    def vpad(self) -> int:
        return (self.cfg.vocab + 63) // 64 * 64

    def _prepare_backward(self) -> None:
        if self._bwd_ready:
            return
        tr = lambda w: None if w is None else w.T.contiguous()
        for b in self.enc + self.dec:
            b.w_qkv_t, b.w_o_t, b.w_i_t, b.w_o_ff_t = tr(b.w_qkv), tr(b.w_o), tr(b.w_i), tr(b.w_o_ff)
            b.w_q_ca_t, b.w_kv_ca_t, b.w_o_ca_t = tr(b.w_q_ca), tr(b.w_kv_ca), tr(b.w_o_ca)
        V, E = self.head.shape
        self.head_t = torch.zeros((E, self.vpad), device=self.device, dtype=self.dtype)
        self.head_t[:, :V] = self.head.T
        self._bwd_ready = True

    # ---------------------------------------------------------------- relative-position bias tables
    def rel_table(self, decoder: bool, S: int):
        """(table [H, 2 S - 1] float32 on the device, index of offset 0): values[h][key - query + S - 1] = rel[bucket(key - query)][h]
        (``compute_bias`` :264-279).  Built once per (stack, S) on the host from the 32 x H embedding - a few KiB."""
        key = (decoder, S)
        if key not in self._rel_cache:
            c = self.cfg
            off = torch.arange(-(S - 1), S, dtype=torch.long)
            b = relative_bucket(off, not decoder, c.num_buckets, c.max_distance)
            tab = (self.dec_rel if decoder else self.enc_rel)[b].T.contiguous()            # [H, 2S-1]
            self._rel_cache[key] = (tab.to(self.device), S - 1)
        return self._rel_cache[key]

    # ---------------------------------------------------------------- forward
    def _ffn(self, b: _Block, a2: Tensor, x: Tensor, save: bool):
        c = self.cfg
        if c.gated:
            u = ops.gemm(a2, b.w_i)                                  # [M, 2F] = [wi_0 x | wi_1 x]
            h = ops.gated_act_fwd(u, c.act)
        else:
            u = torch.empty((a2.shape[0], c.d_ff), device=self.device, dtype=self.dtype) if save else None
            h = ops.gemm(a2, b.w_i, act="relu", aux_out=u)
        return ops.gemm(h, b.w_o_ff, residual=x, out_f32=True), u

    def encode(self, x: Tensor, mask: Tensor, B: int, S: int, save: bool = False):
        """``x``: float32 [B*S, E] input embeddings (rows (b, s)); ``mask`` int32 [B, S].  Returns (encoder output in the compute dtype,
        tape)."""
        c, T = self.cfg, self.dtype
        I, H, dkv = c.inner, c.n_head, c.d_kv
        rel, zero = self.rel_table(False, S)
        tape = [] if save else None
        for b in self.enc:
            a, r1 = ops.rmsnorm_fwd(x, b.ln_sa, c.eps, T, save_stats=True)
            qkv = ops.gemm(a, b.w_qkv)
            ctx, lse = ops.attention_fwd_rel(qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:], B, H, S, S, dkv, rel_bias=rel, rel_zero=zero,
                                             key_mask=mask, causal=False, scale=1.0, save_lse=True)
            x1 = ops.gemm(ctx, b.w_o, residual=x, out_f32=True)
            a2, r2 = ops.rmsnorm_fwd(x1, b.ln_ff, c.eps, T, save_stats=True)
            x2, u = self._ffn(b, a2, x1, save)
            if save:
                tape.append((x, r1, qkv, ctx, lse, x1, r2, u))
            x = x2
        out, rf = ops.rmsnorm_fwd(x, self.enc_final, c.eps, T, save_stats=True)
        return out, (dict(layers=tape, x_last=x, rf=rf, mask=mask, B=B, S=S) if save else None)

    def cross_kv(self, enc_out: Tensor) -> List[Tensor]:
        """K | V of every decoder layer's cross-attention over the encoder output (computed once per generation)."""
        return [ops.gemm(enc_out, b.w_kv_ca) for b in self.dec]

    def decode(self, y: Tensor, enc_out: Tensor, enc_mask: Tensor, B: int, Td: int, S: int, save: bool = False, kv: Optional[List[Tensor]] = None,
               dec_mask: Optional[Tensor] = None):
        """Teacher-forced / re-forward decoder over ``Td`` positions: ``y`` float32 [B*Td, E] decoder input embeddings.  ``dec_mask`` (int32
        [B, Td], generation with a padded decoder prompt only): 0 = a key the self-attention must not see (HF's ``decoder_attention_mask``;
        positions stay absolute, as in HF).  Returns (final hidden rows in the compute dtype [B*Td, E], tape)."""
        c, T = self.cfg, self.dtype
        I, H, dkv = c.inner, c.n_head, c.d_kv
        rel, zero = self.rel_table(True, Td)
        tape = [] if save else None
        x = y
        for li, b in enumerate(self.dec):
            a, r1 = ops.rmsnorm_fwd(x, b.ln_sa, c.eps, T, save_stats=True)
            qkv = ops.gemm(a, b.w_qkv)
            ctx, lse = ops.attention_fwd_rel(qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:], B, H, Td, Td, dkv, rel_bias=rel, rel_zero=zero,
                                             key_mask=dec_mask, causal=True, scale=1.0, save_lse=True)
            x1 = ops.gemm(ctx, b.w_o, residual=x, out_f32=True)
            ac, rc = ops.rmsnorm_fwd(x1, b.ln_ca, c.eps, T, save_stats=True)
            qc = ops.gemm(ac, b.w_q_ca)
            kvc = kv[li] if kv is not None else ops.gemm(enc_out, b.w_kv_ca)
            cctx, clse = ops.attention_fwd_rel(qc, kvc[:, :I], kvc[:, I:], B, H, Td, S, dkv, rel_bias=None, key_mask=enc_mask, causal=False,
                                               scale=1.0, save_lse=True)
            x2 = ops.gemm(cctx, b.w_o_ca, residual=x1, out_f32=True)
            a3, r3 = ops.rmsnorm_fwd(x2, b.ln_ff, c.eps, T, save_stats=True)
            x3, u = self._ffn(b, a3, x2, save)
            if save:
                tape.append((x, r1, qkv, ctx, lse, x1, rc, qc, kvc, cctx, clse, x2, r3, u))
            x = x3
        out, rf = ops.rmsnorm_fwd(x, self.dec_final, c.eps, T, save_stats=True)
        return out, (dict(layers=tape, x_last=x, rf=rf, enc_mask=enc_mask, B=B, Td=Td, S=S) if save else None)

    def decode_step(self, y_last: Tensor, cache: List, enc_mask: Tensor, B: int, t: int, S: int, kv: List[Tensor], t_max: int, rel=None) -> Tensor:
        """One cached decoder step: ``y_last`` float32 [B, E] is the input embedding of decoder position t - 1; the self-attention K / V of
        positions 0 .. t - 2 are in ``cache`` (per layer two [B * t_max, inner] row views), position t - 1 is appended here.  Same
        arithmetic as the last row of :meth:`decode` over t positions (one query at the END of t keys: causal offset and relative-position
        bias as there), on B rows instead of B * t - which keeps every GEMM on the M <= 64 weight-streaming kernels.  Returns [B, E]."""
        c, T = self.cfg, self.dtype
        I, H, dkv = c.inner, c.n_head, c.d_kv
        rel, zero = rel if rel is not None else self.rel_table(True, t)     # (one table of span t_max per generation: see _StepDriver.step)
        if self.splitk_step_plan(B, t, S) is not None and self.step_route != 1:
            return self._decode_step_splitk(y_last, cache, enc_mask, B, t, S, kv, t_max, rel, zero)
        x = y_last
        for li, b in enumerate(self.dec):
            a = ops.rmsnorm_fwd(x, b.ln_sa, c.eps, T)
            qkv = ops.gemm(a, b.w_qkv)
            kc, vc = cache[li]
            ops.copy_rows(qkv[:, I:2 * I], kc, B, 1, I, 1, t_max, t - 1)
            ops.copy_rows(qkv[:, 2 * I:], vc, B, 1, I, 1, t_max, t - 1)
            ctx = ops.attention_fwd_rel(qkv[:, :I], kc, vc, B, H, 1, t, dkv, rel_bias=rel, rel_zero=zero, causal=True, scale=1.0,
                                        q_batch_rows=1, kv_batch_rows=t_max)
            x1 = ops.gemm(ctx, b.w_o, residual=x, out_f32=True)
            ac = ops.rmsnorm_fwd(x1, b.ln_ca, c.eps, T)
            qc = ops.gemm(ac, b.w_q_ca)
            kvc = kv[li]
            cctx = ops.attention_fwd_rel(qc, kvc[:, :I], kvc[:, I:], B, H, 1, S, dkv, rel_bias=None, key_mask=enc_mask, causal=False, scale=1.0)
            x2 = ops.gemm(cctx, b.w_o_ca, residual=x1, out_f32=True)
            a3 = ops.rmsnorm_fwd(x2, b.ln_ff, c.eps, T)
            x, _ = self._ffn(b, a3, x2, False)
        return ops.rmsnorm_fwd(x, self.dec_final, c.eps, T)

    def splitk_step_plan(self, B: int, t: int, S: int):
        """The split counts of the six projections when a cached step can take the split-K route of ``eavqa_t5_decoder_step`` (bf16, B <= 64,
        head dim a multiple of 8, every shape plannable) - the conditions of csrc/t5_block.cpp ``plan_t5`` -, else None."""
        c = self.cfg
        E, I, F = c.d_model, c.inner, c.d_ff
        if self.dtype != torch.bfloat16 or B > 64 or E % 8 or I % 8 or F % 4 or c.d_kv % 8 or c.d_kv > 128 or t > 3584 or S > 3584:
            return None
        plan = _lib.load().eavqa_gemm_splitk_plan
        ks = dict(qkv=plan(B, 3 * I, E), o=plan(B, E, I), qc=plan(B, I, E), wi=plan(B, (2 if c.gated else 1) * F, E), wo=plan(B, E, F))
        return ks if all(v > 0 for v in ks.values()) else None

    def _decode_step_splitk(self, y_last, cache, enc_mask, B, t, S, kv, t_max, rel, zero) -> Tensor:
        """The split-K route of ``eavqa_t5_decoder_step`` call for call (csrc/t5_block.cpp): every projection leaves fp32 partial sums that its
        consumer adds up - the RMSNorm pass, the decode attention (which also appends K / V), the gated finish."""
        c, T = self.cfg, self.dtype
        I, H, dkv = c.inner, c.n_head, c.d_kv
        ks = self.splitk_step_plan(B, t, S)
        x, x2, part = y_last, None, None
        for li, b in enumerate(self.dec):
            if li == 0:
                a = ops.rmsnorm_splitk(x, b.ln_sa, c.eps, T)
            else:
                x = torch.empty_like(y_last)
                a = ops.rmsnorm_splitk(x2, b.ln_sa, c.eps, T, part=part, x_out=x)
            kc, vc = cache[li]
            part = ops.gemm_splitk(a, b.w_qkv, ks=ks["qkv"])
            ctx = ops.attention_decode_splitk_rel(part, kc, vc, B, H, t, dkv, kv_batch_rows=t_max, scale=1.0, rel_bias=rel, rel_zero=zero)
            part = ops.gemm_splitk(ctx, b.w_o, ks=ks["o"])
            x1 = torch.empty_like(y_last)
            a = ops.rmsnorm_splitk(x, b.ln_ca, c.eps, T, part=part, x_out=x1)
            part = ops.gemm_splitk(a, b.w_q_ca, ks=ks["qc"])
            kvc = kv[li]
            ctx = ops.attention_decode_splitk_rel(part, kvc[:, :I], kvc[:, I:], B, H, S, dkv, kv_batch_rows=S, key_mask=enc_mask, scale=1.0)
            part = ops.gemm_splitk(ctx, b.w_o_ca, ks=ks["o"])
            x2 = torch.empty_like(y_last)
            a = ops.rmsnorm_splitk(x1, b.ln_ff, c.eps, T, part=part, x_out=x2)
            part = ops.gemm_splitk(a, b.w_i, ks=ks["wi"])
            if c.gated:
                h = ops.splitk_finish_gated(part, c.act, T)
            else:
                h = torch.empty((B, c.d_ff), device=self.device, dtype=T)
                ops.splitk_finish(part, [h], act=c.act)
            part = ops.gemm_splitk(h, b.w_o_ff, ks=ks["wo"])
        return ops.rmsnorm_splitk(x2, self.dec_final, c.eps, T, part=part)

    def logits(self, hidden: Tensor) -> Tensor:
        lg = torch.empty((hidden.shape[0], self.vpad), device=self.device, dtype=torch.float32)
        ops.gemm(hidden, self.head, out=lg[:, :self.cfg.vocab], alpha=self.head_alpha)
        return lg

    def shift_right(self, labels: Tensor) -> Tensor:
        """``_shift_right`` :618-637 (int64 index bookkeeping on a [B, T] tensor)."""
        c = self.cfg
        out = torch.full_like(labels, c.decoder_start_token_id)
        out[:, 1:] = labels[:, :-1]
        return out.masked_fill(out == -100, c.pad_token_id)

    def embed(self, ids: Tensor, prefix_rows: Optional[Tensor] = None) -> Tensor:
        """float32 rows: ``shared[id]`` for id >= 0, ``prefix_rows[-id - 1]`` otherwise (the sentinel expansion)."""
        return ops.embed_assemble(ids.reshape(-1).to(torch.int32).contiguous(), None, self.shared, prefix_rows, None)

    def forward_train(self, enc_rows: Tensor, B: int, S: int, labels: Tensor):
        """``lm(inputs_embeds=enc_rows, labels=labels)`` (vct0.py:390-393): encoder over the given rows (all attended), teacher-forced
        decoder, CE over labels != -100.  ``enc_rows``: compute-dtype [B*S, E] (the mapper's output).  Returns a dict with loss,
        logits [B*T, vpad] float32 and the tape."""
        Td = labels.shape[1]
        mask = torch.ones((B, S), device=self.device, dtype=torch.int32)
        src = -(torch.arange(B * S, device=self.device, dtype=torch.int32) + 1)                  # every encoder row is a prefix row
        x = ops.embed_assemble(src, None, self.shared, enc_rows, None)
        enc_out, etape = self.encode(x, mask, B, S, save=True)
        y = self.embed(self.shift_right(labels))
        hid, dtape = self.decode(y, enc_out, mask, B, Td, S, save=True)
        lg = self.logits(hid)
        flat = labels.reshape(-1).contiguous()
        loss, count, row_lse = ops.ce_fwd(lg, flat, self.cfg.vocab)
        return dict(loss=loss, logits=lg, tape=dict(enc=etape, dec=dtape, enc_out=enc_out, hid=hid, logits=lg, labels=flat, row_lse=row_lse, count=count,
                                                    src=src, n_rows=enc_rows.shape[0]))

    # ---------------------------------------------------------------- backward (dgrad only)
    def _ffn_bwd(self, b: _Block, dxT: Tensor, u: Tensor) -> Tensor:
        c = self.cfg
        if c.gated:
            dh = ops.gemm(dxT, b.w_o_ff_t)                           # [M, F]
            du = ops.gated_act_bwd(u, dh, c.act)                     # [M, 2F]
        else:
            du = ops.gemm(dxT, b.w_o_ff_t, act="relu", aux_in=u)
        return ops.gemm(du, b.w_i_t)                                 # [M, E]

    def backward(self, tape: dict, gloss: Tensor) -> Tensor:
        """d loss / d enc_rows (compute dtype, [n_rows, E])."""
        self._prepare_backward()
        c, T = self.cfg, self.dtype
        I, H, dkv = c.inner, c.n_head, c.d_kv
        lowp = T != torch.float32
        dt, et = tape["dec"], tape["enc"]
        B, Td, S = dt["B"], dt["Td"], dt["S"]
        Md, Me = B * Td, B * S
        new_lowp = lambda M: torch.empty((M, c.d_model), device=self.device, dtype=T) if lowp else None
        dlog = ops.ce_bwd(tape["logits"], tape["labels"], c.vocab, tape["row_lse"], tape["count"], gloss, T, self.vpad)
        dhid = ops.gemm(dlog, self.head_t, alpha=self.head_alpha)
        dxT = new_lowp(Md)
        dx = ops.rmsnorm_bwd(dt["x_last"], dhid, self.dec_final, dt["rf"], lowp_out=dxT)
        d_enc = torch.zeros((Me, c.d_model), device=self.device, dtype=torch.float32)       # cross-attention gradients of every layer
        rel, zero = self.rel_table(True, Td)
        for b, (x, r1, qkv, ctx, lse, x1, rc, qc, kvc, cctx, clse, x2, r3, u) in zip(reversed(self.dec), reversed(dt["layers"])):
            da3 = self._ffn_bwd(b, dxT if lowp else dx, u)
            dx2 = ops.rmsnorm_bwd(x2, da3, b.ln_ff, r3, dres=dx, out=dx, lowp_out=dxT)
            dcctx = ops.gemm(dxT if lowp else dx2, b.w_o_ca_t)
            dkvc = torch.empty_like(kvc)                                  # [Me, 2 I]: dK | dV written in place by the kernel
            dqc, _, _ = ops.attention_bwd_rel(qc, kvc[:, :I], kvc[:, I:], cctx, dcctx, clse, B, H, Td, S, dkv, rel_bias=None,
                                              key_mask=dt["enc_mask"], causal=False, scale=1.0, dk=dkvc[:, :I], dv=dkvc[:, I:])
            ops.gemm(dkvc, b.w_kv_ca_t, residual=d_enc, out=d_enc)
            dac = ops.gemm(dqc, b.w_q_ca_t)
            dx1 = ops.rmsnorm_bwd(x1, dac, b.ln_ca, rc, dres=dx2, out=dx2, lowp_out=dxT)
            dctx = ops.gemm(dxT if lowp else dx1, b.w_o_t)
            dqkv = torch.empty_like(qkv)
            ops.attention_bwd_rel(qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:], ctx, dctx, lse, B, H, Td, Td, dkv, rel_bias=rel, rel_zero=zero,
                                  causal=True, scale=1.0, dq=dqkv[:, :I], dk=dqkv[:, I:2 * I], dv=dqkv[:, 2 * I:])
            da = ops.gemm(dqkv, b.w_qkv_t)
            dx = ops.rmsnorm_bwd(x, da, b.ln_sa, r1, dres=dx1, out=dx1, lowp_out=dxT)
        # encoder
        rel, zero = self.rel_table(False, S)
        d_out = ops.cast_rows(d_enc, T) if lowp else d_enc
        exT = new_lowp(Me)
        dx = ops.rmsnorm_bwd(et["x_last"], d_out, self.enc_final, et["rf"], lowp_out=exT)
        for b, (x, r1, qkv, ctx, lse, x1, r2, u) in zip(reversed(self.enc), reversed(et["layers"])):
            da2 = self._ffn_bwd(b, exT if lowp else dx, u)
            dx1 = ops.rmsnorm_bwd(x1, da2, b.ln_ff, r2, dres=dx, out=dx, lowp_out=exT)
            dctx = ops.gemm(exT if lowp else dx1, b.w_o_t)
            dqkv = torch.empty_like(qkv)
            ops.attention_bwd_rel(qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:], ctx, dctx, lse, B, H, S, S, dkv, rel_bias=rel, rel_zero=zero,
                                  key_mask=et["mask"], causal=False, scale=1.0, dq=dqkv[:, :I], dk=dqkv[:, I:2 * I], dv=dqkv[:, 2 * I:])
            da = ops.gemm(dqkv, b.w_qkv_t)
            dx = ops.rmsnorm_bwd(x, da, b.ln_sa, r1, dres=dx1, out=dx1, lowp_out=exT)
        return ops.embed_assemble_bwd(tape["src"], dx, tape["n_rows"], T)

    # ---------------------------------------------------------------- greedy generation
    @torch.no_grad()
    def greedy(self, enc_out: Tensor, enc_mask: Tensor, B: int, S: int, max_length: int, dec_prompt: Optional[Tensor] = None,
               output_scores: bool = False, use_cache: bool = True, dec_mask: Optional[Tensor] = None):
        """HF greedy search for an encoder-decoder: start = decoder_start_token_id, a row that produced eos emits pad afterwards, stop
        when every row is finished or ``max_length`` decoder positions exist.  The cross-attention K / V of every layer are computed
        once; a step runs the decoder on the newest position against a self-attention K / V cache (``use_cache``; with a multi-token
        decoder prompt, or ``use_cache=False``, every step re-runs the decoder over its own short prefix - the same logits).  Returns
        ``(sequences int64 [B, <= max_length] on the host, [per-step logits float32 [B, V] on the host] | None)``."""
        c = self.cfg
        kv = self.cross_kv(enc_out)
        start = torch.full((B, 1), c.decoder_start_token_id, dtype=torch.int64, device=self.device)
        dkey = None
        if dec_prompt is not None:
            # HF ``_prepare_decoder_input_ids_for_generation``: a prompt of which NO row begins with the decoder start id gets it prepended
            # (and its mask a column of ones); a left-padded prompt (pad id == start id for T5, module_parser.py:397-399) is taken as is
            prompt = dec_prompt.to(self.device)
            pm = dec_mask.to(self.device) if dec_mask is not None else torch.ones_like(prompt)
            if bool((prompt[:, 0] != c.decoder_start_token_id).all()):
                prompt, pm = torch.cat([start, prompt], dim=1), torch.cat([torch.ones_like(pm[:, :1]), pm], dim=1)
            start = prompt
            if not bool((pm != 0).all()):
                dkey = torch.ones((B, max(max_length, start.shape[1])), dtype=torch.int32, device=self.device)
                dkey[:, :start.shape[1]] = (pm != 0).to(torch.int32)
        P = start.shape[1]
        seq = torch.full((B, max(max_length, P)), c.pad_token_id, dtype=torch.int64, device=self.device)
        seq[:, :P] = start
        raw = torch.empty(B, dtype=torch.int32, device=self.device)
        unfinished = torch.ones(B, dtype=torch.int32, device=self.device)
        scores = [] if output_scores else None
        # `alive[t]` = some row still unfinished after position t was written.  The host looks at it every fourth step only (one
        # device -> host round trip per step would keep the launch queue empty); positions written after every row had finished hold
        # pad and are cut off below, so the result is the one of a check after every step.
        alive = torch.zeros(max(max_length, P) + 1, dtype=torch.int32, device=self.device)     # written by the pick kernel (eavqa_greedy_pick)
        cached = use_cache and P == 1 and max_length > 1
        if cached:
            t_max = max_length
            cache = [(torch.empty((B * t_max, c.inner), device=self.device, dtype=self.dtype),
                      torch.empty((B * t_max, c.inner), device=self.device, dtype=self.dtype)) for _ in self.dec]
            driver = _StepDriver(self, cache, kv, B, t_max) if self.native_step else None
            # ONE decoder bias table per generation (span t_max): a table per step length meant max_length - 1 host-built tables and a
            # blocking pageable host -> device copy per step of the first generation
            rel_gen = self.rel_table(True, t_max)
        t = P
        while t < max_length:
            if cached and driver is not None:
                last = driver.step(self.embed(seq[:, t - 1].contiguous()), enc_mask, t, S, rel_gen)
            elif cached:
                last = self.decode_step(self.embed(seq[:, t - 1].contiguous()), cache, enc_mask, B, t, S, kv, t_max, rel_gen)
            else:
                y = self.embed(seq[:, :t].contiguous())
                hid, _ = self.decode(y, enc_out, enc_mask, B, t, S, kv=kv, dec_mask=dkey[:, :t].contiguous() if dkey is not None else None)
                last = hid.view(B, t, c.d_model)[:, -1].contiguous()
            lg = self.logits(last)
            if output_scores:
                scores.append(lg[:, :c.vocab].float())
            ops.greedy_pick(lg, c.vocab, c.pad_token_id, c.eos_token_id, raw, seq[:, t], unfinished,      # emitted token = what is fed back
                            any_unfinished=alive[t:t + 1])
            t += 1
            if (t - P) % 4 == 0 and int(alive[t - 1].item()) == 0:
                break
        dead = (alive[P:t] == 0).nonzero()
        if dead.numel():
            t = P + int(dead[0].item()) + 1
        if scores is not None:
            scores = [x.cpu() for x in scores[:t - P]]
        return seq[:, :t].cpu(), scores
