"""Frozen causal LM (GPT-2 / OPT) executed entirely by the HIP kernels.

Arithmetic mirrored:
  * GPT-2: HF ``GPT2LMHeadModel`` transformers/models/gpt2/modeling_gpt2.py:571-620 (positions =
    arange, wpe add), :246-309 (pre-LN block), :75-226 (c_attn Conv1D ``[in,out]``, scale hd^-0.5,
    causal + key-padding mask), :229-243 (gelu_new MLP), :698 (lm_head tied to wte);
  * OPT: HF ``OPTForCausalLM`` transformers/models/opt/modeling_opt.py:45-70 (learned positions,
    offset 2, ``cumsum(mask)*mask-1``), :97-181 (q scaled before q.k), :184-254 (pre-LN, ReLU),
    :273-397 (final_layer_norm), :443-538 (lm_head);
  * loss: transformers/loss/loss_utils.py:32-71 (shift, ignore -100, mean).

The LM is frozen (``ClipCaptionPrefix.train`` clipcap.py:594-599), so the backward pass is dgrad
only: no weight gradient and no GEMM input needs saving - only what LayerNorm, attention and the
activation need.  Weights are pre-packed once at load into the k-contiguous layouts the MFMA GEMM
streams fastest: a forward copy ``[N,K]`` and (lazily, for training) a backward copy ``[K,N]``;
288 GB of HBM makes the second copy free.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .. import ops

Tensor = torch.Tensor


@dataclass
class LMConfig:
    arch: str            # "gpt2" | "opt"
    n_layer: int
    n_head: int
    n_embd: int
    ffn: int
    vocab: int
    n_pos: int           # number of learned positions (OPT: excluding the offset of 2)
    eps: float = 1e-5
    act: str = "gelu_new"
    eos_token_id: Optional[int] = None
    pad_token_id: Optional[int] = None

    @property
    def head_dim(self) -> int:
        return self.n_embd // self.n_head

    @property
    def pos_mode(self) -> int:
        return 0 if self.arch == "gpt2" else 1

    @staticmethod
    def from_hf_dict(d: dict) -> "LMConfig":
        mt = d.get("model_type", "gpt2")
        if mt == "gpt2":
            E = d.get("n_embd", 768)
            return LMConfig("gpt2", d.get("n_layer", 12), d.get("n_head", 12), E, d.get("n_inner") or 4 * E,
                            d.get("vocab_size", 50257), d.get("n_positions", 1024), d.get("layer_norm_epsilon", 1e-5),
                            d.get("activation_function", "gelu_new"), d.get("eos_token_id", 50256), d.get("pad_token_id"))
        if mt == "opt":
            E = d.get("hidden_size", 768)
            if d.get("word_embed_proj_dim", E) != E or not d.get("do_layer_norm_before", True):
                raise NotImplementedError("OPT-350m style project_in/out / post-LN is not on the hot path")
            return LMConfig("opt", d.get("num_hidden_layers", 12), d.get("num_attention_heads", 12), E, d.get("ffn_dim", 3072),
                            d.get("vocab_size", 50272), d.get("max_position_embeddings", 2048), 1e-5,
                            d.get("activation_function", "relu"), d.get("eos_token_id", 2), d.get("pad_token_id", 1))
        raise NotImplementedError(f"model_type {mt!r} is not a causal LM on the hot path")


# named model shapes used by BASELINE.json configs (public architecture constants)
KNOWN_CONFIGS = {
    "gpt2": dict(model_type="gpt2", n_embd=768, n_layer=12, n_head=12, vocab_size=50257, n_positions=1024),
    "gpt2-medium": dict(model_type="gpt2", n_embd=1024, n_layer=24, n_head=16, vocab_size=50257, n_positions=1024),
    "gpt2-large": dict(model_type="gpt2", n_embd=1280, n_layer=36, n_head=20, vocab_size=50257, n_positions=1024),
    "gpt2-xl": dict(model_type="gpt2", n_embd=1600, n_layer=48, n_head=25, vocab_size=50257, n_positions=1024),
    "facebook/opt-125m": dict(model_type="opt", hidden_size=768, num_hidden_layers=12, num_attention_heads=12, ffn_dim=3072),
    "facebook/opt-1.3b": dict(model_type="opt", hidden_size=2048, num_hidden_layers=24, num_attention_heads=32, ffn_dim=8192),
    "facebook/opt-2.7b": dict(model_type="opt", hidden_size=2560, num_hidden_layers=32, num_attention_heads=32, ffn_dim=10240),
    "facebook/opt-6.7b": dict(model_type="opt", hidden_size=4096, num_hidden_layers=32, num_attention_heads=32, ffn_dim=16384),
}


def random_init_state_dict(cfg: LMConfig, seed: int = 2021, device="cpu") -> Dict[str, Tensor]:
    """Seeded random-init weights in HF key names / HF shapes (normal(0, 0.02), LayerNorm 1/0,
    zero biases; GPT-2 c_proj scaled by 1/sqrt(2*n_layer) as HF ``_init_weights`` does).  Used for
    synthetic benchmarks - there are no pretrained weights offline."""
    g = torch.Generator(device=device).manual_seed(seed)
    E, F, V = cfg.n_embd, cfg.ffn, cfg.vocab

    def n(*shape, std=0.02):
        return torch.randn(*shape, generator=g, device=device) * std

    sd: Dict[str, Tensor] = {}
    ones, zeros = (lambda k: torch.ones(k, device=device)), (lambda k: torch.zeros(k, device=device))
    if cfg.arch == "gpt2":
        sd["transformer.wte.weight"] = n(V, E)
        sd["transformer.wpe.weight"] = n(cfg.n_pos, E)
        ps = 0.02 / math.sqrt(2 * cfg.n_layer)
        for i in range(cfg.n_layer):
            p = f"transformer.h.{i}."
            sd[p + "ln_1.weight"], sd[p + "ln_1.bias"] = ones(E), zeros(E)
            sd[p + "attn.c_attn.weight"], sd[p + "attn.c_attn.bias"] = n(E, 3 * E), zeros(3 * E)
            sd[p + "attn.c_proj.weight"], sd[p + "attn.c_proj.bias"] = n(E, E, std=ps), zeros(E)
            sd[p + "ln_2.weight"], sd[p + "ln_2.bias"] = ones(E), zeros(E)
            sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"] = n(E, F), zeros(F)
            sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"] = n(F, E, std=ps), zeros(E)
        sd["transformer.ln_f.weight"], sd["transformer.ln_f.bias"] = ones(E), zeros(E)
    else:
        pre = "model.decoder."
        sd[pre + "embed_tokens.weight"] = n(V, E)
        sd[pre + "embed_positions.weight"] = n(cfg.n_pos + 2, E)
        for i in range(cfg.n_layer):
            p = f"{pre}layers.{i}."
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                sd[p + f"self_attn.{nm}.weight"], sd[p + f"self_attn.{nm}.bias"] = n(E, E), zeros(E)
            sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"] = ones(E), zeros(E)
            sd[p + "fc1.weight"], sd[p + "fc1.bias"] = n(F, E), zeros(F)
            sd[p + "fc2.weight"], sd[p + "fc2.bias"] = n(E, F), zeros(E)
            sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"] = ones(E), zeros(E)
        sd[pre + "final_layer_norm.weight"], sd[pre + "final_layer_norm.bias"] = ones(E), zeros(E)
    return sd


def synthetic_weights_notice(model_version: str) -> None:
    """A known architecture NAME without a local HF directory gets seeded random-init weights (there is no network): say so once per
    name, loudly - answers and losses from such a model mean nothing - and refuse outright under ``EAVQA_REQUIRE_PRETRAINED=1``."""
    if os.environ.get("EAVQA_REQUIRE_PRETRAINED", "0") == "1":
        raise FileNotFoundError(f"{model_version!r}: no local HF directory (EAVQA_MODEL_DIR) and EAVQA_REQUIRE_PRETRAINED=1 forbids "
                                "the seeded random-init stand-in")
    import warnings
    warnings.warn(f"{model_version!r} is not a local HF directory: using SEEDED RANDOM-INIT weights of that architecture (synthetic runs "
                  "only; set EAVQA_MODEL_DIR to a directory of HF checkpoints, or EAVQA_REQUIRE_PRETRAINED=1 to make this an error)",
                  RuntimeWarning, stacklevel=3)


def load_local_hf(path: str):
    """(config dict, state dict) from a local HF directory: safetensors, else a torch file loaded
    with ``weights_only=True``.  Nothing is fetched from the network."""
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return cfg, load_file(st)
    pt = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(pt):
        return cfg, torch.load(pt, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")


class Fp8Weight:
    """A frozen Linear weight ``[N, K]`` held as OCP e4m3 bytes with ONE float scale (``w ~ scale * e4m3``): BASELINE
    configs[4].  ``t()`` is the transposed copy for the dgrad GEMM - the same bytes transposed, the same scale."""

    __slots__ = ("q", "scale", "shape")

    def __init__(self, q: Tensor, scale: float):
        self.q, self.scale, self.shape = q, float(scale), tuple(q.shape)

    @staticmethod
    def quantize(w: Tensor, device) -> "Fp8Weight":
        w = w.to(device=device, dtype=torch.float32)
        amax = float(w.abs().max().item())
        scale = amax / 448.0 if amax > 0 else 1.0
        q = (w / scale).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()      # load-time cast (torch), not on the hot path
        return Fp8Weight(q, scale)

    def t(self) -> "Fp8Weight":
        return Fp8Weight(self.q.T.contiguous(), self.scale)

    def dequantize(self) -> Tensor:
        return self.q.view(torch.float8_e4m3fn).float() * self.scale


def linear(a: Tensor, w, **kw) -> Tensor:
    """``epilogue(a @ w^T)``: the bf16 / fp32 MFMA GEMM, or - for an :class:`Fp8Weight` - row-wise e4m3 quantisation of ``a``
    followed by the block-scaled fp8 GEMM (eavqa_quantize_rows_fp8 + eavqa_gemm_fp8)."""
    if isinstance(w, Fp8Weight):
        # (a producer that already quantised its rows - eavqa_layernorm_fwd_fp8 / _bwd_fp8 - hands over the pair)
        aq, a_scale = a if isinstance(a, tuple) else ops.quantize_rows_fp8(a)
        return ops.gemm_fp8(aq, a_scale, w.q, w.scale, **kw)
    return ops.gemm(a, w, **kw)


class _Layer:
    __slots__ = ("ln1_g", "ln1_b", "w_qkv", "b_qkv", "w_o", "b_o", "ln2_g", "ln2_b", "w_fc1", "b_fc1", "w_fc2", "b_fc2",
                 "w_qkv_t", "w_o_t", "w_fc1_t", "w_fc2_t",
                 "w_qkv_f", "c_qkv", "d_qkv", "w_fc1_f", "c_fc1", "d_fc1")      # LayerNorm folded into the consuming Linear (_prepare_fold)


class FrozenCausalLM:
    """Weights of a frozen GPT-2 / OPT decoder pre-packed for the HIP kernels + forward / dgrad drivers."""

    def __init__(self, cfg: LMConfig, state_dict: Dict[str, Tensor], dtype: torch.dtype = torch.bfloat16, device="cuda",
                 weight_format: str = "native"):
        """``weight_format="fp8"`` (with ``dtype=torch.bfloat16``): the Linear weights of every layer and the lm_head are held
        in e4m3 with per-tensor scales and multiplied on the block-scaled MFMA; activations stay bf16 in HBM and are quantised
        row-wise right in front of each GEMM; LayerNorm, attention, the residual stream (fp32) and the loss are unchanged."""
        self.cfg = cfg
        self.dtype = dtype
        self.device = torch.device(device)
        if weight_format not in ("native", "fp8"):
            raise ValueError("weight_format must be 'native' or 'fp8'")
        if weight_format == "fp8" and (dtype != torch.bfloat16 or cfg.n_embd % 128 or cfg.ffn % 128):
            raise ValueError("fp8 weights need bfloat16 activations and n_embd / ffn multiples of 128")
        self.weight_format = weight_format
        self._pack(state_dict)
        self._bwd_ready = False
        # eavqa_gemm_ln: ln_1 / ln_2 of layers 1.. (and ln_2 of layer 0) as an epilogue term of the QKV / FFN-up products instead of a
        # kernel of their own - bf16 weights only.  Measured equal to the LayerNorm kernels on cfg2 (what the 72 removed launches save, the
        # four longer GEMM epilogues per layer cost: profiles/round4_ln_fold.md), so it is an option (EAVQA_LN_FOLD=1), not the default.
        self.fold_layernorm = (dtype == torch.bfloat16 and weight_format == "native" and os.environ.get("EAVQA_LN_FOLD", "0") == "1")
        self._fold_ready = False
        # e4m3 weights: ln_1 / ln_2 (forward) and the LayerNorm backward hand their result to the next Linear already row-quantised
        # (EAVQA_FUSE_QUANT=0 keeps the separate eavqa_quantize_rows_fp8 launches: same bytes, the A / B switch)
        self.fuse_quantizer = os.environ.get("EAVQA_FUSE_QUANT", "1") != "0"

    # ---------------------------------------------------------------- packing
    def _T(self, t: Tensor) -> Tensor:
        return t.to(device=self.device, dtype=self.dtype).contiguous()

    def _F(self, t: Tensor) -> Tensor:
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    def _pack(self, sd: Dict[str, Tensor]) -> None:
        c = self.cfg
        self.layers = []
        if c.arch == "gpt2":
            self.wte = self._T(sd["transformer.wte.weight"])
            self.wpe = self._T(sd["transformer.wpe.weight"])
            for i in range(c.n_layer):
                p = f"transformer.h.{i}."
                L = _Layer()
                L.ln1_g, L.ln1_b = self._F(sd[p + "ln_1.weight"]), self._F(sd[p + "ln_1.bias"])
                # Conv1D stores [in,out]; the forward GEMM wants [out,in] (k contiguous)
                L.w_qkv, L.b_qkv = self._T(sd[p + "attn.c_attn.weight"].T), self._F(sd[p + "attn.c_attn.bias"])
                L.w_o, L.b_o = self._T(sd[p + "attn.c_proj.weight"].T), self._F(sd[p + "attn.c_proj.bias"])
                L.ln2_g, L.ln2_b = self._F(sd[p + "ln_2.weight"]), self._F(sd[p + "ln_2.bias"])
                L.w_fc1, L.b_fc1 = self._T(sd[p + "mlp.c_fc.weight"].T), self._F(sd[p + "mlp.c_fc.bias"])
                L.w_fc2, L.b_fc2 = self._T(sd[p + "mlp.c_proj.weight"].T), self._F(sd[p + "mlp.c_proj.bias"])
                L.w_qkv_t = L.w_o_t = L.w_fc1_t = L.w_fc2_t = None
                self.layers.append(L)
            self.lnf_g, self.lnf_b = self._F(sd["transformer.ln_f.weight"]), self._F(sd["transformer.ln_f.bias"])
            head = sd.get("lm_head.weight")
        else:
            pre = "model.decoder."
            self.wte = self._T(sd[pre + "embed_tokens.weight"])
            self.wpe = self._T(sd[pre + "embed_positions.weight"])
            for i in range(c.n_layer):
                p = f"{pre}layers.{i}."
                L = _Layer()
                L.ln1_g, L.ln1_b = self._F(sd[p + "self_attn_layer_norm.weight"]), self._F(sd[p + "self_attn_layer_norm.bias"])
                L.w_qkv = self._T(torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], dim=0))
                L.b_qkv = self._F(torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], dim=0))
                L.w_o, L.b_o = self._T(sd[p + "self_attn.out_proj.weight"]), self._F(sd[p + "self_attn.out_proj.bias"])
                L.ln2_g, L.ln2_b = self._F(sd[p + "final_layer_norm.weight"]), self._F(sd[p + "final_layer_norm.bias"])
                L.w_fc1, L.b_fc1 = self._T(sd[p + "fc1.weight"]), self._F(sd[p + "fc1.bias"])
                L.w_fc2, L.b_fc2 = self._T(sd[p + "fc2.weight"]), self._F(sd[p + "fc2.bias"])
                L.w_qkv_t = L.w_o_t = L.w_fc1_t = L.w_fc2_t = None
                self.layers.append(L)
            self.lnf_g, self.lnf_b = self._F(sd[pre + "final_layer_norm.weight"]), self._F(sd[pre + "final_layer_norm.bias"])
            head = sd.get("lm_head.weight")
        # lm_head is tied to wte in both families; an untied head is kept separately
        self.head = self.wte if head is None or head.shape == self.wte.shape and _same(head, self.wte) else self._T(head)
        self.head_t = None
        self.head_q = None
        if self.weight_format == "fp8":
            for L in self.layers:
                for name in ("w_qkv", "w_o", "w_fc1", "w_fc2"):
                    setattr(L, name, Fp8Weight.quantize(getattr(L, name), self.device))
            self.head_q = Fp8Weight.quantize(self.head, self.device)      # the bf16 wte stays for the embedding gather

    @property
    def vocab(self) -> int:
        return self.wte.shape[0]

    @property
    def vpad(self) -> int:
        q = 128 if self.weight_format == "fp8" else 64
        return (self.vocab + q - 1) // q * q   # K of the lm_head dgrad GEMM: a multiple of the kernels' K-step

    def _prepare_backward(self) -> None:
        """Transposed weight copies for the dgrad GEMMs (made once, on the first training step)."""
        if self._bwd_ready:
            return
        tr = (lambda w: w.t()) if self.weight_format == "fp8" else (lambda w: w.T.contiguous())
        for L in self.layers:
            L.w_qkv_t = tr(L.w_qkv)
            L.w_o_t = tr(L.w_o)
            L.w_fc1_t = tr(L.w_fc1)
            L.w_fc2_t = tr(L.w_fc2)
        V, E = self.head.shape
        if self.weight_format == "fp8":
            qt = torch.zeros((E, self.vpad), device=self.device, dtype=torch.uint8)      # e4m3 zero = byte 0
            qt[:, :V] = self.head_q.q.T
            self.head_t = Fp8Weight(qt, self.head_q.scale)
        else:
            self.head_t = torch.zeros((E, self.vpad), device=self.device, dtype=self.dtype)
            self.head_t[:, :V] = self.head.T
        self._bwd_ready = True

    def _prepare_fold(self) -> None:
        """``LayerNorm(x) W^T + b = rstd (x W'^T - mean c) + d`` with ``W' = W * gamma``, ``c = W' 1``, ``d = W beta + b`` (include/eavqa.h,
        eavqa_gemm_ln): the folded copies of the two weights of every layer that follow a LayerNorm, made once on the first training /
        scoring forward.  ``c`` sums W' AS STORED (bf16), so the rank-1 term cancels the mean of exactly the product the kernel forms."""
        if self._fold_ready:
            return
        for L in self.layers:
            for w, b, g, be, nm in ((L.w_qkv, L.b_qkv, L.ln1_g, L.ln1_b, "qkv"), (L.w_fc1, L.b_fc1, L.ln2_g, L.ln2_b, "fc1")):
                w32 = w.float()
                wf = (w32 * g[None, :]).to(self.dtype).contiguous()
                setattr(L, f"w_{nm}_f", wf)
                setattr(L, f"c_{nm}", wf.float().sum(1).contiguous())
                setattr(L, f"d_{nm}", (w32 @ be + b).contiguous())
                del w32
        self._fold_ready = True

    def resize_token_embeddings(self, n: int) -> None:
        """``model.gpt.resize_token_embeddings(len(tokenizer))`` src/trainers/clipcap_exector.py:56.
        New rows get the mean of the existing embeddings (HF draws them around that mean)."""
        V, E = self.wte.shape
        if n == V:
            return
        tied = self.head is self.wte
        new = torch.empty((n, E), device=self.device, dtype=self.dtype)
        keep = min(n, V)
        new[:keep] = self.wte[:keep]
        if n > V:
            new[V:] = self.wte.float().mean(0).to(self.dtype)
        self.wte = new
        if tied:
            self.head = new
        else:
            h = torch.empty((n, E), device=self.device, dtype=self.dtype)
            h[:keep] = self.head[:keep]
            if n > V:
                h[V:] = self.head.float().mean(0).to(self.dtype)
            self.head = h
        self.cfg.vocab = n
        self._bwd_ready = False
        if self.weight_format == "fp8":
            self.head_q = Fp8Weight.quantize(self.head, self.device)

    def load_token_embeddings(self, weight: Tensor) -> None:
        """Replace the token embedding matrix (and the tied head) by ``weight`` [vocab, E] - e.g. the rows a reference
        checkpoint holds after its ``resize_token_embeddings`` drew the new ones at random."""
        if tuple(weight.shape) != tuple(self.wte.shape):
            raise ValueError(f"token embedding shape {tuple(weight.shape)} does not match {tuple(self.wte.shape)}")
        tied = self.head is self.wte
        self.wte = self._T(weight)
        if tied:
            self.head = self.wte
        self._bwd_ready = False
        if self.weight_format == "fp8":
            self.head_q = Fp8Weight.quantize(self.head, self.device)

    # ---------------------------------------------------------------- forward
    def forward(self, prefix_rows: Optional[Tensor], src: Tensor, pos: Tensor, mask: Tensor, B: int, S: int, *,
                labels: Optional[Tensor] = None, save: bool = False, logits: str = "none", pack: bool = False,
                lengths=None, n_scored: Optional[int] = None):
        """Run the decoder over ``B`` rows of ``S`` positions.

        ``src/pos/mask``: int32 [B,S] from ``ops.build_prefix_rows`` / ``build_fewshot_rows``;
        ``prefix_rows``: mapper output rows in the compute dtype.  ``logits``: "none" | "all" | "last".
        ``pack``: drop the padded positions (mask == 0) before the first GEMM (``eavqa_build_row_plan``): the loss
        and every attended position are unchanged, the work shrinks from B*S to sum(lengths) rows.  ``lengths``
        (host ints, attended positions per sample) saves the one device->host read of the packed row count.
        Returns a dict with ``loss``/``count`` (when labels), ``logits`` ([rows or B, vpad] fp32; with ``pack`` also
        ``flat_index`` mapping packed rows to b*S+s) and, when ``save``, the tape for :meth:`backward`.
        ``n_scored`` (host int, with ``pack`` and ``labels``): the number of positions that carry a label - exactly
        ``(labels != -100).sum()`` when every label has a predecessor position, as behind a prefix; the collate has it on
        the host.  Then only those rows go through the lm_head and its dgrad (``out["logits"]`` holds just them, ``sel``
        their packed row numbers).
        """
        c, T = self.cfg, self.dtype
        E, H, hd = c.n_embd, c.n_head, c.head_dim
        scale = hd ** -0.5
        cu = flat = row_labels = None
        M = B * S
        if pack:
            cu, src_r, pos_r, row_labels, flat = ops.build_row_plan(mask, labels, src, pos, True)
            M = int(sum(lengths)) if lengths is not None else int(cu[-1].item())
            src, pos, flat = src_r[:M], pos_r[:M], flat[:M]
            row_labels = row_labels[:M] if labels is not None else None
            attn_mask = None
        else:
            attn_mask = mask
        x = ops.embed_assemble(src, pos, self.wte, prefix_rows, self.wpe)
        tape = [] if save else None
        fold = self.fold_layernorm and M > 64
        if fold:
            self._prepare_fold()
        n_slots = (E + 63) // 64
        f32 = dict(device=self.device, dtype=torch.float32)
        fuse_q = self.weight_format == "fp8" and self.fuse_quantizer

        def ln(xx, g, b, stats):
            """ln_1 / ln_2 in front of a Linear: the bf16 operand - or, with e4m3 weights, its row-quantised form straight from the
            LayerNorm kernel (eavqa_layernorm_fwd_fp8: the bytes eavqa_quantize_rows_fp8 would produce, one pass less)."""
            if fuse_q:
                r = ops.layernorm_fwd_fp8(xx, g, b, c.eps, save_stats=stats)
                return ((r[0], r[1]), r[2], r[3]) if stats else r
            return ops.layernorm_fwd(xx, g, b, c.eps, T, save_stats=stats)

        xT = st = None                                      # with `fold`: the stream in the compute dtype and its row sums (from layer 0's FFN-down on)
        for li, L in enumerate(self.layers):
            if st is not None:                              # ln_1 folded into the QKV projection
                mean1, rstd1 = (torch.empty(M, **f32), torch.empty(M, **f32)) if save else (None, None)
                qkv = ops.gemm(xT, L.w_qkv_f, bias=L.d_qkv, ln_stats=st, ln_c=L.c_qkv, ln_eps=c.eps, ln_save=(mean1, rstd1) if save else None)
            else:
                if save:
                    a, mean1, rstd1 = ln(x, L.ln1_g, L.ln1_b, True)
                else:
                    a = ln(x, L.ln1_g, L.ln1_b, False)
                qkv = linear(a, L.w_qkv, bias=L.b_qkv)
            q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
            if save:
                ctx, lse = ops.attention_fwd(q, k, v, B, H, S, S, hd, key_mask=attn_mask, causal=True, scale=scale,
                                             save_lse=True, cu_seqlens=cu)
            else:
                ctx = ops.attention_fwd(q, k, v, B, H, S, S, hd, key_mask=attn_mask, causal=True, scale=scale, cu_seqlens=cu)
            if fold:
                # the out-projection leaves the stream a second time in the compute dtype + its row sums; ln_2 is a term of FFN-up's epilogue
                x1T = torch.empty((M, E), device=self.device, dtype=T)
                st1 = torch.empty((M, n_slots, 2), **f32)
                x1 = ops.gemm(ctx, L.w_o, bias=L.b_o, residual=x, out_f32=True, copy_out=x1T, stats_out=st1)
                mean2, rstd2 = (torch.empty(M, **f32), torch.empty(M, **f32)) if save else (None, None)
                u = torch.empty((M, c.ffn), device=self.device, dtype=T) if save else None
                f = ops.gemm(x1T, L.w_fc1_f, bias=L.d_fc1, act=c.act, aux_out=u, ln_stats=st1, ln_c=L.c_fc1, ln_eps=c.eps,
                             ln_save=(mean2, rstd2) if save else None)
                if li + 1 < len(self.layers):
                    xT = torch.empty((M, E), device=self.device, dtype=T)
                    st = torch.empty((M, n_slots, 2), **f32)
                    x2 = ops.gemm(f, L.w_fc2, bias=L.b_fc2, residual=x1, out_f32=True, copy_out=xT, stats_out=st)
                else:
                    x2 = ops.gemm(f, L.w_fc2, bias=L.b_fc2, residual=x1, out_f32=True)
            else:
                x1 = linear(ctx, L.w_o, bias=L.b_o, residual=x, out_f32=True)
                if save:
                    a2, mean2, rstd2 = ln(x1, L.ln2_g, L.ln2_b, True)
                    u = torch.empty((M, c.ffn), device=self.device, dtype=T)
                    f = linear(a2, L.w_fc1, bias=L.b_fc1, act=c.act, aux_out=u)
                else:
                    a2 = ln(x1, L.ln2_g, L.ln2_b, False)
                    f = linear(a2, L.w_fc1, bias=L.b_fc1, act=c.act)
                x2 = linear(f, L.w_fc2, bias=L.b_fc2, residual=x1, out_f32=True)
            if save:
                tape.append((x, mean1, rstd1, qkv, ctx, lse, x1, mean2, rstd2, u))
            x = x2
        out = {"rows": M, "flat_index": flat}
        if logits == "last" and labels is None and not pack:
            xl = x.view(B, S, E)[:, -1]                       # strided rows: only the last position is normalised
            hf = ops.layernorm_fwd(xl, self.lnf_g, self.lnf_b, c.eps, T)
            out["logits"] = self._head(hf)
            return out
        if save:
            hf, meanf, rstdf = ops.layernorm_fwd(x, self.lnf_g, self.lnf_b, c.eps, T, save_stats=True)
        else:
            hf = ops.layernorm_fwd(x, self.lnf_g, self.lnf_b, c.eps, T)
        out["hidden"] = hf
        if labels is not None or logits == "all":
            sel = sel_count = None
            if pack and labels is not None and n_scored is not None and logits != "all" and 0 < n_scored < M:
                sel, ce_labels, sel_count = ops.select_rows(row_labels, int(n_scored))
                lg = self._head(ops.gather_rows(hf, sel))
            else:
                lg = self._head(hf)
                ce_labels = row_labels if pack else labels
            out["logits"], out["sel"] = lg, sel
            if labels is not None:
                loss, count, row_lse = ops.ce_fwd(lg, ce_labels, self.vocab)
                if sel_count is not None:
                    # the host-side label count sized the compaction: more labelled rows than that would have been dropped
                    # silently, so the loss turns NaN instead (on the device, no synchronisation)
                    ops.guard_count(sel_count, int(n_scored), loss)
                out["loss"], out["count"] = loss, count
                if save:
                    out["tape"] = dict(layers=tape, x_last=x, meanf=meanf, rstdf=rstdf, logits=lg, labels=ce_labels, sel=sel,
                                       row_lse=row_lse, count=count, src=src, mask=attn_mask, cu=cu, B=B, S=S, M=M)
        return out

    def _head(self, hf: Tensor) -> Tensor:
        """lm_head GEMM into a [rows, vpad] fp32 buffer (pad columns are never read)."""
        lg = torch.empty((hf.shape[0], self.vpad), device=self.device, dtype=torch.float32)
        if self.head_q is not None:
            linear(hf, self.head_q, out=lg[:, :self.vocab])
            return lg
        if hf.shape[0] <= 64 and hf.dtype == torch.bfloat16 and self.vocab % 4 == 0 and self.cfg.n_embd % 32 == 0:
            # a decode step: stream the [V, E] head once with K split over workgroups (csrc/decode.hip)
            ops.splitk_finish(ops.gemm_splitk(hf, self.head[:self.vocab]), [lg[:, :self.vocab]])
        else:
            ops.gemm(hf, self.head, out=lg[:, :self.vocab])
        return lg

    # ---------------------------------------------------------------- backward (dgrad only)
    def backward(self, tape: dict, gloss: Tensor, n_prefix_rows: int) -> Tensor:
        """d loss / d prefix_rows.  ``gloss``: float32 [1] upstream gradient of the scalar loss."""
        self._prepare_backward()
        c, T = self.cfg, self.dtype
        E, H, hd = c.n_embd, c.n_head, c.head_dim
        B, S, M = tape["B"], tape["S"], tape["M"]
        scale = hd ** -0.5
        mask, cu = tape["mask"], tape["cu"]
        lowp = T != torch.float32

        # the residual-stream gradient lives in fp32 (dx); each LayerNorm backward also emits the copy in the
        # compute dtype that the next dgrad GEMM consumes as its A operand (no separate cast pass)
        dlog = ops.ce_bwd(tape["logits"], tape["labels"], self.vocab, tape["row_lse"], tape["count"], gloss, T, self.vpad)
        dhf = linear(dlog, self.head_t)                                   # [M,E] (or [n_scored,E])
        if tape.get("sel") is not None:
            dhf = ops.scatter_rows(dhf, tape["sel"], M)                      # rows without a label get no gradient here
        fuse_q = self.weight_format == "fp8" and self.fuse_quantizer
        if fuse_q:
            # the copy of the stream gradient that the next dgrad GEMM multiplies leaves the LayerNorm backward already row-quantised
            dxq = (torch.empty((M, E), device=self.device, dtype=torch.uint8), torch.empty(M, device=self.device, dtype=torch.float32))
            dxT = dxq

            def ln_bwd(xx, dy, g, mean, rstd, **kw):
                return ops.layernorm_bwd_fp8(xx, dy, g, mean, rstd, dxq[0], dxq[1], **kw)
        else:
            dxT = torch.empty((M, E), device=self.device, dtype=T) if lowp else None

            def ln_bwd(xx, dy, g, mean, rstd, **kw):
                return ops.layernorm_bwd(xx, dy, g, mean, rstd, lowp_out=dxT, **kw)
        dx = ln_bwd(tape["x_last"], dhf, self.lnf_g, tape["meanf"], tape["rstdf"])
        for L, (x, mean1, rstd1, qkv, ctx, lse, x1, mean2, rstd2, u) in zip(reversed(self.layers), reversed(tape["layers"])):
            du = linear(dxT if lowp else dx, L.w_fc2_t, act=c.act, aux_in=u)   # (dx W2) * act'(u)   [M,F]
            da2 = linear(du, L.w_fc1_t)                                    # [M,E]
            dx1 = ln_bwd(x1, da2, L.ln2_g, mean2, rstd2, dres=dx, out=dx)
            dctx = linear(dxT if lowp else dx1, L.w_o_t)                   # [M,E]
            dqkv = torch.empty_like(qkv)
            q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
            ops.attention_bwd(q, k, v, ctx, dctx, lse, B, H, S, S, hd, key_mask=mask, causal=True, scale=scale,
                              dq=dqkv[:, :E], dk=dqkv[:, E:2 * E], dv=dqkv[:, 2 * E:], cu_seqlens=cu)
            da = linear(dqkv, L.w_qkv_t)                                   # [M,E]
            dx = ln_bwd(x, da, L.ln1_g, mean1, rstd1, dres=dx1, out=dx1)
        return ops.embed_assemble_bwd(tape["src"], dx, n_prefix_rows, T)


def _same(a: Tensor, b: Tensor) -> bool:
    try:
        return a.data_ptr() == b.data_ptr() or bool(torch.equal(a.to(b.device, b.dtype), b))
    except Exception:
        return False
