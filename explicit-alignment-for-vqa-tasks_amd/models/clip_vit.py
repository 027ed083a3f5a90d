"""CLIP vision tower (``model.encode_image``) executed by the HIP kernels.

The reference calls OpenAI-CLIP's ``encode_image`` offline
(src/tools/extract_clip_embeddings_conceptual_captions.py:83-88,
src/tools/extract_contrastive_image_embeddings.py:59-63) and stores float32 ``[D]`` rows; this build
runs the same arithmetic in the loop.  Weights use the HF ``CLIPVisionModelWithProjection`` key
names (transformers/models/clip/modeling_clip.py:138-219 embeddings, :280-385 layers, :898-960 tower
+ projection; QuickGELU transformers/activations.py:117-123), which is the same graph as OpenAI's
``VisionTransformer``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch

from .. import ops

Tensor = torch.Tensor


@dataclass
class ViTConfig:
    width: int
    n_layer: int
    n_head: int
    mlp: int
    patch: int
    image: int
    proj: int
    eps: float = 1e-5
    act: str = "quick_gelu"

    @property
    def n_patch(self) -> int:
        return (self.image // self.patch) ** 2


KNOWN_VITS = {
    "ViT-B/32": ViTConfig(768, 12, 12, 3072, 32, 224, 512),
    "ViT-B/16": ViTConfig(768, 12, 12, 3072, 16, 224, 512),
    "ViT-L/14": ViTConfig(1024, 24, 16, 4096, 14, 224, 768),
    "ViT-L/14@336px": ViTConfig(1024, 24, 16, 4096, 14, 336, 768),
}


def random_init_vit_state_dict(cfg: ViTConfig, seed: int = 2021, device="cpu") -> Dict[str, Tensor]:
    """Seeded random-init weights under HF key names (std 0.02-ish like HF's CLIP init; LayerNorm 1/0)."""
    g = torch.Generator(device=device).manual_seed(seed)
    W = cfg.width

    def n(*shape, std=0.02):
        return torch.randn(*shape, generator=g, device=device) * std

    ones, zeros = (lambda k: torch.ones(k, device=device)), (lambda k: torch.zeros(k, device=device))
    p = "vision_model."
    sd = {
        p + "embeddings.class_embedding": n(W, std=W ** -0.5),
        p + "embeddings.patch_embedding.weight": n(W, 3, cfg.patch, cfg.patch),
        p + "embeddings.position_embedding.weight": n(cfg.n_patch + 1, W),
        p + "pre_layrnorm.weight": ones(W), p + "pre_layrnorm.bias": zeros(W),
        p + "post_layernorm.weight": ones(W), p + "post_layernorm.bias": zeros(W),
        "visual_projection.weight": n(cfg.proj, W, std=W ** -0.5),
    }
    for i in range(cfg.n_layer):
        q = f"{p}encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[q + f"self_attn.{nm}.weight"], sd[q + f"self_attn.{nm}.bias"] = n(W, W, std=W ** -0.5), zeros(W)
        sd[q + "layer_norm1.weight"], sd[q + "layer_norm1.bias"] = ones(W), zeros(W)
        sd[q + "layer_norm2.weight"], sd[q + "layer_norm2.bias"] = ones(W), zeros(W)
        sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"] = n(cfg.mlp, W), zeros(cfg.mlp)
        sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"] = n(W, cfg.mlp), zeros(W)
    return sd


class ClipVisionEncoder:
    """Frozen CLIP ViT: ``encode_image(pixels [B,3,H,W] float32) -> float32 [B, D]``."""

    def __init__(self, cfg: ViTConfig, state_dict: Dict[str, Tensor], dtype: torch.dtype = torch.bfloat16, device="cuda",
                 stream_dtype: torch.dtype = None):
        """``stream_dtype``: storage type of the residual stream.  The tower is frozen and forward-only, so with bf16 operands the
        stream is kept in 16 bits: the out-projection / FFN-down epilogues and both LayerNorms then move half the bytes of a float32
        stream (round 2: 2 x 168 MB per launch at 160 images).  Default in bf16 mode: ``torch.float16`` - what OpenAI CLIP itself holds
        on the GPU (the whole tower in fp16, extract_clip_embeddings_conceptual_captions.py:26,86): 11 significant bits, so the
        2 x n_layer roundings of the residual sums stay below the bf16 operands' own error (measured on MI355X, ViT-L/14, max |d
        image_embeds| against the fp32 oracle: float32 stream 1.6e-2, bfloat16 stream 3.4e-2).  ``torch.bfloat16`` and
        ``torch.float32`` (the round-2 stream) are accepted too; fp32 mode always streams in float32."""
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        if stream_dtype is None:
            stream_dtype = torch.float16 if dtype == torch.bfloat16 else torch.float32
        self.stream_dtype = stream_dtype
        if dtype == torch.float32 and stream_dtype != torch.float32:
            raise ValueError("fp32 mode streams in float32")
        if stream_dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("stream_dtype must be float32, bfloat16 or float16")
        T = lambda t: t.to(device=self.device, dtype=dtype).contiguous()
        F = lambda t: t.to(device=self.device, dtype=torch.float32).contiguous()
        p = "vision_model."
        W = cfg.width
        K = 3 * cfg.patch * cfg.patch
        self.kpad = (K + 31) // 32 * 32       # ViT-L/14: K = 588 -> 608 (zero columns contribute nothing)
        wp = torch.zeros((W, self.kpad), dtype=torch.float32)
        wp[:, :K] = state_dict[p + "embeddings.patch_embedding.weight"].reshape(W, K).float().cpu()
        self.w_patch = T(wp)
        self.cls = F(state_dict[p + "embeddings.class_embedding"])
        self.pos = F(state_dict[p + "embeddings.position_embedding.weight"])
        self.pre_g, self.pre_b = F(state_dict[p + "pre_layrnorm.weight"]), F(state_dict[p + "pre_layrnorm.bias"])
        self.post_g, self.post_b = F(state_dict[p + "post_layernorm.weight"]), F(state_dict[p + "post_layernorm.bias"])
        self.w_proj = T(state_dict["visual_projection.weight"])
        self.layers = []
        for i in range(cfg.n_layer):
            q = f"{p}encoder.layers.{i}."
            self.layers.append(dict(
                ln1_g=F(state_dict[q + "layer_norm1.weight"]), ln1_b=F(state_dict[q + "layer_norm1.bias"]),
                w_qkv=T(torch.cat([state_dict[q + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)),
                b_qkv=F(torch.cat([state_dict[q + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)),
                w_o=T(state_dict[q + "self_attn.out_proj.weight"]), b_o=F(state_dict[q + "self_attn.out_proj.bias"]),
                ln2_g=F(state_dict[q + "layer_norm2.weight"]), ln2_b=F(state_dict[q + "layer_norm2.bias"]),
                w_fc1=T(state_dict[q + "mlp.fc1.weight"]), b_fc1=F(state_dict[q + "mlp.fc1.bias"]),
                w_fc2=T(state_dict[q + "mlp.fc2.weight"]), b_fc2=F(state_dict[q + "mlp.fc2.bias"]),
            ))

    @torch.no_grad()
    def encode_image(self, pixels: Tensor) -> Tensor:
        c, T = self.cfg, self.dtype
        B = pixels.shape[0]
        if pixels.shape[-1] != c.image or pixels.shape[-2] != c.image:
            raise ValueError(f"Input image size ({pixels.shape[-2]}*{pixels.shape[-1]}) doesn't match model ({c.image}*{c.image}).")
        W, H, N = c.width, c.n_head, c.n_patch + 1
        hd = W // H
        patches = ops.patchify(pixels.to(self.device), c.patch, T, self.kpad)
        pe = ops.gemm(patches, self.w_patch)
        x = ops.vit_assemble(pe, self.cls, self.pos, B, c.n_patch)
        ST = self.stream_dtype
        x = ops.layernorm_fwd(x, self.pre_g, self.pre_b, c.eps, ST)                # the residual stream: ``stream_dtype``
        x1 = torch.empty_like(x)
        for L in self.layers:
            a = ops.layernorm_fwd(x, L["ln1_g"], L["ln1_b"], c.eps, T)
            qkv = ops.gemm(a, L["w_qkv"], bias=L["b_qkv"])
            ctx = ops.attention_fwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], B, H, N, N, hd, causal=False, scale=hd ** -0.5)
            ops.gemm(ctx, L["w_o"], bias=L["b_o"], residual=x, out=x1)
            a2 = ops.layernorm_fwd(x1, L["ln2_g"], L["ln2_b"], c.eps, T)
            f = ops.gemm(a2, L["w_fc1"], bias=L["b_fc1"], act=c.act)
            ops.gemm(f, L["w_fc2"], bias=L["b_fc2"], residual=x1, out=x)
        pooled = ops.layernorm_fwd(x.view(B, N, W)[:, 0], self.post_g, self.post_b, c.eps, T)
        return ops.gemm(pooled, self.w_proj, out_f32=True)


class EncodeAhead:
    """One batch of look-ahead for a frozen image tower in an inference loop: ``submit(pixels)`` issues ``encode_image`` on a second HIP
    stream and returns a ticket, ``result(ticket)`` makes the calling stream wait for it and hands over the embeddings - so the tower of
    batch i + 1 runs while batch i is being generated (prefill + decode steps, which leave the matrix cores idle between their short kernels).

        ahead = EncodeAhead(vit); t = ahead.submit(px[0])
        for i in range(n):
            emb = ahead.result(t)
            if i + 1 < n: t = ahead.submit(px[i + 1])          # before generating batch i
            model.generate_fewshot(..., emb, ...)

    The embeddings are those of ``vit.encode_image`` bit for bit (same kernels, another stream).  Measured on the few-shot batch of BASELINE
    configs[3] (ViT-L/14 + OPT-2.7B): 84.2 -> 81.3 ms per 32 questions - the tower's 1024-thread workgroups fill the register file, so the
    decode kernels only get a CU when one of them retires and most of the two streams' work still runs one after the other."""

    def __init__(self, vit: "ClipVisionEncoder"):
        self.vit = vit
        self.side = torch.cuda.Stream(device=vit.device)

    def submit(self, pixels: Tensor):
        main = torch.cuda.current_stream(self.vit.device)
        self.side.wait_stream(main)                      # whatever produced `pixels` (and freed the buffers the allocator may reuse)
        if pixels.is_cuda:
            pixels.record_stream(self.side)              # the caller may drop its reference before the side stream has read it
        with torch.cuda.stream(self.side):
            emb = self.vit.encode_image(pixels)
            done = torch.cuda.Event()
            done.record(self.side)
        return emb, done

    def result(self, ticket) -> Tensor:
        emb, done = ticket
        main = torch.cuda.current_stream(self.vit.device)
        main.wait_event(done)
        emb.record_stream(main)                          # allocated on the side stream, consumed on this one
        return emb

