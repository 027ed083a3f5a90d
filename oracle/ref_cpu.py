"""Plain-torch fp32 CPU restatement of the reference hot path (TEST INFRASTRUCTURE).

Every function cites the reference file:line it follows.  Paths are relative
to ``/root/reference`` unless they start with ``HF:`` which means
``/usr/local/lib/python3.10/dist-packages/transformers`` (transformers 5.15.0;
the reference pins 4.12.5 and vendors nothing - SURVEY.md section 8c).

No HuggingFace / reference import happens here: weights come in as plain
``{name: tensor}`` dicts using the HF state-dict key names, so the same dict
drives the reference model (fixture generation), this oracle and the HIP path.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor
NEG = torch.finfo(torch.float32).min  # HF eager masks add finfo.min (HF:masking_utils)

__all__ = [
    "layer_norm", "gelu_new", "quick_gelu", "mlp_mapper", "transformer_mapper",
    "attention_bias", "gpt2_hidden", "gpt2_logits", "opt_hidden", "opt_logits",
    "lm_logits", "causal_lm_loss", "clipcap_forward", "clipcap_generate",
    "label_mask_vqa", "label_mask_cc", "insert_prefix_into_input",
    "clip_vit_encode", "adamw_step", "mapper_project",
    "t5_rms_norm", "t5_relative_bucket", "t5_position_bias", "t5_encoder", "t5_decoder", "t5_lm_logits", "t5_shift_right",
    "vct0_forward", "vct0_generate",
]


# --------------------------------------------------------------------------
# elementwise / normalisation
# --------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float = 1e-5) -> Tensor:
    """torch.nn.LayerNorm over the last dim (biased variance)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) / torch.sqrt(var + eps)
    if w is not None:
        y = y * w
    if b is not None:
        y = y + b
    return y


def gelu_new(x: Tensor) -> Tensor:
    """HF:activations.py:59-66 (NewGELUActivation, tanh approximation)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def quick_gelu(x: Tensor) -> Tensor:
    """HF:activations.py:117-123 (QuickGELUActivation)."""
    return x * torch.sigmoid(1.702 * x)


_ACT = {
    "gelu_new": gelu_new,
    "quick_gelu": quick_gelu,
    "relu": torch.relu,
    "tanh": torch.tanh,
    "none": lambda x: x,
}


# --------------------------------------------------------------------------
# mapping networks  (src/models/clipcap.py)
# --------------------------------------------------------------------------
def mlp_mapper(x: Tensor, p: Dict[str, Tensor]) -> Tensor:
    """``MLP`` clipcap.py:31-42 as built at :256-262: Linear -> Tanh -> Linear.

    Keys follow ``nn.Sequential``: ``model.0.weight [H,D]``, ``model.0.bias``,
    ``model.2.weight [E*L,H]``, ``model.2.bias``.
    """
    h = torch.tanh(x @ p["model.0.weight"].T + p["model.0.bias"])
    return h @ p["model.2.weight"].T + p["model.2.bias"]


def _mapper_mha(x: Tensor, y: Tensor, p: Dict[str, Tensor], pre: str, heads: int) -> Tensor:
    """``MultiHeadAttention.forward`` clipcap.py:81-104 (bias=False for q/kv)."""
    b, n, c = x.shape
    m = y.shape[1]
    hd = c // heads
    q = (x @ p[pre + "to_queries.weight"].T).reshape(b, n, heads, hd)
    kv = (y @ p[pre + "to_keys_values.weight"].T).reshape(b, m, 2, heads, hd)
    k, v = kv[:, :, 0], kv[:, :, 1]
    att = torch.einsum("bnhd,bmhd->bnmh", q, k) * (hd ** -0.5)
    att = att.softmax(dim=2)
    out = torch.einsum("bnmh,bmhd->bnhd", att, v).reshape(b, n, c)
    return out @ p[pre + "project.weight"].T + p[pre + "project.bias"]


def transformer_mapper(x: Tensor, p: Dict[str, Tensor], clip_length: int, num_layers: int,
                       heads: int = 8) -> Tensor:
    """``TransformerMapper.forward`` clipcap.py:213-221 with ``Transformer``
    (:141-157, enc_dec=False), ``TransformerLayer`` (:114-117, pre-LN,
    mlp_ratio 2.0, ReLU) and 8 heads (:233).  Returns ``[B, L, E]``."""
    B = x.shape[0]
    h = (x @ p["linear.weight"].T + p["linear.bias"]).view(B, clip_length, -1)
    const = p["prefix_const"].unsqueeze(0).expand(B, *p["prefix_const"].shape)
    h = torch.cat((h, const), dim=1)
    for i in range(num_layers):
        pre = f"transformer.layers.{i}."
        a = layer_norm(h, p[pre + "norm1.weight"], p[pre + "norm1.bias"])
        h = h + _mapper_mha(a, a, p, pre + "attn.", heads)
        m = layer_norm(h, p[pre + "norm2.weight"], p[pre + "norm2.bias"])
        m = torch.relu(m @ p[pre + "mlp.fc1.weight"].T + p[pre + "mlp.fc1.bias"])
        h = h + (m @ p[pre + "mlp.fc2.weight"].T + p[pre + "mlp.fc2.bias"])
    return h[:, clip_length:]


def mapper_project(prefix: Tensor, mapper: Dict[str, Tensor], mapping_type: str, prefix_length: int,
                   embed: int, clip_length: Optional[int] = None, num_layers: int = 8) -> Tensor:
    """``self.clip_project(prefix).view(-1, L, E)`` clipcap.py:318-320."""
    if mapping_type == "mlp":
        out = mlp_mapper(prefix, mapper)
    else:
        # TransformerMapper.forward does x.view(x.shape[0], clip_length, -1): 2-D input only
        out = transformer_mapper(prefix, mapper, clip_length, num_layers)
    return out.reshape(-1, prefix_length, embed)


# --------------------------------------------------------------------------
# causal LMs (HF GPT-2 / OPT)
# --------------------------------------------------------------------------
def _lin(cfg: dict, x: Tensor, w_out_in: Tensor, b: Optional[Tensor]) -> Tensor:
    """``x @ w^T + b`` (``w`` as ``[out, in]``).  ``cfg["linear_fn"]``, when present, replaces the product: the fp8 numerics
    model of oracle/fp8_sim.py plugs in there so that the LM code below stays the single restatement of HF's forward."""
    fn = cfg.get("linear_fn")
    y = fn(x, w_out_in) if fn is not None else x @ w_out_in.T
    return y if b is None else y + b



def attention_bias(attention_mask: Tensor, causal: bool = True) -> Tensor:
    """Additive ``[B,1,S,S]`` bias: ``finfo.min`` where the key is padded or (causal) in
    the future, else 0 - what HF's eager path adds (HF:models/gpt2/modeling_gpt2.py:54-72,
    HF:masking_utils create_causal_mask)."""
    B, S = attention_mask.shape
    keep = (attention_mask != 0)[:, None, None, :].expand(B, 1, S, S)
    if causal:
        tri = torch.ones(S, S, dtype=torch.bool).tril()
        keep = keep & tri[None, None]
    return torch.where(keep, 0.0, NEG)


def _sdpa(q: Tensor, k: Tensor, v: Tensor, bias: Optional[Tensor], scale: float) -> Tensor:
    """Eager attention: softmax(q k^T * scale + bias) v;  q,k,v ``[B,H,S,hd]``."""
    w = (q @ k.transpose(-1, -2)) * scale
    if bias is not None:
        w = w + bias
    w = torch.softmax(w, dim=-1)
    return w @ v


def gpt2_hidden(sd: Dict[str, Tensor], cfg: dict, inputs_embeds: Tensor, attention_mask: Tensor) -> Tensor:
    """``GPT2Model.forward`` HF:models/gpt2/modeling_gpt2.py:571-620 (positions are
    ``arange(S)``, NOT mask-aware), block :246-309, attention :75-226 (Conv1D weight is
    ``[in,out]``), MLP :229-243 (gelu_new).  Returns ``ln_f`` output ``[B,S,E]``."""
    B, S, E = inputs_embeds.shape
    H = cfg["n_head"]
    hd = E // H
    eps = cfg.get("eps", 1e-5)
    act = _ACT[cfg.get("act", "gelu_new")]
    h = inputs_embeds + sd["transformer.wpe.weight"][:S][None]
    bias = attention_bias(attention_mask)
    for i in range(cfg["n_layer"]):
        p = f"transformer.h.{i}."
        a = layer_norm(h, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps)
        qkv = _lin(cfg, a, sd[p + "attn.c_attn.weight"].T, sd[p + "attn.c_attn.bias"])
        q, k, v = qkv.split(E, dim=2)
        q = q.view(B, S, H, hd).transpose(1, 2)
        k = k.view(B, S, H, hd).transpose(1, 2)
        v = v.view(B, S, H, hd).transpose(1, 2)
        ctx = _sdpa(q, k, v, bias, hd ** -0.5).transpose(1, 2).reshape(B, S, E)
        h = h + _lin(cfg, ctx, sd[p + "attn.c_proj.weight"].T, sd[p + "attn.c_proj.bias"])
        m = layer_norm(h, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps)
        m = act(_lin(cfg, m, sd[p + "mlp.c_fc.weight"].T, sd[p + "mlp.c_fc.bias"]))
        h = h + _lin(cfg, m, sd[p + "mlp.c_proj.weight"].T, sd[p + "mlp.c_proj.bias"])
    return layer_norm(h, sd["transformer.ln_f.weight"], sd["transformer.ln_f.bias"], eps)


def gpt2_logits(sd, cfg, inputs_embeds, attention_mask) -> Tensor:
    """``GPT2LMHeadModel`` lm_head (tied to wte, no bias) HF:...modeling_gpt2.py:698."""
    w = sd.get("lm_head.weight", sd["transformer.wte.weight"])
    return _lin(cfg, gpt2_hidden(sd, cfg, inputs_embeds, attention_mask), w, None)


def opt_hidden(sd: Dict[str, Tensor], cfg: dict, inputs_embeds: Tensor, attention_mask: Tensor) -> Tensor:
    """``OPTDecoder.forward`` HF:models/opt/modeling_opt.py:319-397: learned positions with
    offset 2 from ``cumsum(mask)*mask-1`` (:45-70), pre-LN layers (:184-254, ReLU), q scaled
    BEFORE q.k (:141), ``final_layer_norm``.  ``project_in/out`` (350m only) unsupported."""
    B, S, E = inputs_embeds.shape
    H = cfg["n_head"]
    hd = E // H
    eps = cfg.get("eps", 1e-5)
    act = _ACT[cfg.get("act", "relu")]
    pre = "model.decoder."
    am = attention_mask
    pos = (torch.cumsum(am, dim=1) * am - 1).long() + 2
    h = inputs_embeds + sd[pre + "embed_positions.weight"][pos]
    bias = attention_bias(attention_mask)
    for i in range(cfg["n_layer"]):
        p = f"{pre}layers.{i}."
        a = layer_norm(h, sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], eps)
        q = _lin(cfg, a, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * (hd ** -0.5)
        k = _lin(cfg, a, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = _lin(cfg, a, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q = q.view(B, S, H, hd).transpose(1, 2)
        k = k.view(B, S, H, hd).transpose(1, 2)
        v = v.view(B, S, H, hd).transpose(1, 2)
        ctx = _sdpa(q, k, v, bias, 1.0).transpose(1, 2).reshape(B, S, E)
        h = h + _lin(cfg, ctx, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        m = layer_norm(h, sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], eps)
        m = act(_lin(cfg, m, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
        h = h + _lin(cfg, m, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    return layer_norm(h, sd[pre + "final_layer_norm.weight"], sd[pre + "final_layer_norm.bias"], eps)


def opt_logits(sd, cfg, inputs_embeds, attention_mask) -> Tensor:
    """``OPTForCausalLM`` lm_head HF:models/opt/modeling_opt.py:443-538 (tied, no bias)."""
    w = sd.get("lm_head.weight", sd["model.decoder.embed_tokens.weight"])
    return _lin(cfg, opt_hidden(sd, cfg, inputs_embeds, attention_mask), w, None)


def lm_logits(sd, cfg, inputs_embeds, attention_mask) -> Tensor:
    return (gpt2_logits if cfg["arch"] == "gpt2" else opt_logits)(sd, cfg, inputs_embeds, attention_mask)


def _wte(sd, cfg) -> Tensor:
    return sd["transformer.wte.weight"] if cfg["arch"] == "gpt2" else sd["model.decoder.embed_tokens.weight"]


def causal_lm_loss(logits: Tensor, labels: Tensor, ignore_index: int = -100) -> Tensor:
    """``ForCausalLMLoss`` HF:loss/loss_utils.py:49-71 + ``fixed_cross_entropy`` :32-46:
    labels padded right with -100, shifted left by one, mean CE over non-ignored."""
    V = logits.shape[-1]
    lab = torch.nn.functional.pad(labels, (0, 1), value=ignore_index)[..., 1:].contiguous()
    lg = logits.float().reshape(-1, V)
    lab = lab.reshape(-1)
    keep = lab != ignore_index
    lse = torch.logsumexp(lg, dim=-1)
    picked = lg.gather(1, lab.clamp(min=0)[:, None])[:, 0]
    nll = (lse - picked) * keep
    return nll.sum() / keep.sum()


# --------------------------------------------------------------------------
# ClipCaptionModel (src/models/clipcap.py:240-471)
# --------------------------------------------------------------------------
def _prefix_inputs(sd, cfg, mapper, mcfg, question_tokens, prefix, question_mask):
    """Mask concat + wte gather + mapper + embedding concat, clipcap.py:303-321 / :353-381."""
    B = question_tokens.shape[0]
    L = mcfg["prefix_length"]
    wte = _wte(sd, cfg)
    E = wte.shape[1]
    prefix_mask = torch.ones((B, L))
    attention_mask = torch.cat((prefix_mask, question_mask), dim=1)
    embedding_text = wte[question_tokens]
    proj = mapper_project(prefix, mapper, mcfg["mapping_type"], L, E, mcfg.get("clip_length"),
                          mcfg.get("num_layers", 8))
    return torch.cat((proj, embedding_text), dim=1), attention_mask


def clipcap_forward(sd, cfg, mapper, mcfg, question_tokens: Tensor, prefix: Tensor,
                    question_mask: Tensor, labels: Optional[Tensor] = None) -> Tuple[Optional[Tensor], Tensor]:
    """``ClipCaptionModel.forward`` clipcap.py:290-342 -> ``(loss, logits[B,L+T,V])``."""
    emb, am = _prefix_inputs(sd, cfg, mapper, mcfg, question_tokens, prefix, question_mask)
    logits = lm_logits(sd, cfg, emb, am)
    loss = None
    if labels is not None:
        B, L = question_tokens.shape[0], mcfg["prefix_length"]
        full = torch.cat((torch.full((B, L), -100, dtype=torch.int64), labels), dim=1)  # :323-335
        loss = causal_lm_loss(logits, full)
    return loss, logits


def clipcap_generate(sd, cfg, mapper, mcfg, question_tokens: Tensor, prefix: Tensor, question_mask: Tensor,
                     max_length: int = 10, pad_token_id: Optional[int] = None,
                     eos_token_id: Optional[int] = None, output_scores: bool = False):
    """``generate`` + ``_generate_from_embeddings`` clipcap.py:344-471: greedy decode by full
    re-forward each step (no KV cache); the embedding appended is that of the RAW argmax
    (:423) while the emitted token is pad-substituted for finished rows (:431-434, float
    math on ``unfinished_sequences`` :408-410); early stop when all rows finished (:463).
    ``output_scores``: also the [B, steps] log-softmax of the raw argmax (``torch.log(torch.stack(
    outputs.scores).softmax(-1))`` gathered at the token, few_shot_vqa_executor.py:314-321)."""
    emb, am = _prefix_inputs(sd, cfg, mapper, mcfg, question_tokens, prefix, question_mask)
    wte = _wte(sd, cfg)
    B = emb.shape[0]
    unfinished = torch.ones(B, 1)
    tokens = None
    scores = []
    for _ in range(max_length):
        logits = lm_logits(sd, cfg, emb, am)
        nxt = torch.argmax(logits[:, -1, :], -1).unsqueeze(1)
        scores.append(torch.log_softmax(logits[:, -1, :].float(), -1).gather(1, nxt))
        nxt_embed = wte[nxt]
        if eos_token_id is not None:
            if pad_token_id is None:
                raise ValueError("If `eos_token_id` is defined, make sure that `pad_token_id` is defined.")
            nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
        tokens = nxt if tokens is None else torch.cat((tokens, nxt), dim=1)
        emb = torch.cat((emb, nxt_embed), dim=1)
        am = torch.cat([am, torch.ones((B, 1))], dim=-1)
        if eos_token_id is not None:
            unfinished = unfinished.mul((nxt != eos_token_id).long())
        if unfinished.max() == 0:
            break
    ids = tokens.cpu().numpy().astype(int).tolist()
    if output_scores:
        return ids, torch.cat(scores, dim=1)
    return ids


# --------------------------------------------------------------------------
# integer / index work (bit-exact)
# --------------------------------------------------------------------------
def label_mask_vqa(input_ids: Tensor, pad_token_id: int, bos_token_id: int) -> Tensor:
    """``ClipCapExecutor.training_step`` label construction src/trainers/clipcap_exector.py:134-150
    (restated loop-for-loop; the executor itself is not importable here)."""
    labels = input_ids.detach().clone()
    labels[labels == pad_token_id] = -100
    for i in range(labels.shape[0]):
        answer_tokens = False
        for j in range(labels.shape[1]):
            token = int(labels[i, j])
            if token == -100:
                labels[i, j] = pad_token_id
                break
            if token == bos_token_id:
                answer_tokens = True
                labels[i, j] = -100
                continue
            if answer_tokens:
                continue
            labels[i, j] = -100
    return labels


def label_mask_cc(input_ids: Tensor, pad_token_id: int) -> Tensor:
    """Conceptual-Captions collate: ``labels[labels == pad] = -100``
    src/data_loader_manager/data_loader_conceptual_captions.py:94-95."""
    labels = input_ids.detach().clone()
    labels[labels == pad_token_id] = -100
    return labels


def insert_prefix_into_input(prefix_length: int, num_shots: int, question_tokens: Tensor, text_embeddings: Tensor,
                             prefix_projections: Tensor, question_masks: Tensor,
                             special_token_id: int = 32099) -> Tuple[Tensor, Tensor]:
    """``VCT0Model.insert_prefix_into_input`` src/models/vct0.py:494-533, restated as an
    explicit per-row walk: the n-th sentinel (ids ``special_token_id - i``, i=0..num_shots,
    in order of appearance) is replaced by the L prefix vectors of image n; text shifts right.
    Output length ``T + (L-1)*(num_shots+1)``.  Every row must hold exactly num_shots+1
    sentinels (the reference's ``.view`` at :512 fails otherwise)."""
    B, T = question_tokens.shape
    E = text_embeddings.shape[-1]
    n_img = num_shots + 1
    L = prefix_length
    T_out = T + (L - 1) * n_img
    sentinels = {special_token_id - i for i in range(n_img)}
    pp = prefix_projections.reshape(B, n_img, L, E)
    emb = torch.empty(B, T_out, E, dtype=text_embeddings.dtype)
    msk = torch.empty(B, T_out, dtype=torch.int64)
    for b in range(B):
        o = 0
        n = 0
        for t in range(T):
            if int(question_tokens[b, t]) in sentinels:
                if n >= n_img:
                    raise ValueError("row holds more sentinel tokens than images")
                emb[b, o:o + L] = pp[b, n]
                msk[b, o:o + L] = 1
                o += L
                n += 1
            else:
                emb[b, o] = text_embeddings[b, t]
                msk[b, o] = question_masks[b, t]
                o += 1
        if n != n_img:
            raise ValueError("row holds fewer sentinel tokens than images")
    return emb, msk


# --------------------------------------------------------------------------
# CLIP vision tower (offline in the reference, in-loop in the build)
# --------------------------------------------------------------------------
def clip_vit_encode(sd: Dict[str, Tensor], cfg: dict, pixel_values: Tensor) -> Tensor:
    """``model.encode_image`` call sites src/tools/extract_clip_embeddings_conceptual_captions.py:83-88,
    src/tools/extract_contrastive_image_embeddings.py:59-63; arithmetic of OpenAI CLIP
    ``VisionTransformer`` == HF ``CLIPVisionModelWithProjection``
    HF:models/clip/modeling_clip.py:138-219 (patch conv stride=kernel, no bias; class token;
    +pos), :898-960 (pre_layrnorm, encoder, post_layernorm on token 0), :280-385 (attention
    scale hd^-0.5, quick_gelu MLP), :949-950 (visual_projection, no bias).  ``[B,3,H,W] -> [B,D]``."""
    p = "vision_model."
    B = pixel_values.shape[0]
    W = cfg["width"]
    H = cfg["n_head"]
    hd = W // H
    ps = cfg["patch"]
    eps = cfg.get("eps", 1e-5)
    act = _ACT[cfg.get("act", "quick_gelu")]
    w = sd[p + "embeddings.patch_embedding.weight"]
    g = pixel_values.shape[-1] // ps
    patches = pixel_values.reshape(B, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * ps * ps)
    x = patches @ w.reshape(W, -1).T
    cls = sd[p + "embeddings.class_embedding"].expand(B, 1, W)
    x = torch.cat([cls, x], dim=1) + sd[p + "embeddings.position_embedding.weight"][None]
    x = layer_norm(x, sd[p + "pre_layrnorm.weight"], sd[p + "pre_layrnorm.bias"], eps)
    N = x.shape[1]
    for i in range(cfg["n_layer"]):
        q_ = f"{p}encoder.layers.{i}."
        a = layer_norm(x, sd[q_ + "layer_norm1.weight"], sd[q_ + "layer_norm1.bias"], eps)
        q = (a @ sd[q_ + "self_attn.q_proj.weight"].T + sd[q_ + "self_attn.q_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
        k = (a @ sd[q_ + "self_attn.k_proj.weight"].T + sd[q_ + "self_attn.k_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
        v = (a @ sd[q_ + "self_attn.v_proj.weight"].T + sd[q_ + "self_attn.v_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
        ctx = _sdpa(q, k, v, None, hd ** -0.5).transpose(1, 2).reshape(B, N, W)
        x = x + (ctx @ sd[q_ + "self_attn.out_proj.weight"].T + sd[q_ + "self_attn.out_proj.bias"])
        m = layer_norm(x, sd[q_ + "layer_norm2.weight"], sd[q_ + "layer_norm2.bias"], eps)
        m = act(m @ sd[q_ + "mlp.fc1.weight"].T + sd[q_ + "mlp.fc1.bias"])
        x = x + (m @ sd[q_ + "mlp.fc2.weight"].T + sd[q_ + "mlp.fc2.bias"])
    pooled = layer_norm(x[:, 0], sd[p + "post_layernorm.weight"], sd[p + "post_layernorm.bias"], eps)
    return pooled @ sd["visual_projection.weight"].T


# --------------------------------------------------------------------------
# optimiser (src/trainers/clipcap_exector.py:79-81 -> torch.optim.AdamW defaults)
# --------------------------------------------------------------------------
def adamw_step(param: Tensor, grad: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
               beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.01) -> None:
    """One in-place ``torch.optim.AdamW`` update (decoupled decay first, bias-corrected
    moments, ``denom = sqrt(v)/sqrt(1-b2^t) + eps``) - the single-tensor path of
    torch/optim/adamw.py as used by ``configure_optimizers`` clipcap_exector.py:79-81."""
    param.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(grad, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------
# T5 / T0 encoder-decoder and VCT0Model (src/models/vct0.py:301-491)
# --------------------------------------------------------------------------
def t5_rms_norm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    """``T5LayerNorm`` HF:models/t5/modeling_t5.py:50-72: no mean subtraction, no bias."""
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def t5_relative_bucket(rel: Tensor, bidirectional: bool, num_buckets: int = 32, max_distance: int = 128) -> Tensor:
    """``T5Attention._relative_position_bucket`` HF:...modeling_t5.py:217-262 (rel = key position - query position)."""
    out = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        out = out + (rel > 0).long() * num_buckets
        rel = rel.abs()
    else:
        rel = -torch.min(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).long()
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return out + torch.where(is_small, rel, large)


def t5_position_bias(table: Tensor, q_len: int, k_len: int, bidirectional: bool, q_offset: int = 0, max_distance: int = 128) -> Tensor:
    """``compute_bias`` HF:...modeling_t5.py:264-279: ``[1, H, q_len, k_len]`` from the ``[num_buckets, H]`` embedding of the FIRST layer of
    a stack (shared by all its layers); ``q_offset`` = tokens already in the cache."""
    ctx = torch.arange(q_len)[:, None] + q_offset
    mem = torch.arange(k_len)[None, :]
    b = t5_relative_bucket(mem - ctx, bidirectional, table.shape[0], max_distance)
    return table[b].permute(2, 0, 1)[None]


def _t5_attention(sd, pre: str, x: Tensor, kv: Tensor, bias: Tensor, H: int, dkv: int) -> Tensor:
    """``T5Attention.forward`` HF:...modeling_t5.py:281-369: q / k / v / o without bias, scores NOT scaled, ``bias`` (position bias +
    additive mask) added before the softmax."""
    B, S, _ = x.shape
    Sk = kv.shape[1]
    q = (x @ sd[pre + "q.weight"].T).view(B, S, H, dkv).transpose(1, 2)
    k = (kv @ sd[pre + "k.weight"].T).view(B, Sk, H, dkv).transpose(1, 2)
    v = (kv @ sd[pre + "v.weight"].T).view(B, Sk, H, dkv).transpose(1, 2)
    w = torch.softmax(q @ k.transpose(-1, -2) + bias, dim=-1)
    return (w @ v).transpose(1, 2).reshape(B, S, H * dkv) @ sd[pre + "o.weight"].T


def _t5_ffn(sd, pre: str, x: Tensor, cfg: dict) -> Tensor:
    """``T5DenseActDense`` (:75-94, relu) or ``T5DenseGatedActDense`` (:97-123, gelu_new(wi_0 x) * wi_1 x)."""
    if cfg.get("gated", True):
        return (gelu_new(x @ sd[pre + "wi_0.weight"].T) * (x @ sd[pre + "wi_1.weight"].T)) @ sd[pre + "wo.weight"].T
    return torch.relu(x @ sd[pre + "wi.weight"].T) @ sd[pre + "wo.weight"].T


def t5_encoder(sd, cfg: dict, inputs_embeds: Tensor, attention_mask: Optional[Tensor] = None) -> Tensor:
    """``T5Stack`` (encoder) HF:...modeling_t5.py:640-751: blocks of [RMSNorm -> self-attention + residual, RMSNorm -> FFN + residual],
    final RMSNorm; bidirectional relative bias from block 0; padded keys get ``finfo.min`` added."""
    B, S, _ = inputs_embeds.shape
    H, dkv, eps = cfg["n_head"], cfg["d_kv"], cfg.get("eps", 1e-6)
    if attention_mask is None:
        attention_mask = torch.ones(B, S, dtype=torch.long)
    bias = t5_position_bias(sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], S, S, True,
                            max_distance=cfg.get("max_distance", 128))
    bias = bias + torch.where(attention_mask[:, None, None, :] != 0, 0.0, NEG)
    h = inputs_embeds
    for i in range(cfg["n_layer"]):
        p = f"encoder.block.{i}.layer."
        h = h + _t5_attention(sd, p + "0.SelfAttention.", t5_rms_norm(h, sd[p + "0.layer_norm.weight"], eps), t5_rms_norm(h, sd[p + "0.layer_norm.weight"], eps), bias, H, dkv)
        h = h + _t5_ffn(sd, p + "1.DenseReluDense.", t5_rms_norm(h, sd[p + "1.layer_norm.weight"], eps), cfg)
    return t5_rms_norm(h, sd["encoder.final_layer_norm.weight"], eps)


def t5_decoder(sd, cfg: dict, dec_embeds: Tensor, enc_out: Tensor, enc_mask: Optional[Tensor] = None, dec_mask: Optional[Tensor] = None) -> Tensor:
    """``T5Stack`` (decoder): causal self-attention with the unidirectional relative bias of block 0, cross-attention over the encoder
    output (zero position bias, encoder padding masked), FFN; final RMSNorm.  ``dec_mask`` [B, T] (HF ``decoder_attention_mask``, a padded
    decoder prompt): 0 = a key the self-attention does not see; positions stay absolute."""
    B, T, _ = dec_embeds.shape
    S = enc_out.shape[1]
    H, dkv, eps = cfg["n_head"], cfg["d_kv"], cfg.get("eps", 1e-6)
    n_dec = cfg.get("n_dec_layer", cfg["n_layer"])
    if enc_mask is None:
        enc_mask = torch.ones(B, S, dtype=torch.long)
    self_bias = t5_position_bias(sd["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], T, T, False,
                                 max_distance=cfg.get("max_distance", 128))
    self_bias = self_bias + torch.where(torch.ones(T, T, dtype=torch.bool).tril()[None, None], 0.0, NEG)
    if dec_mask is not None:
        self_bias = self_bias + torch.where(dec_mask[:, None, None, :] != 0, 0.0, NEG)
    cross_bias = torch.where(enc_mask[:, None, None, :] != 0, 0.0, NEG).expand(B, 1, T, S)
    h = dec_embeds
    for i in range(n_dec):
        p = f"decoder.block.{i}.layer."
        a = t5_rms_norm(h, sd[p + "0.layer_norm.weight"], eps)
        h = h + _t5_attention(sd, p + "0.SelfAttention.", a, a, self_bias, H, dkv)
        h = h + _t5_attention(sd, p + "1.EncDecAttention.", t5_rms_norm(h, sd[p + "1.layer_norm.weight"], eps), enc_out, cross_bias, H, dkv)
        h = h + _t5_ffn(sd, p + "2.DenseReluDense.", t5_rms_norm(h, sd[p + "2.layer_norm.weight"], eps), cfg)
    return t5_rms_norm(h, sd["decoder.final_layer_norm.weight"], eps)


def t5_lm_logits(sd, cfg: dict, dec_hidden: Tensor) -> Tensor:
    """``T5ForConditionalGeneration.forward`` HF:...modeling_t5.py:1040-1048: tied embeddings rescale the decoder output by d_model^-0.5
    and use ``shared`` as the head (T5 v1.0); v1.1 / T0 have their own ``lm_head``."""
    if cfg.get("tied", False):
        return (dec_hidden * dec_hidden.shape[-1] ** -0.5) @ sd["shared.weight"].T
    return dec_hidden @ sd["lm_head.weight"].T


def t5_shift_right(labels: Tensor, start_id: int = 0, pad_id: int = 0) -> Tensor:
    """``_shift_right`` HF:...modeling_t5.py:618-637: ``[start, labels[:-1]]`` with -100 -> pad."""
    out = torch.full_like(labels, start_id)
    out[:, 1:] = labels[:, :-1]
    return out.masked_fill(out == -100, pad_id)


def vct0_forward(sd, cfg: dict, mapper, mcfg: dict, prefix: Tensor, labels: Tensor):
    """``VCT0Model.forward`` src/models/vct0.py:380-394: the encoder sees ONLY the projected prefix (``inputs_embeds``), the decoder is
    teacher-forced on ``labels``; loss = CE over labels != -100 (HF:...modeling_t5.py:1050-1056).  Returns ``(loss, logits [B, T, V])``."""
    E = sd["shared.weight"].shape[1]
    pre = mapper_project(prefix, mapper, mcfg.get("mapping_type", "mlp"), mcfg["prefix_length"], E, mcfg.get("clip_length"), mcfg.get("num_layers", 8))
    enc = t5_encoder(sd, cfg, pre)
    dec_in = sd["shared.weight"][t5_shift_right(labels, cfg.get("decoder_start_token_id", 0), cfg.get("pad_token_id", 0))]
    logits = t5_lm_logits(sd, cfg, t5_decoder(sd, cfg, dec_in, enc))
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=-100)
    return loss, logits


def _t5_greedy(sd, cfg: dict, enc: Tensor, enc_mask: Optional[Tensor], max_length: int, dec_prompt: Optional[Tensor] = None,
               dec_mask: Optional[Tensor] = None):
    """HF greedy search for an encoder-decoder (``GenerationMixin._sample`` with do_sample=False): start token = decoder_start_token_id
    (= pad = 0 for T5), every step re-runs the decoder over the whole prefix (same logits as the cached run), rows that produced eos (1)
    emit pad afterwards, stop when all rows are finished or ``max_length`` TOTAL decoder positions exist.  A decoder prompt
    (``_prepare_decoder_input_ids_for_generation``): the start token is prepended - and the mask extended by a column of ones - only when NO
    row of the prompt begins with it; generated positions are attended (mask 1).  Returns ``(sequences [B, <= max_length], per-step logits list)``."""
    B = enc.shape[0]
    start, pad, eos = cfg.get("decoder_start_token_id", 0), cfg.get("pad_token_id", 0), cfg.get("eos_token_id", 1)
    seq = torch.full((B, 1), start, dtype=torch.long)
    dmask = None
    if dec_prompt is not None:
        dmask = dec_mask.clone() if dec_mask is not None else torch.ones_like(dec_prompt)
        if bool((dec_prompt[:, 0] != start).all()):
            dec_prompt, dmask = torch.cat([seq, dec_prompt], dim=1), torch.cat([torch.ones_like(dmask[:, :1]), dmask], dim=1)
        seq = dec_prompt.clone()
    unfinished = torch.ones(B, dtype=torch.long)
    scores = []
    while seq.shape[1] < max_length:
        logits = t5_lm_logits(sd, cfg, t5_decoder(sd, cfg, sd["shared.weight"][seq], enc, enc_mask, dmask))[:, -1]
        scores.append(logits)
        nxt = logits.argmax(-1) * unfinished + pad * (1 - unfinished)
        seq = torch.cat([seq, nxt[:, None]], dim=1)
        if dmask is not None:
            dmask = torch.cat([dmask, torch.ones_like(dmask[:, :1])], dim=1)
        unfinished = unfinished * (nxt != eos).long()
        if unfinished.max() == 0:
            break
    return seq, scores


def vct0_generate(sd, cfg: dict, mapper, mcfg: dict, prefix: Tensor, question_tokens: Optional[Tensor] = None,
                  question_mask: Optional[Tensor] = None, num_shots: Optional[int] = None, max_length: int = 20,
                  special_token_id: int = 32099, one_at_a_time: bool = False, no_prefix: bool = False,
                  decoder_input_ids: Optional[Tensor] = None, decoder_attention_mask: Optional[Tensor] = None):
    """``VCT0Model.generate`` src/models/vct0.py:396-491 (greedy, HF defaults):
      * ``question_tokens is None`` (:485-491, the CC executor): the encoder sees the projected prefix only;
      * few-shot (:452-466): sentinel tokens ``special_token_id - i`` of the prompt are replaced by the L prefix vectors of image i
        (``insert_prefix_into_input`` :494-533), one encoder pass over the joint sequence;
      * ``one_at_a_time`` (:426-442): question_tokens ``[B, n, T]``; example i is encoded by itself with sentinel ``special - i``, the
        encoder outputs and masks are concatenated along the sequence before decoding;
      * ``no_prefix`` (:409-424, text only): plain T5 generate on the tokens;
      * ``decoder_input_ids`` (:468-480): the encoder sees the LAST image's prefix at the one sentinel of each row, the decoder continues the
        prompt; the sequences are cut by the GIVEN prompt length, as the reference does.
    Returns ``(sequences, scores)``."""
    E, L = sd["shared.weight"].shape[1], mcfg["prefix_length"]
    proj = lambda p: mapper_project(p, mapper, mcfg.get("mapping_type", "mlp"), L, E, mcfg.get("clip_length"), mcfg.get("num_layers", 8))
    if no_prefix:
        enc = t5_encoder(sd, cfg, sd["shared.weight"][question_tokens], question_mask)
        return _t5_greedy(sd, cfg, enc, question_mask, max_length)
    if question_tokens is None:
        enc = t5_encoder(sd, cfg, proj(prefix))
        return _t5_greedy(sd, cfg, enc, None, max_length)
    B = question_tokens.shape[0]
    if one_at_a_time:
        n = prefix.shape[1]
        pp = proj(prefix.reshape(-1, prefix.shape[-1])).view(B, n, L, E)
        encs, masks = [], []
        for i in range(n):
            emb, msk = insert_prefix_into_input(L, 0, question_tokens[:, i], sd["shared.weight"][question_tokens[:, i]], pp[:, i], question_mask[:, i],
                                                special_token_id - i)
            encs.append(t5_encoder(sd, cfg, emb, msk))
            masks.append(msk)
        return _t5_greedy(sd, cfg, torch.cat(encs, 1), torch.cat(masks, 1), max_length)
    n_img = prefix.shape[1] if prefix.dim() >= 3 else 1
    ns = (n_img - 1) if not num_shots else num_shots
    pp = proj(prefix.reshape(-1, prefix.shape[-1])).view(B, -1, L, E)
    if decoder_input_ids is not None:
        emb, msk = insert_prefix_into_input(L, 0, question_tokens, sd["shared.weight"][question_tokens], pp[:, -1], question_mask, special_token_id)
        seq, scores = _t5_greedy(sd, cfg, t5_encoder(sd, cfg, emb, msk), msk, max_length, decoder_input_ids, decoder_attention_mask)
        return seq[:, decoder_input_ids.shape[1]:], scores
    emb, msk = insert_prefix_into_input(L, ns, question_tokens, sd["shared.weight"][question_tokens], pp, question_mask, special_token_id)
    return _t5_greedy(sd, cfg, t5_encoder(sd, cfg, emb, msk), msk, max_length)
