"""Synthetic Conceptual-Captions-shaped and VQA-shaped batches (no dataset is reachable offline).

Shapes and keys follow the reference's collate functions:
  * CC: ``clip_embeddings [B,D]``, ``labels [B,T]`` (pad -> -100), ``labels_attention_mask``
    (src/data_loader_manager/data_loader_conceptual_captions.py:78-104); caption lengths U{8..32},
    ids uniform excluding pad/eos, right-padded to the longest with eos-as-pad;
  * in this build the CLIP encode runs in the loop, so a batch also carries ``pixel_values``
    ``[B,3,H,W]`` ~ N(0,1) ("already pre-processed" images).
Seed 2021 is the reference's seed (configs/vqa2/clip_cap.jsonnet:17).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch


def cc_batch(batch_size: int, vocab: int, pad_token_id: int, image_size: int = 224, min_len: int = 8, max_len: int = 32,
             seed: int = 2021, device="cpu", with_pixels: bool = True, embed_dim: Optional[int] = None) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(min_len, max_len + 1, (batch_size,), generator=g)
    lens[0] = max_len                                   # "longest" padding -> a fixed T for graph capture
    T = int(lens.max())
    ids = torch.randint(0, vocab - 2, (batch_size, T), generator=g)
    ids = torch.where(ids >= pad_token_id, ids + 1, ids) if pad_token_id < vocab - 1 else ids   # never draw the pad/eos id
    mask = (torch.arange(T)[None] < lens[:, None]).long()
    input_ids = ids * mask + pad_token_id * (1 - mask)
    labels = input_ids.clone()
    labels[labels == pad_token_id] = -100               # data_loader_conceptual_captions.py:94-95
    batch = dict(input_ids=input_ids, attention_mask=mask, labels=labels, labels_attention_mask=mask.clone())
    if with_pixels:
        batch["pixel_values"] = torch.randn(batch_size, 3, image_size, image_size, generator=g)
    if embed_dim is not None:
        batch["clip_embeddings"] = torch.randn(batch_size, embed_dim, generator=g)
    return {k: v.to(device) for k, v in batch.items()}


def fewshot_batch(batch_size: int, vocab: int, n_shots: int, seg_len: int, sentinel_top: int, image_size: int = 224,
                  seed: int = 2021, device="cpu") -> Dict[str, torch.Tensor]:
    """``n_shots`` in-context examples + the query: n_shots+1 images per question, one sentinel token
    (ids ``sentinel_top - i``) per image followed by ``seg_len`` text tokens (cfg 4)."""
    g = torch.Generator().manual_seed(seed)
    n_img = n_shots + 1
    T = n_img * (1 + seg_len)
    ids = torch.randint(0, min(vocab, sentinel_top - n_img) - 2, (batch_size, T), generator=g)
    for i in range(n_img):
        ids[:, i * (1 + seg_len)] = sentinel_top - i
    mask = torch.ones(batch_size, T, dtype=torch.long)
    px = torch.randn(batch_size, n_img, 3, image_size, image_size, generator=g)
    return dict(input_ids=ids.to(device), attention_mask=mask.to(device), pixel_values=px.to(device))
