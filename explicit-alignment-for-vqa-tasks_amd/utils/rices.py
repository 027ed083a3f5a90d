"""RICES: retrieval of in-context examples by CLIP-embedding similarity, on the GPU.

Mirror of src/in_context_example_selection/get_question_knn.py:64-76 - ``faiss.normalize_L2`` on the
database (train question embeddings) and on the queries (val question embeddings), ``IndexFlatIP`` search with
``k = 2048`` - as three HIP steps per tile of queries: exact-fp32 MFMA GEMM for the inner products, then a
radix-select top-k (``eavqa_topk_rows``).  Returns faiss's ``(D, I)``: similarities sorted descending and int64 row
numbers into the database.

faiss is not installed here and the reference ships no retrieval outputs, so index-for-index parity with faiss is
*unpinned*: scores are exact fp32 sums in a different order than faiss's BLAS call, and among exactly equal scores this
build returns the smaller database row first (faiss leaves that order unspecified).  tests/test_retrieval_gpu.py pins the
kernels against float64 numpy instead.
"""
from __future__ import annotations

from typing import Tuple

import torch

from .. import ops

Tensor = torch.Tensor


def knn_inner_product(database: Tensor, queries: Tensor, k: int = 2048, normalize: bool = True,
                      query_tile: int = 1024) -> Tuple[Tensor, Tensor]:
    """``database`` [Nd, D], ``queries`` [Nq, D] float32 on the GPU (the reference stacks ``[1, D]`` pickled entries and
    squeezes, get_question_knn.py:45,60).  ``normalize=True`` works on copies.  Peak extra memory: one
    ``[query_tile, Nd]`` float32 score tile (1.8 GB at Nd = 443 k)."""
    if database.dim() != 2 or queries.dim() != 2 or database.shape[1] != queries.shape[1]:
        raise ValueError("knn_inner_product: database [Nd, D] and queries [Nq, D] expected")
    if not 0 < k <= min(2048, database.shape[0]):
        raise ValueError("knn_inner_product: 1 <= k <= min(2048, Nd)")
    db = database.to(torch.float32).contiguous()
    q = queries.to(torch.float32).contiguous()
    if normalize:
        db = ops.l2_normalize_rows_(db.clone() if db.data_ptr() == database.data_ptr() else db)
        q = ops.l2_normalize_rows_(q.clone() if q.data_ptr() == queries.data_ptr() else q)
    Nq = q.shape[0]
    D = torch.empty((Nq, k), device=q.device, dtype=torch.float32)
    I = torch.empty((Nq, k), device=q.device, dtype=torch.int64)
    for s in range(0, Nq, query_tile):
        e = min(Nq, s + query_tile)
        scores = ops.gemm(q[s:e], db, out_f32=True)
        D[s:e], I[s:e] = ops.topk_rows(scores, k)
    return D, I


def rices_neighbours(train_embeddings: Tensor, val_embeddings: Tensor, k: int = 2048) -> Tuple[Tensor, Tensor]:
    """The two arrays the reference saves (``text_nearest_neighbours_similarities_2048.npy`` = D,
    ``text_nearest_neighbours_2048.npy`` = I, get_question_knn.py:78-81)."""
    return knn_inner_product(train_embeddings, val_embeddings, k=k, normalize=True)
