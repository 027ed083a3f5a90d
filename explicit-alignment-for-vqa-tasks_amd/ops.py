"""Tensor-level wrappers over the C ABI (include/eavqa.h).

PyTorch supplies device memory and the current HIP stream; every arithmetic op on the hot path
is one of the calls below.  Nothing here computes with torch ops on the data path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT, BF16, F32, call

Tensor = torch.Tensor


F16 = 2   # EAVQA_F16: storage type of a frozen tower's residual stream only (layernorm_fwd x / y, gemm residual / out)


def dtype_id(dt: torch.dtype, stream: bool = False) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if stream and dt == torch.float16:
        return F16
    raise _lib.EavqaError(f"unsupported storage dtype {dt}")


def _stream() -> int:
    # raw HIP stream handle of torch's current stream (the fast private accessor: the public
    # torch.cuda.current_stream() costs several microseconds per call, which adds up over ~1300 launches a step)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class KernelSelect:
    """Kernel selectors handed to the ``*_ex`` entry points of include/eavqa_test.h.  Both are 0 in the product path (the
    library chooses by shape); the parity tests and tools/gemm_bench.py set them to cover / time a particular kernel.  The
    state lives HERE, in the test-facing Python layer - libeavqa_hip.so itself holds no mutable state."""
    gemm = 0
    attention = 0
    decode_route = 0      # eavqa_lm_block_forward_ex: 0 = the shipped decode-step structure, other values = A / B alternatives


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev(t: Tensor) -> None:
    if not t.is_cuda:
        raise _lib.EavqaError("eavqa ops run on the GPU only (no CPU fallback): got a CPU tensor")


def _ld(t: Tensor) -> int:
    """Leading dimension of a 2-D (or flattened row-major) tensor whose last dim is contiguous."""
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        raise _lib.EavqaError("last dimension must be contiguous")
    return t.stride(-2) if t.dim() >= 2 else t.shape[-1]


def gemm(a: Tensor, b: Tensor, *, a_kc: bool = True, b_kc: bool = True, bias: Optional[Tensor] = None,
         act: str = "none", aux_in: Optional[Tensor] = None, aux_out: Optional[Tensor] = None,
         residual: Optional[Tensor] = None, out: Optional[Tensor] = None, out_f32: bool = False,
         alpha: float = 1.0, copy_out: Optional[Tensor] = None, stats_out: Optional[Tensor] = None, ln_stats: Optional[Tensor] = None,
         ln_c: Optional[Tensor] = None, ln_eps: float = 1e-5, ln_save: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """``C = epilogue(alpha * A @ B^T)`` - see eavqa_gemm.  ``a``: [M,K] (a_kc) or [K,M];
    ``b``: [N,K] (b_kc, nn.Linear layout) or [K,N] (Conv1D layout).

    eavqa_gemm_ln (a frozen pre-LN layer's LayerNorm folded into its neighbours), producer side: ``copy_out`` [M, N] in the operand
    dtype, ``stats_out`` float32 [M, >= ceil(N / 64), 2]; consumer side: ``ln_stats`` (a producer's ``stats_out`` for the rows of
    ``a``), ``ln_c`` float32 [N], ``ln_eps``, ``ln_save`` = (mean, rstd) float32 [M] to be written."""
    _dev(a)
    M, K = (a.shape if a_kc else (a.shape[1], a.shape[0]))
    N, Kb = (b.shape if b_kc else (b.shape[1], b.shape[0]))
    if K != Kb:
        raise _lib.EavqaError(f"gemm inner dims differ: {K} vs {Kb}")
    if a.dtype != b.dtype:
        raise _lib.EavqaError("gemm operands must share a dtype")
    dt = dtype_id(a.dtype)
    half = torch.float16                                    # 16-bit stream of a frozen tower (bf16 operands): EAVQA_GEMM_STREAM_F16
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    elif out.dtype != torch.float32 and out.dtype != a.dtype and not (out.dtype == half and a.dtype == torch.bfloat16):
        raise _lib.EavqaError("gemm out dtype must be float32, the operand dtype or (bf16 operands) float16")
    out_f32 = out.dtype == torch.float32
    aux = aux_in if aux_in is not None else aux_out
    flags = int(out_f32)                                    # EAVQA_GEMM_OUT_F32
    if residual is not None and residual.dtype != torch.float32:
        if residual.dtype != a.dtype and not (residual.dtype == half and a.dtype == torch.bfloat16):
            raise _lib.EavqaError("gemm residual must be float32, the operand dtype or (bf16 operands) float16")
        flags |= 2                                          # EAVQA_GEMM_RESIDUAL_LOWP: the residual stream of a frozen tower in 16 bits
    stream_half = (out.dtype == half) or (residual is not None and residual.dtype == half)
    if stream_half:
        if (out.dtype not in (half, torch.float32)) or (residual is not None and residual.dtype not in (half, torch.float32)):
            raise _lib.EavqaError("gemm: a float16 stream needs float16 (or float32) for both the residual and the output")
        flags |= 4                                          # EAVQA_GEMM_STREAM_F16
    args = (dt, int(a_kc), int(b_kc), M, N, K, _p(a), _ld(a), _p(b), _ld(b), _p(out), _ld(out),
            flags, float(alpha), _p(bias), ACT[act], _p(aux_in), _p(aux_out), _ld(aux) if aux is not None else 0,
            _p(residual), _ld(residual) if residual is not None else 0, _stream())
    if copy_out is not None or stats_out is not None or ln_stats is not None:
        g = _lib.GemmLn()
        if copy_out is not None:
            if copy_out.dtype != a.dtype or tuple(copy_out.shape) != (M, N):
                raise _lib.EavqaError("gemm copy_out must be [M, N] in the operand dtype")
            g.copy_out, g.ld_copy = _p(copy_out), _ld(copy_out)
        if stats_out is not None:
            if stats_out.dtype != torch.float32 or stats_out.dim() != 3 or stats_out.shape[0] != M or stats_out.shape[2] != 2 or not stats_out.is_contiguous():
                raise _lib.EavqaError("gemm stats_out must be a contiguous float32 [M, slots, 2]")
            g.stats_out, g.stats_ld = _p(stats_out), stats_out.shape[1]
        if ln_stats is not None:
            if ln_stats.dtype != torch.float32 or ln_stats.dim() != 3 or ln_stats.shape[0] != M or ln_stats.shape[2] != 2 or not ln_stats.is_contiguous():
                raise _lib.EavqaError("gemm ln_stats must be a contiguous float32 [M, slots, 2]")
            if ln_c is None or ln_c.dtype != torch.float32 or ln_c.numel() != N:
                raise _lib.EavqaError("gemm ln_c must be float32 [N]")
            g.ln_stats, g.ln_parts, g.ln_ld, g.ln_cols = _p(ln_stats), ln_stats.shape[1], ln_stats.shape[1], K
            g.ln_c, g.ln_eps = _p(ln_c), float(ln_eps)
            if ln_save is not None:
                g.mean_out, g.rstd_out = _p(ln_save[0]), _p(ln_save[1])
        if KernelSelect.gemm:
            call("eavqa_gemm_ln_ex", *args[:-1], C.byref(g), args[-1], KernelSelect.gemm)
        else:
            call("eavqa_gemm_ln", *args[:-1], C.byref(g), args[-1])
    elif KernelSelect.gemm:
        call("eavqa_gemm_ex", *args, KernelSelect.gemm)
    else:
        call("eavqa_gemm", *args)
    return out


def quantize_rows_fp8(x: Tensor):
    """Rows of ``x`` (bf16 / fp32 [rows, cols]) -> (e4m3 bytes as uint8 [rows, cols], float32 row scales [rows])."""
    _dev(x)
    rows, cols = x.shape
    q = torch.empty((rows, cols), device=x.device, dtype=torch.uint8)
    sc = torch.empty(rows, device=x.device, dtype=torch.float32)
    call("eavqa_quantize_rows_fp8", dtype_id(x.dtype), rows, cols, _p(x), _ld(x), _p(q), _ld(q), _p(sc), _stream())
    return q, sc


def gemm_fp8(a_q: Tensor, a_scale: Tensor, b_q: Tensor, b_scale: float, *, bias: Optional[Tensor] = None, act: str = "none",
             aux_in: Optional[Tensor] = None, aux_out: Optional[Tensor] = None, residual: Optional[Tensor] = None,
             out: Optional[Tensor] = None, out_f32: bool = False, alpha: float = 1.0, tile: int = 0) -> Tensor:
    """``C = epilogue(alpha * b_scale * a_scale[m] * A_q @ B_q^T)`` on e4m3 operands (uint8 storage) - see eavqa_gemm_fp8.
    Outputs / aux are bfloat16 (or float32 ``out``)."""
    _dev(a_q)
    M, K = a_q.shape
    N = b_q.shape[0]
    if b_q.shape[1] != K:
        raise _lib.EavqaError(f"gemm_fp8 inner dims differ: {K} vs {b_q.shape[1]}")
    if out is None:
        out = torch.empty((M, N), device=a_q.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    aux = aux_in if aux_in is not None else aux_out
    call("eavqa_gemm_fp8", M, N, K, _p(a_q), _ld(a_q), _p(a_scale), _p(b_q), _ld(b_q), float(b_scale), _p(out), _ld(out),
         int(out.dtype == torch.float32), float(alpha), _p(bias), ACT[act], _p(aux_in), _p(aux_out), _ld(aux) if aux is not None else 0,
         _p(residual), _ld(residual) if residual is not None else 0, _stream(), int(tile))
    return out


def layernorm_fwd(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float, out_dtype: torch.dtype,
                  save_stats: bool = False, out: Optional[Tensor] = None):
    """Rows of ``x`` ([rows, cols]: float32, bfloat16 or float16) -> ``y`` in ``out_dtype`` (+ mean, rstd).  float16 is the storage
    type of a frozen tower's residual stream only (models/clip_vit.py)."""
    _dev(x)
    rows, cols = x.shape
    y = out if out is not None else torch.empty((rows, cols), device=x.device, dtype=out_dtype)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    x_kind = {torch.float32: 1, torch.bfloat16: 2, torch.float16: 3}.get(x.dtype)
    if x_kind is None:
        raise _lib.EavqaError(f"unsupported layernorm input dtype {x.dtype}")
    call("eavqa_layernorm_fwd", dtype_id(out_dtype, stream=True), x_kind, rows, cols, _p(x), _ld(x),
         _p(gamma), _p(beta), float(eps), _p(y), _ld(y), _p(mean), _p(rstd), _stream())
    return (y, mean, rstd) if save_stats else y


def layernorm_fwd_fp8(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float, save_stats: bool = False):
    """LayerNorm whose bf16 result leaves as e4m3 rows + row scales (``layernorm_fwd`` followed by ``quantize_rows_fp8``, one kernel):
    ``(q uint8 [rows, cols], scale float32 [rows])`` (+ mean, rstd)."""
    _dev(x)
    rows, cols = x.shape
    q = torch.empty((rows, cols), device=x.device, dtype=torch.uint8)
    sc = torch.empty(rows, device=x.device, dtype=torch.float32)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    x_kind = {torch.float32: 1, torch.bfloat16: 2, torch.float16: 3}.get(x.dtype)
    if x_kind is None:
        raise _lib.EavqaError(f"unsupported layernorm input dtype {x.dtype}")
    call("eavqa_layernorm_fwd_fp8", x_kind, rows, cols, _p(x), _ld(x), _p(gamma), _p(beta), float(eps), _p(q), _ld(q), _p(sc), _p(mean), _p(rstd),
         _stream())
    return (q, sc, mean, rstd) if save_stats else (q, sc)


def layernorm_bwd_fp8(x: Tensor, dy: Tensor, gamma: Optional[Tensor], mean: Tensor, rstd: Tensor, q_out: Tensor, scale_out: Tensor,
                      dres: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """``layernorm_bwd`` whose bf16 copy of dx leaves as e4m3 rows + row scales into ``q_out`` (uint8 [rows, cols]) / ``scale_out``."""
    _dev(x)
    rows, cols = x.shape
    if dy.dtype != torch.bfloat16:
        raise _lib.EavqaError("layernorm_bwd_fp8: dy must be bfloat16")
    dx = out if out is not None else torch.empty((rows, cols), device=x.device, dtype=torch.float32)
    if dres is not None and _ld(dres) != _ld(dx):
        raise _lib.EavqaError("dres must share dx's leading dimension")
    call("eavqa_layernorm_bwd_fp8", int(x.dtype == torch.float32), rows, cols, _p(x), _ld(x), _p(dy), _ld(dy), _p(gamma), _p(mean), _p(rstd),
         _p(dres), _p(dx), _ld(dx), _p(q_out), _ld(q_out), _p(scale_out), _stream())
    return dx


def layernorm_bwd(x: Tensor, dy: Tensor, gamma: Optional[Tensor], mean: Tensor, rstd: Tensor,
                  dres: Optional[Tensor] = None, dgamma: Optional[Tensor] = None, dbeta: Optional[Tensor] = None,
                  out: Optional[Tensor] = None, lowp_out: Optional[Tensor] = None) -> Tensor:
    """``lowp_out``: optional [rows, cols] tensor in ``dy.dtype`` that receives a second copy of dx."""
    _dev(x)
    rows, cols = x.shape
    dx = out if out is not None else torch.empty((rows, cols), device=x.device, dtype=torch.float32)
    if dres is not None and _ld(dres) != _ld(dx):
        raise _lib.EavqaError("dres must share dx's leading dimension")
    call("eavqa_layernorm_bwd", dtype_id(dy.dtype), int(x.dtype == torch.float32), rows, cols, _p(x), _ld(x), _p(dy), _ld(dy),
         _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _ld(dx), _p(dgamma), _p(dbeta), _p(lowp_out),
         _ld(lowp_out) if lowp_out is not None else 0, _stream())
    return dx


def attention_fwd(q: Tensor, k: Tensor, v: Tensor, B: int, H: int, Sq: int, Sk: int, hd: int, *,
                  key_mask: Optional[Tensor] = None, causal: bool = False, scale: float = 1.0, save_lse: bool = False,
                  q_batch_rows: int = 0, kv_batch_rows: int = 0, ld_mask: int = 0, out: Optional[Tensor] = None,
                  cu_seqlens: Optional[Tensor] = None):
    """``q``: rows [B*Sq, >=H*hd] views (element (b,s,h,d) at row b*Sq+s, col h*hd+d); same for k/v.
    With ``cu_seqlens`` the rows are packed (sample b = rows cu[b]..cu[b+1]) and ``Sq`` is the longest sample."""
    _dev(q)
    o = out if out is not None else torch.empty((q.shape[0] if cu_seqlens is not None else B * Sq, H * hd), device=q.device, dtype=q.dtype)
    lse = torch.empty((B, H, Sq), device=q.device, dtype=torch.float32) if save_lse else None
    args = (dtype_id(q.dtype), B, H, Sq, Sk, hd, _p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(o), _ld(o),
            q_batch_rows, kv_batch_rows, _p(key_mask), ld_mask, _p(cu_seqlens), int(causal), float(scale), _p(lse), _stream())
    if KernelSelect.attention:
        call("eavqa_attention_fwd_ex", *args, KernelSelect.attention)
    else:
        call("eavqa_attention_fwd", *args)
    return (o, lse) if save_lse else o


def attention_decode(q: Tensor, k_cache: Tensor, v_cache: Tensor, k_new: Tensor, v_new: Tensor, B: int, H: int, Sk: int, hd: int, *,
                     kv_batch_rows: int, key_mask: Optional[Tensor] = None, ld_mask: int = 0, scale: float = 1.0,
                     out: Optional[Tensor] = None) -> Tensor:
    """One decode step that appends: ``k_new`` / ``v_new`` (row views [B, H*hd]) go to position ``Sk - 1`` of every sample's cache
    (``k_cache`` / ``v_cache`` row views [B*kv_batch_rows, H*hd]) and ``q`` [B, H*hd] attends over positions ``0 .. Sk-1``."""
    _dev(q)
    o = out if out is not None else torch.empty((B, H * hd), device=q.device, dtype=q.dtype)
    call("eavqa_attention_decode", dtype_id(q.dtype), B, H, Sk, hd, _p(q), _ld(q), _p(k_cache), _ld(k_cache), _p(v_cache), _ld(v_cache),
         kv_batch_rows, _p(k_new), _p(v_new), _ld(k_new), _p(o), _ld(o), _p(key_mask), ld_mask, float(scale), _stream())
    return o


def attention_decode_splitk(part: Tensor, bias: Optional[Tensor], k_cache: Tensor, v_cache: Tensor, B: int, H: int, Sk: int, hd: int, *,
                            kv_batch_rows: int, key_mask: Optional[Tensor] = None, ld_mask: int = 0, scale: float = 1.0,
                            out: Optional[Tensor] = None) -> Tensor:
    """``attention_decode`` straight from the QKV projection's split-K partial sums ``part`` [ks, B, 3*H*hd] (+ ``bias`` [3*H*hd])."""
    _dev(part)
    o = out if out is not None else torch.empty((B, H * hd), device=part.device, dtype=k_cache.dtype)
    call("eavqa_attention_decode_splitk", dtype_id(k_cache.dtype), B, H, Sk, hd, _p(part), part.shape[0], _p(bias), _p(k_cache), _ld(k_cache),
         _p(v_cache), _ld(v_cache), kv_batch_rows, _p(o), _ld(o), _p(key_mask), ld_mask, float(scale), _stream())
    return o


def attention_bwd(q, k, v, o, d_o, lse, B, H, Sq, Sk, hd, *, key_mask=None, causal=False, scale=1.0,
                  dq=None, dk=None, dv=None, cu_seqlens=None):
    _dev(q)
    nq = q.shape[0] if cu_seqlens is not None else B * Sq
    nk = k.shape[0] if cu_seqlens is not None else B * Sk
    dq = dq if dq is not None else torch.empty((nq, H * hd), device=q.device, dtype=q.dtype)
    dk = dk if dk is not None else torch.empty((nk, H * hd), device=q.device, dtype=q.dtype)
    dv = dv if dv is not None else torch.empty((nk, H * hd), device=q.device, dtype=q.dtype)
    delta = torch.empty((B, H, Sq), device=q.device, dtype=torch.float32)
    args = (dtype_id(q.dtype), B, H, Sq, Sk, hd, _p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(o), _ld(o),
            _p(d_o), _ld(d_o), _p(dq), _ld(dq), _p(dk), _ld(dk), _p(dv), _ld(dv), _p(key_mask), _p(cu_seqlens), int(causal),
            float(scale), _p(lse), _p(delta), _stream())
    if KernelSelect.attention:
        call("eavqa_attention_bwd_ex", *args, KernelSelect.attention)
    else:
        call("eavqa_attention_bwd", *args)
    return dq, dk, dv


def build_prefix_rows(tokens: Tensor, question_mask: Tensor, L: int, pos_mode: int, prefix_row_stride: Optional[int] = None,
                      prefix_row_offset: int = 0):
    """int64 [B,T] tokens/mask -> (src, mask, pos) int32 [B, L+T]."""
    _dev(tokens)
    B, T = tokens.shape
    tokens = tokens.contiguous()
    qm = question_mask.to(torch.int64).contiguous()
    S = L + T
    src = torch.empty((B, S), device=tokens.device, dtype=torch.int32)
    msk = torch.empty_like(src)
    pos = torch.empty_like(src)
    call("eavqa_build_prefix_rows", B, L, T, _p(tokens), _p(qm), pos_mode, L if prefix_row_stride is None else prefix_row_stride,
         prefix_row_offset, _p(src), _p(msk), _p(pos), _stream())
    return src, msk, pos


def build_fewshot_rows(tokens: Tensor, question_mask: Tensor, L: int, n_img: int, special_token_id: int, pos_mode: int):
    _dev(tokens)
    B, T = tokens.shape
    tokens = tokens.contiguous()
    qm = question_mask.to(torch.int64).contiguous()
    T_out = T + (L - 1) * n_img
    src = torch.empty((B, T_out), device=tokens.device, dtype=torch.int32)
    msk = torch.empty_like(src)
    pos = torch.empty_like(src)
    status = torch.empty(B, device=tokens.device, dtype=torch.int32)
    call("eavqa_build_fewshot_rows", B, T, L, n_img, int(special_token_id), _p(tokens), _p(qm), pos_mode, _p(src), _p(msk),
         _p(pos), _p(status), _stream())
    return src, msk, pos, status


def embed_assemble(src: Tensor, pos: Optional[Tensor], wte: Tensor, prefix_rows: Optional[Tensor], wpe: Optional[Tensor],
                   out: Optional[Tensor] = None) -> Tensor:
    _dev(src)
    rows = src.numel()
    E = wte.shape[1]
    x = out if out is not None else torch.empty((rows, E), device=src.device, dtype=torch.float32)
    call("eavqa_embed_assemble", dtype_id(wte.dtype), rows, E, _p(src), _p(pos), _p(wte), _ld(wte), _p(prefix_rows),
         _ld(prefix_rows) if prefix_rows is not None else 0, _p(wpe), _ld(wpe) if wpe is not None else 0, _p(x), _ld(x), _stream())
    return x


def embed_assemble_bwd(src: Tensor, dx: Tensor, n_prefix_rows: int, dtype: torch.dtype) -> Tensor:
    rows, E = dx.shape
    d = torch.zeros((n_prefix_rows, E), device=dx.device, dtype=dtype)
    call("eavqa_embed_assemble_bwd", dtype_id(dtype), rows, E, _p(src), _p(dx), _ld(dx), _p(d), _ld(d), _stream())
    return d


def build_labels(input_ids: Tensor, L: int, pad_token_id: int, bos_token_id: int = -1, mode: int = 0) -> Tensor:
    _dev(input_ids)
    B, T = input_ids.shape
    ids = input_ids.contiguous()
    out = torch.empty((B, L + T), device=ids.device, dtype=torch.int64)
    call("eavqa_build_labels", mode, B, T, L, _p(ids), int(pad_token_id), int(bos_token_id), _p(out), _stream())
    return out


def build_row_plan(mask: Tensor, labels: Optional[Tensor], src: Tensor, pos: Tensor, pack: bool):
    """-> (cu_seqlens [B+1], src_rows, pos_rows, row_labels, flat_index), the row buffers sized B*S (first
    ``cu_seqlens[-1]`` entries valid)."""
    _dev(mask)
    B, S = mask.shape
    dev = mask.device
    cu = torch.empty(B + 1, device=dev, dtype=torch.int32)
    src_r = torch.empty(B * S, device=dev, dtype=torch.int32)
    pos_r = torch.empty(B * S, device=dev, dtype=torch.int32)
    flat = torch.empty(B * S, device=dev, dtype=torch.int32)
    lab = torch.empty(B * S, device=dev, dtype=torch.int64)
    call("eavqa_build_row_plan", B, S, int(pack), _p(mask), _p(labels), _p(src), _p(pos), _p(cu), _p(src_r), _p(pos_r), _p(lab),
         _p(flat), _stream())
    return cu, src_r, pos_r, lab, flat


def _ce_dims(labels: Tensor):
    """(B, S, rows): 2-D labels = unshifted [B,S]; 1-D labels = one shifted label per row (S = 0)."""
    if labels.dim() == 2:
        return labels.shape[0], labels.shape[1], labels.shape[0] * labels.shape[1]
    return labels.shape[0], 0, labels.shape[0]


def ce_fwd(logits: Tensor, labels: Tensor, V: int):
    """logits float32 [rows, ld>=V]; labels int64 [B,S] unshifted or [rows] shifted -> (loss[1], count[1], row_lse)."""
    _dev(logits)
    B, S, rows = _ce_dims(labels)
    row_loss = torch.empty(rows, device=logits.device, dtype=torch.float32)
    row_lse = torch.empty(rows, device=logits.device, dtype=torch.float32)
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    count = torch.empty(1, device=logits.device, dtype=torch.float32)
    call("eavqa_ce_fwd", B, S, V, _p(logits), _ld(logits), _p(labels), _p(row_loss), _p(row_lse), _p(loss), _p(count), _stream())
    return loss, count, row_lse


def guard_count(count: Tensor, capacity: int, loss: Tensor) -> None:
    """``loss[0] = NaN`` on the device when ``count[0] > capacity`` (an under-sized scored-row compaction)."""
    call("eavqa_guard_count", _p(count), int(capacity), _p(loss), _stream())


def ce_bwd(logits: Tensor, labels: Tensor, V: int, row_lse: Tensor, count: Tensor, gscale: Tensor, dtype: torch.dtype,
           ldd: int) -> Tensor:
    B, S, rows = _ce_dims(labels)
    d = torch.empty((rows, ldd), device=logits.device, dtype=dtype)
    call("eavqa_ce_bwd", dtype_id(dtype), B, S, V, _p(logits), _ld(logits), _p(labels), _p(row_lse), _p(count), _p(gscale),
         _p(d), ldd, _stream())
    return d


def greedy_pick(logits: Tensor, V: int, pad_token_id: int, eos_token_id: Optional[int], raw: Tensor, emitted_col: Tensor,
                unfinished: Tensor, logprob: Optional[Tensor] = None, any_unfinished: Optional[Tensor] = None) -> None:
    """``emitted_col``: int64 view [B] of column t of the [B, max_length] token matrix; ``logprob``: float32 [B] or None;
    ``any_unfinished``: a ZEROED int32 element that reads 1 afterwards when some row is still unfinished (the early-stop flag)."""
    B = logits.shape[0]
    call("eavqa_greedy_pick", B, V, _p(logits), _ld(logits), int(pad_token_id if pad_token_id is not None else 0),
         int(eos_token_id) if eos_token_id is not None else -1, _p(raw), _p(emitted_col), emitted_col.stride(0),
         _p(unfinished), _p(logprob), _p(any_unfinished), _stream())


def adamw(param: Tensor, grad: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999,
          eps: float = 1e-8, weight_decay: float = 0.01, grad_scale: float = 1.0, shadow: Optional[Tensor] = None) -> None:
    _dev(param)
    call("eavqa_adamw", param.numel(), _p(param), _p(grad), _p(m), _p(v), int(step), float(lr), float(beta1), float(beta2),
         float(eps), float(weight_decay), float(grad_scale), dtype_id(shadow.dtype) if shadow is not None else 0, _p(shadow), _stream())


def patchify(pixels: Tensor, ps: int, dtype: torch.dtype, ldp: int) -> Tensor:
    _dev(pixels)
    B, C3, img, _ = pixels.shape
    g = img // ps
    pixels = pixels.contiguous().float()
    out = torch.empty((B * g * g, ldp), device=pixels.device, dtype=dtype)
    call("eavqa_patchify", dtype_id(dtype), B, img, ps, _p(pixels), _p(out), ldp, _stream())
    return out


def vit_assemble(patch_embed: Tensor, cls: Tensor, pos: Tensor, B: int, n_patch: int) -> Tensor:
    W = patch_embed.shape[1]
    x = torch.empty((B * (n_patch + 1), W), device=patch_embed.device, dtype=torch.float32)
    call("eavqa_vit_assemble", dtype_id(patch_embed.dtype), B, n_patch, W, _p(patch_embed), _ld(patch_embed), _p(cls), _p(pos),
         _p(x), _ld(x), _stream())
    return x


def cast_rows(x: Tensor, dtype: torch.dtype, out: Optional[Tensor] = None) -> Tensor:
    rows, cols = x.shape
    y = out if out is not None else torch.empty((rows, cols), device=x.device, dtype=dtype)
    call("eavqa_cast_rows", dtype_id(dtype), rows, cols, _p(x), _ld(x), _p(y), _ld(y), _stream())
    return y


def copy_rows(src: Tensor, dst: Tensor, B: int, S: int, cols: int, src_batch_rows: int, dst_batch_rows: int, dst_row0: int) -> None:
    """Strided row copy (KV-cache fill / append); ``src``/``dst`` are 2-D row views."""
    call("eavqa_copy_rows", dtype_id(src.dtype), B, S, cols, _p(src), _ld(src), src_batch_rows, _p(dst), _ld(dst),
         dst_batch_rows, dst_row0, _stream())


def colsum(x: Tensor, out: Tensor, accumulate: bool) -> None:
    rows, cols = x.shape
    call("eavqa_colsum", dtype_id(x.dtype), rows, cols, _p(x), _ld(x), _p(out), int(accumulate), _stream())


def transpose(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    rows, cols = x.shape
    y = out if out is not None else torch.empty((cols, rows), device=x.device, dtype=x.dtype)
    call("eavqa_transpose", dtype_id(x.dtype), rows, cols, _p(x), _ld(x), _p(y), _ld(y), _stream())
    return y


def gemm_splitk(a: Tensor, b: Tensor, ks: Optional[int] = None, unroll: int = 0, out: Optional[Tensor] = None) -> Tensor:
    """fp32 partial sums [ks, M, N] of a [M,K] @ b [N,K]^T (bf16, M <= 64): the decode-step weight-streaming GEMM.
    ``unroll`` (tests / tools only) selects the depth of the weight-load window through ``eavqa_gemm_splitk_ex``."""
    M, K = a.shape
    N = b.shape[0]
    if ks is None:
        ks = int(_lib.load().eavqa_gemm_splitk_plan(M, N, K))
        if ks <= 0:
            raise _lib.EavqaError(f"eavqa_gemm_splitk: unsupported shape M={M} N={N} K={K}")
    part = out if out is not None else torch.empty((ks, M, N), device=a.device, dtype=torch.float32)
    if unroll:
        call("eavqa_gemm_splitk_ex", dtype_id(a.dtype), M, N, K, _p(a), _ld(a), _p(b), _ld(b), _p(part), ks, _stream(), unroll)
    else:
        call("eavqa_gemm_splitk", dtype_id(a.dtype), M, N, K, _p(a), _ld(a), _p(b), _ld(b), _p(part), ks, _stream())
    return part


def splitk_finish(part: Tensor, outs, bias: Optional[Tensor] = None, act: str = "none", residual: Optional[Tensor] = None) -> None:
    """outs: 1..3 2-D tensors (row stride = stride(0)), each receiving N / len(outs) columns of act(sum(part) + bias) (+ residual)."""
    ks, M, N = part.shape
    o = list(outs) + [None] * (3 - len(outs))
    call("eavqa_splitk_finish", dtype_id(outs[0].dtype), M, N, _p(part), ks, _p(bias),
         _lib.ACT[act], _p(residual), residual.stride(0) if residual is not None else 0, int(outs[0].dtype == torch.float32), len(outs),
         _p(o[0]), o[0].stride(0), _p(o[1]), o[1].stride(0) if o[1] is not None else 0, _p(o[2]), o[2].stride(0) if o[2] is not None else 0,
         _stream())


def layernorm_splitk(x_in: Tensor, gamma: Tensor, beta: Tensor, eps: float, out_dtype, part: Optional[Tensor] = None,
                     bias: Optional[Tensor] = None, x_out: Optional[Tensor] = None) -> Tensor:
    rows, cols = x_in.shape
    y = torch.empty((rows, cols), device=x_in.device, dtype=out_dtype)
    call("eavqa_layernorm_splitk", dtype_id(out_dtype), rows, cols, _p(x_in), _ld(x_in), _p(part), 0 if part is None else part.shape[0],
         _p(bias), _p(x_out), _ld(x_out) if x_out is not None else 0, _p(gamma), _p(beta), float(eps), _p(y), _ld(y), _stream())
    return y


def gemm_fp8_splitk(a_q: Tensor, a_scale: Tensor, b_q: Tensor, b_scale: float, ks: Optional[int] = None, out: Optional[Tensor] = None) -> Tensor:
    """fp32 partial sums [ks, M, N] (scales applied) of e4m3 rows ``a_q`` [M, K] x e4m3 weights ``b_q`` [N, K]^T (``eavqa_gemm_fp8_splitk``)."""
    M, K = a_q.shape
    N = b_q.shape[0]
    if ks is None:
        ks = int(_lib.load().eavqa_gemm_fp8_splitk_plan(M, N, K))
        if ks <= 0:
            raise _lib.EavqaError(f"eavqa_gemm_fp8_splitk: unsupported shape M={M} N={N} K={K}")
    part = out if out is not None else torch.empty((ks, M, N), device=a_q.device, dtype=torch.float32)
    call("eavqa_gemm_fp8_splitk", M, N, K, _p(a_q), _ld(a_q), _p(a_scale), _p(b_q), _ld(b_q), float(b_scale), _p(part), ks, _stream())
    return part


def layernorm_splitk_fp8(x_in: Tensor, gamma: Tensor, beta: Tensor, eps: float, part: Optional[Tensor] = None, bias: Optional[Tensor] = None,
                         x_out: Optional[Tensor] = None):
    """``layernorm_splitk`` whose output feeds an fp8 GEMM: (e4m3 bytes uint8 [rows, cols], float32 row scales [rows])."""
    rows, cols = x_in.shape
    yq = torch.empty((rows, cols), device=x_in.device, dtype=torch.uint8)
    sc = torch.empty(rows, device=x_in.device, dtype=torch.float32)
    call("eavqa_layernorm_splitk_fp8", rows, cols, _p(x_in), _ld(x_in), _p(part), 0 if part is None else part.shape[0], _p(bias), _p(x_out),
         _ld(x_out) if x_out is not None else 0, _p(gamma), _p(beta), float(eps), _p(yq), _ld(yq), _p(sc), _stream())
    return yq, sc


def rmsnorm_splitk(x_in: Tensor, gamma: Tensor, eps: float, out_dtype, part: Optional[Tensor] = None, x_out: Optional[Tensor] = None,
                   out: Optional[Tensor] = None) -> Tensor:
    """x = x_in + sum(part) (-> ``x_out``); returns T5LayerNorm(x) in ``out_dtype`` (``eavqa_rmsnorm_splitk``)."""
    rows, cols = x_in.shape
    y = out if out is not None else torch.empty((rows, cols), device=x_in.device, dtype=out_dtype)
    call("eavqa_rmsnorm_splitk", dtype_id(out_dtype), rows, cols, _p(x_in), _ld(x_in), _p(part), 0 if part is None else part.shape[0],
         _p(x_out), _ld(x_out) if x_out is not None else 0, _p(gamma), float(eps), _p(y), _ld(y), _stream())
    return y


def splitk_finish_gated(part: Tensor, act: str, out_dtype) -> Tensor:
    """h = act(sum(part)[:, :F]) * sum(part)[:, F:] for partial sums [ks, M, 2F] (``eavqa_splitk_finish_gated``)."""
    ks, M, N2 = part.shape
    h = torch.empty((M, N2 // 2), device=part.device, dtype=out_dtype)
    call("eavqa_splitk_finish_gated", dtype_id(out_dtype), M, N2 // 2, _p(part), ks, _lib.ACT[act], _p(h), _ld(h), _stream())
    return h


def attention_decode_splitk_rel(part: Tensor, k: Tensor, v: Tensor, B: int, H: int, Sk: int, hd: int, *, kv_batch_rows: int,
                                key_mask: Optional[Tensor] = None, scale: float = 1.0, rel_bias: Optional[Tensor] = None, rel_zero: int = 0) -> Tensor:
    """One decode step of attention whose query (and, with 3 H hd columns, new K / V row) is summed from split-K partial sums
    [ks, B, H hd | 3 H hd]; optional T5 relative-position bias table [H, ld] (``eavqa_attention_decode_splitk_rel``)."""
    ks, _, cols = part.shape
    o = torch.empty((B, H * hd), device=part.device, dtype=k.dtype)
    call("eavqa_attention_decode_splitk_rel", dtype_id(k.dtype), B, H, Sk, hd, _p(part), ks, cols, _p(k), k.stride(0), _p(v), v.stride(0),
         kv_batch_rows, _p(o), _ld(o), _p(key_mask), key_mask.stride(0) if key_mask is not None else 0, float(scale),
         _p(rel_bias), rel_bias.stride(0) if rel_bias is not None else 0, int(rel_zero), _stream())
    return o


def gemm_decode_cols(M: int, N: int, K: int, a_kind: int = 0, gated: bool = False) -> int:
    """Columns of C one workgroup of ``eavqa_gemm_decode`` owns (0: unsupported shape) - the width of one statistics partial."""
    return int(_lib.load().eavqa_gemm_decode_cols(M, N, K, a_kind, int(gated)))


def gemm_decode(a: Tensor, b: Tensor, outs, *, norm: Optional[str] = None, gamma: Optional[Tensor] = None, beta: Optional[Tensor] = None,
                eps: float = 1e-5, stats_in: Optional[Tensor] = None, stats_in_cols: int = 0, gated: bool = False,
                bias: Optional[Tensor] = None, act: str = "none", residual: Optional[Tensor] = None, want_stats: bool = False, sel: int = 0):
    """The decode-step GEMM without partial sums (``eavqa_gemm_decode``): ``outs`` (1..3 2-D tensors, bf16 or fp32) receive equal
    column segments of ``epilogue(norm(a) @ b^T)``.  ``norm``: None (``a`` bf16) | "layer" | "rms" (``a`` = the fp32 stream, ``stats_in``
    [M, n, 2] from the producing call).  ``gated``: ``b`` = [wi_0; wi_1].  Returns the statistics partials [M, n, 2] of the output rows
    when ``want_stats``.  ``sel`` (tests / tools): selector of ``eavqa_gemm_decode_ex``."""
    M, K = a.shape
    N = b.shape[0] // 2 if gated else b.shape[0]
    a_kind = {None: 0, "layer": 1, "rms": 2}[norm]
    g = _lib.DecodeGemm()
    g.M, g.N, g.K = M, N, K
    g.A, g.lda, g.a_kind = _p(a), _ld(a), a_kind
    g.gamma, g.beta, g.eps = _p(gamma), _p(beta), float(eps)
    if a_kind:
        g.stats_in, g.n_stats_in, g.stats_in_cols = _p(stats_in), stats_in.shape[1], int(stats_in_cols)
    g.B, g.ldb, g.gated_rows = _p(b), _ld(b), N if gated else 0
    g.bias, g.act = _p(bias), _lib.ACT[act]
    g.residual, g.ld_residual = _p(residual), residual.stride(0) if residual is not None else 0
    g.out_f32, g.n_seg = int(outs[0].dtype == torch.float32), len(outs)
    for i, o in enumerate(outs):
        g.out[i], g.ld_out[i] = _p(o), o.stride(0)
    stats = None
    if want_stats:
        nf = (sel & 0xF) * (8 if gated else 16) if (sel & 0xF) else gemm_decode_cols(M, N, K, a_kind, gated)
        if nf <= 0:
            raise _lib.EavqaError(f"eavqa_gemm_decode: unsupported shape M={M} N={N} K={K}")
        stats = torch.empty((M, (N + nf - 1) // nf, 2), device=a.device, dtype=torch.float32)
        g.stats_out = _p(stats)
    import ctypes
    if sel:
        call("eavqa_gemm_decode_ex", ctypes.byref(g), _stream(), sel)
    else:
        call("eavqa_gemm_decode", ctypes.byref(g), _stream())
    return stats


def l2_normalize_rows_(x: Tensor) -> Tensor:
    """In place, fp32 [rows, cols] (faiss.normalize_L2)."""
    if x.dtype != torch.float32:
        raise _lib.EavqaError("l2_normalize_rows_: float32 only")
    call("eavqa_l2_normalize_rows", x.shape[0], x.shape[1], _p(x), _ld(x), _stream())
    return x


def topk_rows(scores: Tensor, k: int):
    """(values float32 [rows, k] descending, indices int64 [rows, k]); ties go to the smaller column."""
    if scores.dtype != torch.float32:
        raise _lib.EavqaError("topk_rows: float32 scores only")
    rows, cols = scores.shape
    val = torch.empty((rows, k), device=scores.device, dtype=torch.float32)
    idx = torch.empty((rows, k), device=scores.device, dtype=torch.int64)
    call("eavqa_topk_rows", rows, cols, _p(scores), _ld(scores), int(k), _p(val), _p(idx), _stream())
    return val, idx


def select_rows(row_labels: Tensor, capacity: int):
    """(sel_idx int32 [capacity], sel_labels int64 [capacity], count int32 [1]) of the rows with a label >= 0, in order."""
    M = row_labels.shape[0]
    idx = torch.zeros(capacity, device=row_labels.device, dtype=torch.int32)
    lab = torch.full((capacity,), -100, device=row_labels.device, dtype=torch.int64)
    cnt = torch.zeros(1, device=row_labels.device, dtype=torch.int32)
    call("eavqa_select_rows", M, _p(row_labels), int(capacity), _p(idx), _p(lab), _p(cnt), _stream())
    return idx, lab, cnt


def gather_rows(src: Tensor, idx: Tensor) -> Tensor:
    out = torch.empty((idx.shape[0], src.shape[1]), device=src.device, dtype=src.dtype)
    call("eavqa_move_rows", dtype_id(src.dtype), 0, idx.shape[0], src.shape[1], _p(src), _ld(src), _p(idx), _p(out), _ld(out), _stream())
    return out


def scatter_rows(src: Tensor, idx: Tensor, rows: int) -> Tensor:
    """[rows, cols] zeros with out[idx[i]] = src[i]."""
    out = torch.zeros((rows, src.shape[1]), device=src.device, dtype=src.dtype)
    call("eavqa_move_rows", dtype_id(src.dtype), 1, idx.shape[0], src.shape[1], _p(src), _ld(src), _p(idx), _p(out), _ld(out), _stream())
    return out


# ----------------------------------------------------------------------------------------------------- T5 / T0 (models/t5.py)
_X_KIND = {torch.float32: 1, torch.bfloat16: 2, torch.float16: 3}


def rmsnorm_fwd(x: Tensor, gamma: Optional[Tensor], eps: float, out_dtype: torch.dtype, save_stats: bool = False):
    """T5LayerNorm rows of ``x`` -> ``y`` in ``out_dtype`` (+ rstd)."""
    _dev(x)
    rows, cols = x.shape
    y = torch.empty((rows, cols), device=x.device, dtype=out_dtype)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if save_stats else None
    call("eavqa_rmsnorm_fwd", dtype_id(out_dtype), _X_KIND[x.dtype], rows, cols, _p(x), _ld(x), _p(gamma), float(eps), _p(y), _ld(y), _p(rstd), _stream())
    return (y, rstd) if save_stats else y


def rmsnorm_bwd(x: Tensor, dy: Tensor, gamma: Optional[Tensor], rstd: Tensor, dres: Optional[Tensor] = None, out: Optional[Tensor] = None,
                lowp_out: Optional[Tensor] = None) -> Tensor:
    _dev(x)
    rows, cols = x.shape
    dx = out if out is not None else torch.empty((rows, cols), device=x.device, dtype=torch.float32)
    if dres is not None and _ld(dres) != _ld(dx):
        raise _lib.EavqaError("dres must share dx's leading dimension")
    call("eavqa_rmsnorm_bwd", dtype_id(dy.dtype), _X_KIND[x.dtype], rows, cols, _p(x), _ld(x), _p(dy), _ld(dy), _p(gamma), _p(rstd), _p(dres),
         _p(dx), _ld(dx), _p(lowp_out), _ld(lowp_out) if lowp_out is not None else 0, _stream())
    return dx


def gated_act_fwd(u: Tensor, act: str) -> Tensor:
    """``u`` [rows, 2F] -> ``act(u[:, :F]) * u[:, F:]`` [rows, F]."""
    _dev(u)
    rows, F2 = u.shape
    h = torch.empty((rows, F2 // 2), device=u.device, dtype=u.dtype)
    call("eavqa_gated_act_fwd", dtype_id(u.dtype), rows, F2 // 2, ACT[act], _p(u), _ld(u), _p(h), _ld(h), _stream())
    return h


def gated_act_bwd(u: Tensor, dh: Tensor, act: str) -> Tensor:
    rows, F2 = u.shape
    du = torch.empty_like(u)
    call("eavqa_gated_act_bwd", dtype_id(u.dtype), rows, F2 // 2, ACT[act], _p(u), _ld(u), _p(dh), _ld(dh), _p(du), _ld(du), _stream())
    return du


def attention_fwd_rel(q: Tensor, k: Tensor, v: Tensor, B: int, H: int, Sq: int, Sk: int, hd: int, *, rel_bias: Optional[Tensor], rel_zero: int = 0,
                      key_mask: Optional[Tensor] = None, causal: bool = False, scale: float = 1.0, save_lse: bool = False,
                      q_batch_rows: int = 0, kv_batch_rows: int = 0, ld_mask: int = 0):
    """Attention with T5's relative-position bias table ``rel_bias`` [H, n_offsets] float32 (offset key - query at column + rel_zero)."""
    _dev(q)
    o = torch.empty((B * Sq, H * hd), device=q.device, dtype=q.dtype)
    lse = torch.empty((B, H, Sq), device=q.device, dtype=torch.float32) if save_lse else None
    call("eavqa_attention_fwd_rel", dtype_id(q.dtype), B, H, Sq, Sk, hd, _p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(o), _ld(o), q_batch_rows,
         kv_batch_rows, _p(key_mask), ld_mask, int(causal), float(scale), _p(rel_bias), rel_bias.stride(0) if rel_bias is not None else 0,
         int(rel_zero), _p(lse), _stream())
    return (o, lse) if save_lse else o


def attention_bwd_rel(q, k, v, o, d_o, lse, B, H, Sq, Sk, hd, *, rel_bias=None, rel_zero=0, key_mask=None, causal=False, scale=1.0,
                      dq=None, dk=None, dv=None):
    _dev(q)
    dq = dq if dq is not None else torch.empty((B * Sq, H * hd), device=q.device, dtype=q.dtype)
    dk = dk if dk is not None else torch.empty((B * Sk, H * hd), device=q.device, dtype=q.dtype)
    dv = dv if dv is not None else torch.empty((B * Sk, H * hd), device=q.device, dtype=q.dtype)
    delta = torch.empty((B, H, Sq), device=q.device, dtype=torch.float32)
    call("eavqa_attention_bwd_rel", dtype_id(q.dtype), B, H, Sq, Sk, hd, _p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(o), _ld(o), _p(d_o), _ld(d_o),
         _p(dq), _ld(dq), _p(dk), _ld(dk), _p(dv), _ld(dv), _p(key_mask), int(causal), float(scale), _p(rel_bias),
         rel_bias.stride(0) if rel_bias is not None else 0, int(rel_zero), _p(lse), _p(delta), _stream())
    return dq, dk, dv
