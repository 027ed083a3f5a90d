"""ctypes binding of libeavqa_hip.so (the C ABI declared in include/eavqa.h).

The product path has NO fallback: if the shared library is missing or a call returns an error
code, an exception is raised.  The signature table below mirrors include/eavqa.h one-for-one;
``tests/test_abi.py`` checks that every ``eavqa_*`` function declared in the header is exported
by the library and listed here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libeavqa_hip.so")

F32, BF16 = 0, 1
ACT = {"none": 0, "tanh": 1, "relu": 2, "gelu_new": 3, "quick_gelu": 4}

i32, i64, f32, ptr = C.c_int, C.c_int64, C.c_float, C.c_void_p

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "eavqa_abi_version": [],
    "eavqa_strerror": [i32],
    "eavqa_check_device": [],
    "eavqa_gemm": [i32, i32, i32, i32, i32, i32, ptr, i64, ptr, i64, ptr, i64, i32, f32, ptr, i32, ptr, ptr, i64, ptr, i64, ptr],
    "eavqa_layernorm_fwd": [i32, i32, i32, i32, ptr, i64, ptr, ptr, f32, ptr, i64, ptr, ptr, ptr],
    "eavqa_layernorm_bwd": [i32, i32, i32, i32, ptr, i64, ptr, i64, ptr, ptr, ptr, ptr, ptr, i64, ptr, ptr, ptr, i64, ptr],
    "eavqa_layernorm_fwd_fp8": [i32, i32, i32, ptr, i64, ptr, ptr, f32, ptr, i64, ptr, ptr, ptr, ptr],
    "eavqa_layernorm_bwd_fp8": [i32, i32, i32, ptr, i64, ptr, i64, ptr, ptr, ptr, ptr, ptr, i64, ptr, i64, ptr, ptr],
    "eavqa_attention_fwd": [i32, i32, i32, i32, i32, i32, ptr, i64, ptr, i64, ptr, i64, ptr, i64, i64, i64, ptr, i64, ptr, i32, f32, ptr, ptr],
    "eavqa_attention_decode": [i32, i32, i32, i32, i32, ptr, i64, ptr, i64, ptr, i64, i64, ptr, ptr, i64, ptr, i64, ptr, i64, f32, ptr],
    "eavqa_attention_decode_splitk": [i32, i32, i32, i32, i32, ptr, i32, ptr, ptr, i64, ptr, i64, i64, ptr, i64, ptr, i64, f32, ptr],
    "eavqa_attention_bwd": [i32, i32, i32, i32, i32, i32, ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i64,
                            ptr, i64, ptr, i64, ptr, i64, ptr, ptr, i32, f32, ptr, ptr, ptr],
    "eavqa_build_prefix_rows": [i32, i32, i32, ptr, ptr, i32, i32, i32, ptr, ptr, ptr, ptr],
    "eavqa_build_fewshot_rows": [i32, i32, i32, i32, i64, ptr, ptr, i32, ptr, ptr, ptr, ptr, ptr],
    "eavqa_build_row_plan": [i32, i32, i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr],
    "eavqa_copy_rows": [i32, i32, i32, i32, ptr, i64, i64, ptr, i64, i64, i64, ptr],
    "eavqa_transpose": [i32, i32, i32, ptr, i64, ptr, i64, ptr],
    "eavqa_select_rows": [i32, ptr, i32, ptr, ptr, ptr, ptr],
    "eavqa_move_rows": [i32, i32, i32, i32, ptr, i64, ptr, ptr, i64, ptr],
    "eavqa_l2_normalize_rows": [i32, i32, ptr, i64, ptr],
    "eavqa_topk_rows": [i32, i32, ptr, i64, i32, ptr, ptr, ptr],
    "eavqa_gemm_splitk_plan": [i32, i32, i32],
    "eavqa_gemm_splitk": [i32, i32, i32, i32, ptr, i64, ptr, i64, ptr, i32, ptr],
    "eavqa_splitk_finish": [i32, i32, i32, ptr, i32, ptr, i32, ptr, i64, i32, i32, ptr, i64, ptr, i64, ptr, i64, ptr],
    "eavqa_layernorm_splitk": [i32, i32, i32, ptr, i64, ptr, i32, ptr, ptr, i64, ptr, ptr, f32, ptr, i64, ptr],
    "eavqa_gemm_fp8_splitk_plan": [i32, i32, i32],
    "eavqa_gemm_fp8_splitk": [i32, i32, i32, ptr, i64, ptr, ptr, i64, f32, ptr, i32, ptr],
    "eavqa_layernorm_splitk_fp8": [i32, i32, ptr, i64, ptr, i32, ptr, ptr, i64, ptr, ptr, f32, ptr, i64, ptr, ptr],
    "eavqa_rmsnorm_splitk": [i32, i32, i32, ptr, i64, ptr, i32, ptr, i64, ptr, f32, ptr, i64, ptr],
    "eavqa_splitk_finish_gated": [i32, i32, i32, ptr, i32, i32, ptr, i64, ptr],
    "eavqa_attention_decode_splitk_rel": [i32, i32, i32, i32, i32, ptr, i32, i32, ptr, i64, ptr, i64, i64, ptr, i64, ptr, i64, f32, ptr, i64, i32, ptr],
    "eavqa_colsum": [i32, i32, i32, ptr, i64, ptr, i32, ptr],
    "eavqa_embed_assemble": [i32, i32, i32, ptr, ptr, ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr],
    "eavqa_embed_assemble_bwd": [i32, i32, i32, ptr, ptr, i64, ptr, i64, ptr],
    "eavqa_build_labels": [i32, i32, i32, i32, ptr, i64, i64, ptr, ptr],
    "eavqa_ce_fwd": [i32, i32, i32, ptr, i64, ptr, ptr, ptr, ptr, ptr, ptr],
    "eavqa_guard_count": [ptr, i32, ptr, ptr],
    "eavqa_ce_bwd": [i32, i32, i32, i32, ptr, i64, ptr, ptr, ptr, ptr, ptr, i64, ptr],
    "eavqa_greedy_pick": [i32, i32, ptr, i64, i64, i64, ptr, ptr, i64, ptr, ptr, ptr, ptr],
    "eavqa_adamw": [i64, ptr, ptr, ptr, ptr, i32, f32, f32, f32, f32, f32, f32, i32, ptr, ptr],
    "eavqa_patchify": [i32, i32, i32, i32, ptr, ptr, i64, ptr],
    "eavqa_vit_assemble": [i32, i32, i32, i32, ptr, i64, ptr, ptr, ptr, i64, ptr],
    "eavqa_cast_rows": [i32, i32, i64, ptr, i64, ptr, i64, ptr],
    "eavqa_quantize_rows_fp8": [i32, i32, i32, ptr, i64, ptr, i64, ptr, ptr],
    "eavqa_rmsnorm_fwd": [i32, i32, i32, i32, ptr, i64, ptr, f32, ptr, i64, ptr, ptr],
    "eavqa_rmsnorm_bwd": [i32, i32, i32, i32, ptr, i64, ptr, i64, ptr, ptr, ptr, ptr, i64, ptr, i64, ptr],
    "eavqa_gated_act_fwd": [i32, i32, i32, i32, ptr, i64, ptr, i64, ptr],
    "eavqa_gated_act_bwd": [i32, i32, i32, i32, ptr, i64, ptr, i64, ptr, i64, ptr],
    "eavqa_attention_fwd_rel": [i32, i32, i32, i32, i32, i32, ptr, i64, ptr, i64, ptr, i64, ptr, i64, i64, i64, ptr, i64, i32, f32, ptr, i64, i32, ptr, ptr],
    "eavqa_attention_bwd_rel": [i32, i32, i32, i32, i32, i32, ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i64,
                                ptr, i32, f32, ptr, i64, i32, ptr, ptr, ptr],
    "eavqa_gemm_fp8": [i32, i32, i32, ptr, i64, ptr, ptr, i64, f32, ptr, i64, i32, f32, ptr, i32, ptr, ptr, i64, ptr, i64, ptr, i32],
}
class LMLayer(C.Structure):
    """``eavqa_lm_layer_t``."""
    _fields_ = [(n, C.c_void_p) for n in ("ln1_g", "ln1_b", "w_qkv", "b_qkv", "w_o", "b_o", "ln2_g", "ln2_b", "w_fc1", "b_fc1",
                                          "w_fc2", "b_fc2", "k_cache", "v_cache")]


class LMLayerScales(C.Structure):
    """``eavqa_lm_layer_scales_t``."""
    _fields_ = [(n, C.c_float) for n in ("s_qkv", "s_o", "s_fc1", "s_fc2")]


class T5DecLayer(C.Structure):
    """``eavqa_t5_dec_layer_t``."""
    _fields_ = [(n, C.c_void_p) for n in ("ln_sa", "w_qkv", "w_o", "ln_ca", "w_q_ca", "w_o_ca", "ln_ff", "w_i", "w_o_ff", "k_cache", "v_cache",
                                          "cross_kv")]


class DecodeGemm(C.Structure):
    """``eavqa_decode_gemm_t``."""
    _fields_ = [("M", i32), ("N", i32), ("K", i32), ("A", ptr), ("lda", i64), ("a_kind", i32), ("gamma", ptr), ("beta", ptr), ("eps", f32),
                ("stats_in", ptr), ("n_stats_in", i32), ("stats_in_cols", i32), ("B", ptr), ("ldb", i64), ("gated_rows", i32),
                ("bias", ptr), ("act", i32), ("residual", ptr), ("ld_residual", i64), ("out_f32", i32), ("n_seg", i32),
                ("out", ptr * 3), ("ld_out", i64 * 3), ("stats_out", ptr)]


class GemmLn(C.Structure):
    """``eavqa_gemm_ln_t``."""
    _fields_ = [("copy_out", ptr), ("ld_copy", i64), ("stats_out", ptr), ("stats_ld", i32), ("ln_stats", ptr), ("ln_parts", i32),
                ("ln_ld", i32), ("ln_cols", i32), ("ln_c", ptr), ("ln_eps", f32), ("mean_out", ptr), ("rstd_out", ptr)]


SIGNATURES["eavqa_gemm_ln"] = SIGNATURES["eavqa_gemm"][:-1] + [C.POINTER(GemmLn), ptr]
SIGNATURES["eavqa_gemm_decode_cols"] = [i32, i32, i32, i32, i32]
SIGNATURES["eavqa_gemm_decode"] = [C.POINTER(DecodeGemm), ptr]
SIGNATURES["eavqa_t5_decoder_step_workspace_bytes"] = [i32, i32, i32, i32, i32, i32]
SIGNATURES["eavqa_t5_decoder_step"] = [i32, i32, C.POINTER(T5DecLayer), ptr, i32, i32, i32, i32, i32, i32, f32, i32, i32, i32, i32, ptr, ptr, ptr, i64,
                                       ptr, i64, i32, ptr, i64, ptr]
SIGNATURES["eavqa_lm_block_workspace_bytes"] = [i32, i32, i32, i32]
SIGNATURES["eavqa_lm_block_fp8_workspace_bytes"] = [i32, i32, i32]
SIGNATURES["eavqa_lm_block_forward_fp8"] = [i32, C.POINTER(LMLayer), C.POINTER(LMLayerScales), i32, i32, i32, i32, f32, i32, i32, i32, i32, ptr, ptr, i64, ptr, i64, ptr]
SIGNATURES["eavqa_lm_block_forward"] = [i32, i32, C.POINTER(LMLayer), i32, i32, i32, i32, f32, i32, i32, i32, i32, ptr, ptr, i64, ptr, i64, ptr]

# include/eavqa_test.h: the same entry points with an explicit kernel selector (tests and tools only)
SIGNATURES["eavqa_gemm_ex"] = SIGNATURES["eavqa_gemm"] + [i32]
SIGNATURES["eavqa_gemm_ln_ex"] = SIGNATURES["eavqa_gemm_ln"] + [i32]
SIGNATURES["eavqa_attention_fwd_ex"] = SIGNATURES["eavqa_attention_fwd"] + [i32]
SIGNATURES["eavqa_attention_bwd_ex"] = SIGNATURES["eavqa_attention_bwd"] + [i32]
SIGNATURES["eavqa_gemm_splitk_ex"] = SIGNATURES["eavqa_gemm_splitk"] + [i32]
SIGNATURES["eavqa_lm_block_forward_ex"] = SIGNATURES["eavqa_lm_block_forward"] + [i32]
SIGNATURES["eavqa_gemm_decode_ex"] = SIGNATURES["eavqa_gemm_decode"] + [i32]
SIGNATURES["eavqa_t5_decoder_step_ex"] = SIGNATURES["eavqa_t5_decoder_step"] + [i32]

_RESTYPES = {"eavqa_strerror": C.c_char_p, "eavqa_lm_block_workspace_bytes": C.c_int64, "eavqa_lm_block_fp8_workspace_bytes": C.c_int64, "eavqa_t5_decoder_step_workspace_bytes": C.c_int64}

_lib = None


class EavqaError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EavqaError(
            f"{LIB_PATH} is missing: the HIP extension is the only compute path. "
            "Build it with `python -m eavqa_amd.build` (needs hipcc, no GPU required to compile)."
        )
    # torch ships its own libamdhip64: it must be in the process before this library asks the loader for that SONAME,
    # otherwise the system runtime gets bound first and torch later runs on a HIP runtime it was not built against
    # (seen as hipGetDevice failing when build() and smoke() share one process)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    got = lib.eavqa_abi_version()
    if got != 1:
        raise EavqaError(f"libeavqa_hip.so ABI version {got}, host code expects 1")
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    """Invoke an int-returning entry point and raise on a non-zero code."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise EavqaError(f"{name} failed: {lib.eavqa_strerror(rc).decode()} (code {rc})")
