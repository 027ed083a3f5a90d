// Host-side driver: ONE cached greedy step through all decoder layers of a frozen T5 (eavqa_t5_decoder_step in include/eavqa.h) -
// the calls of FrozenT5.decode_step (models/t5.py) in the same order on the same kernels, enqueued from C++ instead of ~340 ctypes
// calls per step (3 ms of Python for ~2.2 ms of GPU work on T0_3B).  Pure enqueue: no allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "eavqa.h"
#include "eavqa_test.h"

namespace {
inline size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }
}

extern "C" int64_t eavqa_t5_decoder_step_workspace_bytes(int dtype, int B, int E, int inner, int F, int gated) {
    const size_t es = dtype == EAVQA_BF16 ? 2 : 4;
    size_t b = 0;
    b += align_up((size_t)B * E * es);                         // RMSNorm output (operand of the next GEMM)
    b += align_up((size_t)B * 3 * inner * es);                 // q | k | v of the new position
    b += align_up((size_t)B * inner * es);                     // attention output (self, then cross)
    b += align_up((size_t)B * inner * es);                     // cross-attention query
    b += 2 * align_up((size_t)B * E * 4);                      // residual stream after self / cross attention (float32)
    b += align_up((size_t)B * (gated ? 2 : 1) * F * es);       // FFN up-projection ([wi_0 x | wi_1 x] when gated)
    b += align_up((size_t)B * F * es);                         // gated activation
    return (int64_t)b;
}

extern "C" int eavqa_t5_decoder_step(int dtype, int n_layer, const eavqa_t5_dec_layer_t* layers, const float* ln_final, int E, int inner, int H,
                                     int F, int gated, int act, float eps, int B, int t, int t_max, int S, float* x, void* out,
                                     const int32_t* enc_mask, int64_t ld_mask, const float* rel_bias, int64_t rel_ld, int rel_zero,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
    if (!layers || !ln_final || !x || !out || !workspace || n_layer <= 0 || B <= 0 || t <= 0 || t > t_max || S <= 0) return EAVQA_E_ARG;
    if (dtype != EAVQA_BF16 && dtype != EAVQA_F32) return EAVQA_E_DTYPE;
    if (inner % H) return EAVQA_E_SHAPE;
    if (workspace_bytes < eavqa_t5_decoder_step_workspace_bytes(dtype, B, E, inner, F, gated)) return EAVQA_E_ARG;
    const size_t es = dtype == EAVQA_BF16 ? 2 : 4;
    const int dkv = inner / H, I = inner;
    const int xk = 1;                                          // x_kind of eavqa_rmsnorm_fwd: the residual stream is float32
    char* w = static_cast<char*>(workspace);
    void* a = w;            w += align_up((size_t)B * E * es);
    char* qkv = w;          w += align_up((size_t)B * 3 * I * es);
    void* ctx = w;          w += align_up((size_t)B * I * es);
    void* qc = w;           w += align_up((size_t)B * I * es);
    float* x1 = reinterpret_cast<float*>(w); w += align_up((size_t)B * E * 4);
    float* x2 = reinterpret_cast<float*>(w); w += align_up((size_t)B * E * 4);
    void* u = w;            w += align_up((size_t)B * (gated ? 2 : 1) * F * es);
    void* h = w;
    int rc;
    for (int l = 0; l < n_layer; ++l) {
        const eavqa_t5_dec_layer_t& L = layers[l];
        // self-attention: the new position's q / k / v, K and V appended to the cache at row t - 1, one query at the end of t keys
        if ((rc = eavqa_rmsnorm_fwd(dtype, xk, B, E, x, E, L.ln_sa, eps, a, E, nullptr, stream))) return rc;
        if ((rc = eavqa_gemm(dtype, 1, 1, B, 3 * I, E, a, E, L.w_qkv, E, qkv, 3 * I, 0, 1.f, nullptr, EAVQA_ACT_NONE, nullptr, nullptr, 0, nullptr, 0, stream))) return rc;
        if ((rc = eavqa_copy_rows(dtype, B, 1, I, qkv + (size_t)I * es, 3 * I, 1, L.k_cache, I, t_max, t - 1, stream))) return rc;
        if ((rc = eavqa_copy_rows(dtype, B, 1, I, qkv + (size_t)2 * I * es, 3 * I, 1, L.v_cache, I, t_max, t - 1, stream))) return rc;
        if ((rc = eavqa_attention_fwd_rel(dtype, B, H, 1, t, dkv, qkv, 3 * I, L.k_cache, I, L.v_cache, I, ctx, I, 1, t_max, nullptr, 0, 1, 1.f,
                                          rel_bias, rel_ld, rel_zero, nullptr, stream))) return rc;
        if ((rc = eavqa_gemm(dtype, 1, 1, B, E, I, ctx, I, L.w_o, I, x1, E, EAVQA_GEMM_OUT_F32, 1.f, nullptr, EAVQA_ACT_NONE, nullptr, nullptr, 0, x, E, stream))) return rc;
        // cross-attention over the encoder output (K / V of every layer computed once by the caller)
        if ((rc = eavqa_rmsnorm_fwd(dtype, xk, B, E, x1, E, L.ln_ca, eps, a, E, nullptr, stream))) return rc;
        if ((rc = eavqa_gemm(dtype, 1, 1, B, I, E, a, E, L.w_q_ca, E, qc, I, 0, 1.f, nullptr, EAVQA_ACT_NONE, nullptr, nullptr, 0, nullptr, 0, stream))) return rc;
        const char* ckv = static_cast<const char*>(L.cross_kv);
        if ((rc = eavqa_attention_fwd_rel(dtype, B, H, 1, S, dkv, qc, I, ckv, 2 * I, ckv + (size_t)I * es, 2 * I, ctx, I, 0, 0, enc_mask, ld_mask, 0, 1.f,
                                          nullptr, 0, 0, nullptr, stream))) return rc;
        if ((rc = eavqa_gemm(dtype, 1, 1, B, E, I, ctx, I, L.w_o_ca, I, x2, E, EAVQA_GEMM_OUT_F32, 1.f, nullptr, EAVQA_ACT_NONE, nullptr, nullptr, 0, x1, E, stream))) return rc;
        // feed-forward
        if ((rc = eavqa_rmsnorm_fwd(dtype, xk, B, E, x2, E, L.ln_ff, eps, a, E, nullptr, stream))) return rc;
        if (gated) {
            if ((rc = eavqa_gemm(dtype, 1, 1, B, 2 * F, E, a, E, L.w_i, E, u, 2 * F, 0, 1.f, nullptr, EAVQA_ACT_NONE, nullptr, nullptr, 0, nullptr, 0, stream))) return rc;
            if ((rc = eavqa_gated_act_fwd(dtype, B, F, act, u, 2 * F, h, F, stream))) return rc;
        } else {
            if ((rc = eavqa_gemm(dtype, 1, 1, B, F, E, a, E, L.w_i, E, h, F, 0, 1.f, nullptr, act, nullptr, nullptr, 0, nullptr, 0, stream))) return rc;
        }
        if ((rc = eavqa_gemm(dtype, 1, 1, B, E, F, h, F, L.w_o_ff, F, x, E, EAVQA_GEMM_OUT_F32, 1.f, nullptr, EAVQA_ACT_NONE, nullptr, nullptr, 0, x2, E, stream))) return rc;
    }
    return eavqa_rmsnorm_fwd(dtype, xk, B, E, x, E, ln_final, eps, out, E, nullptr, stream);
}
