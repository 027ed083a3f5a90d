// Host-side driver: ONE cached greedy step through all decoder layers of a frozen T5 (eavqa_t5_decoder_step in include/eavqa.h) -
// the calls of FrozenT5.decode_step (models/t5.py) in the same order on the same kernels, enqueued from C++ instead of ~340 ctypes
// calls per step (3 ms of Python for ~2.2 ms of GPU work on T0_3B).  Pure enqueue: no allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "eavqa.h"
#include "eavqa_test.h"

namespace {
inline size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

// Split-K route (round 4): every projection of the step streams its weights once with K cut over workgroups (eavqa_gemm_splitk) and its
// consumer adds the partial sums up - the RMSNorm pass (residual + finish + norm in one kernel), the decode attention (q, and the new K / V
// row it appends, summed from the partial sums), the gated finish.  12 kernels per layer instead of 14, the six GEMMs at 3-4 TB/s instead
// of the M <= 64 tile kernels' 1.3 (T0_3B, B = 32: 3.1 -> about 2.2 ms per step).  bf16, B <= 64, shapes every plan accepts.
struct T5Plan { int qkv, o, qc, wi, wo; bool ok; size_t part_bytes; };
inline T5Plan plan_t5(int dtype, int B, int E, int I, int H, int F, int gated, int t, int S) {
    T5Plan p{0, 0, 0, 0, 0, false, 0};
    if (dtype != EAVQA_BF16 || B > 64 || E % 8 || I % 8 || F % 4 || H <= 0) return p;
    const int dkv = I / H;
    if (dkv % 8 || dkv > 128 || t > 3584 || S > 3584) return p;
    const int NI = (gated ? 2 : 1) * F;
    p.qkv = eavqa_gemm_splitk_plan(B, 3 * I, E);
    p.o = eavqa_gemm_splitk_plan(B, E, I);
    p.qc = eavqa_gemm_splitk_plan(B, I, E);
    p.wi = eavqa_gemm_splitk_plan(B, NI, E);
    p.wo = eavqa_gemm_splitk_plan(B, E, F);
    p.ok = p.qkv > 0 && p.o > 0 && p.qc > 0 && p.wi > 0 && p.wo > 0;
    size_t m = (size_t)p.qkv * 3 * I;
    const size_t c[] = {(size_t)p.o * E, (size_t)p.qc * I, (size_t)p.wi * NI, (size_t)p.wo * E};
    for (size_t v : c) m = v > m ? v : m;
    p.part_bytes = align_up(m * B * 4);
    return p;
}
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
    const T5Plan p = plan_t5(dtype, B, E, inner, inner / 64 > 0 ? inner / 64 : 1, F, gated, 1, 1);     // (head count does not enter the sizes)
    if (p.ok) b += p.part_bytes;                               // partial sums of the split-K route (one buffer: producer and consumer alternate)
    return (int64_t)b;
}

static int t5_decoder_step_impl(int dtype, int n_layer, const eavqa_t5_dec_layer_t* layers, const float* ln_final, int E, int inner, int H,
                                int F, int gated, int act, float eps, int B, int t, int t_max, int S, float* x, void* out,
                                const int32_t* enc_mask, int64_t ld_mask, const float* rel_bias, int64_t rel_ld, int rel_zero,
                                void* workspace, int64_t workspace_bytes, void* stream, int route) {
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
    void* h = w;            w += align_up((size_t)B * F * es);
    int rc;
    const T5Plan P = plan_t5(dtype, B, E, I, H, F, gated, t, S);
    if (P.ok && route != 1 && rel_bias) {
        float* part = reinterpret_cast<float*>(w);
        const int NI = (gated ? 2 : 1) * F;
        for (int l = 0; l < n_layer; ++l) {
            const eavqa_t5_dec_layer_t& L = layers[l];
            // x = x2 + sum(feed-forward partials of the previous layer); a = RMSNorm(x)
            if (l == 0) rc = eavqa_rmsnorm_splitk(dtype, B, E, x, E, nullptr, 0, nullptr, 0, L.ln_sa, eps, a, E, stream);
            else rc = eavqa_rmsnorm_splitk(dtype, B, E, x2, E, part, P.wo, x, E, L.ln_sa, eps, a, E, stream);
            if (rc) return rc;
            // self-attention: q and the new K / V row summed from the QKV partial sums, K / V appended at row t - 1, bias of offsets -(t-1) .. 0
            if ((rc = eavqa_gemm_splitk(dtype, B, 3 * I, E, a, E, L.w_qkv, E, part, P.qkv, stream))) return rc;
            if ((rc = eavqa_attention_decode_splitk_rel(dtype, B, H, t, dkv, part, P.qkv, 3 * I, L.k_cache, I, L.v_cache, I, t_max, ctx, I, nullptr, 0, 1.f,
                                                        rel_bias, rel_ld, rel_zero, stream))) return rc;
            if ((rc = eavqa_gemm_splitk(dtype, B, E, I, ctx, I, L.w_o, I, part, P.o, stream))) return rc;
            if ((rc = eavqa_rmsnorm_splitk(dtype, B, E, x, E, part, P.o, x1, E, L.ln_ca, eps, a, E, stream))) return rc;
            // cross-attention: q summed from its partial sums, K / V of the encoder output
            if ((rc = eavqa_gemm_splitk(dtype, B, I, E, a, E, L.w_q_ca, E, part, P.qc, stream))) return rc;
            char* ckv = static_cast<char*>(const_cast<void*>(L.cross_kv));
            if ((rc = eavqa_attention_decode_splitk_rel(dtype, B, H, S, dkv, part, P.qc, I, ckv, 2 * I, ckv + (size_t)I * es, 2 * I, S, ctx, I, enc_mask, ld_mask,
                                                        1.f, nullptr, 0, 0, stream))) return rc;
            if ((rc = eavqa_gemm_splitk(dtype, B, E, I, ctx, I, L.w_o_ca, I, part, P.o, stream))) return rc;
            if ((rc = eavqa_rmsnorm_splitk(dtype, B, E, x1, E, part, P.o, x2, E, L.ln_ff, eps, a, E, stream))) return rc;
            // feed-forward
            if ((rc = eavqa_gemm_splitk(dtype, B, NI, E, a, E, L.w_i, E, part, P.wi, stream))) return rc;
            if (gated) rc = eavqa_splitk_finish_gated(dtype, B, F, part, P.wi, act, h, F, stream);
            else rc = eavqa_splitk_finish(dtype, B, F, part, P.wi, nullptr, act, nullptr, 0, 0, 1, h, F, nullptr, 0, nullptr, 0, stream);
            if (rc) return rc;
            if ((rc = eavqa_gemm_splitk(dtype, B, E, F, h, F, L.w_o_ff, F, part, P.wo, stream))) return rc;
        }
        // out = RMSNorm(x2 + sum(last feed-forward partials))
        return eavqa_rmsnorm_splitk(dtype, B, E, x2, E, part, P.wo, nullptr, 0, ln_final, eps, out, E, stream);
    }
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

extern "C" int eavqa_t5_decoder_step(int dtype, int n_layer, const eavqa_t5_dec_layer_t* layers, const float* ln_final, int E, int inner, int H,
                                     int F, int gated, int act, float eps, int B, int t, int t_max, int S, float* x, void* out,
                                     const int32_t* enc_mask, int64_t ld_mask, const float* rel_bias, int64_t rel_ld, int rel_zero,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
    return t5_decoder_step_impl(dtype, n_layer, layers, ln_final, E, inner, H, F, gated, act, eps, B, t, t_max, S, x, out, enc_mask, ld_mask, rel_bias,
                                rel_ld, rel_zero, workspace, workspace_bytes, stream, 0);
}

extern "C" int eavqa_t5_decoder_step_ex(int dtype, int n_layer, const eavqa_t5_dec_layer_t* layers, const float* ln_final, int E, int inner, int H,
                                        int F, int gated, int act, float eps, int B, int t, int t_max, int S, float* x, void* out,
                                        const int32_t* enc_mask, int64_t ld_mask, const float* rel_bias, int64_t rel_ld, int rel_zero,
                                        void* workspace, int64_t workspace_bytes, void* stream, int route) {
    return t5_decoder_step_impl(dtype, n_layer, layers, ln_final, E, inner, H, F, gated, act, eps, B, t, t_max, S, x, out, enc_mask, ld_mask, rel_bias,
                                rel_ld, rel_zero, workspace, workspace_bytes, stream, route);
}
