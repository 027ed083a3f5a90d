// Host-side driver: all decoder layers of a frozen causal LM for `Sq` new positions per sample in ONE C call
// (eavqa_lm_block_forward in include/eavqa.h).  Used by generation (prefill: Sq = prompt length, decode: Sq = 1):
// a decode step is 9 short kernels per layer, and enqueueing them from Python one by one made the step host-bound
// (5 ms of Python for ~2 ms of GPU work on OPT-2.7B).  Pure enqueue: no allocation (the caller passes a workspace),
// no synchronisation, graph-capturable.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "eavqa.h"
#include "eavqa_test.h"

namespace {
inline size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

// decode fast path: one new position per sample, bf16, at most 64 samples, every GEMM shape plannable
struct DecodePlan { int ks_qkv, ks_o, ks_fc1, ks_fc2; bool ok; size_t part_bytes; };
inline DecodePlan plan_decode(int dtype, int rows, int Sq, int E, int F) {
    DecodePlan d{0, 0, 0, 0, false, 0};
    if (dtype != EAVQA_BF16 || Sq != 1 || rows > 64 || E % 4 || F % 4) return d;
    d.ks_qkv = eavqa_gemm_splitk_plan(rows, 3 * E, E);
    d.ks_o = eavqa_gemm_splitk_plan(rows, E, E);
    d.ks_fc1 = eavqa_gemm_splitk_plan(rows, F, E);
    d.ks_fc2 = eavqa_gemm_splitk_plan(rows, E, F);
    d.ok = d.ks_qkv > 0 && d.ks_o > 0 && d.ks_fc1 > 0 && d.ks_fc2 > 0;
    size_t a = (size_t)d.ks_qkv * rows * 3 * E, b = (size_t)d.ks_o * rows * E, c = (size_t)d.ks_fc1 * rows * F;
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    d.part_bytes = align_up(m * 4) + align_up((size_t)d.ks_fc2 * rows * E * 4);   // scratch partials + the FFN-down partials that
    return d;                                                                       // live until the next layer's LayerNorm
}
}

extern "C" int64_t eavqa_lm_block_workspace_bytes(int dtype, int rows, int E, int F) {
    const size_t es = dtype == EAVQA_BF16 ? 2 : 4;
    size_t b = 0;
    b += align_up((size_t)rows * E * es);        // a / a2 (LayerNorm output)
    b += align_up((size_t)rows * 3 * E * es);    // qkv
    b += align_up((size_t)rows * E * es);        // attention output
    b += align_up((size_t)rows * E * 4);         // x1 (fp32 residual stream after attention)
    b += align_up((size_t)rows * F * es);        // FFN activation
    const DecodePlan d = plan_decode(dtype, rows, 1, E, F);      // rows <= 64: may be a decode step (Sq = 1)
    if (d.ok) b += d.part_bytes;
    return (int64_t)b;
}

// route (include/eavqa_test.h): selects among decode-step structures for A / B measurements and parity tests; 0 = what the library ships.
// (Round 2's persistent one-kernel step - route 2 then - measured 2.25x slower and now lives under tools/experiments/persistent_decode.)
static int lm_block_forward_impl(int dtype, int n_layer, const eavqa_lm_layer_t* layers, int E, int H, int F, int act,
                                 float eps, int B, int Sq, int row0, int S_max, float* x, const int32_t* key_mask,
                                 int64_t ld_mask, void* workspace, int64_t workspace_bytes, void* stream, int route) {
    if (!layers || !x || !workspace || n_layer <= 0 || B <= 0 || Sq <= 0 || row0 < 0 || S_max < row0 + Sq) return EAVQA_E_ARG;
    if (E % H) return EAVQA_E_SHAPE;
    const int rows = B * Sq, hd = E / H, Sk = row0 + Sq;
    if (workspace_bytes < eavqa_lm_block_workspace_bytes(dtype, rows, E, F)) return EAVQA_E_ARG;
    const size_t es = dtype == EAVQA_BF16 ? 2 : 4;
    char* w = static_cast<char*>(workspace);
    void* a = w;            w += align_up((size_t)rows * E * es);
    char* qkv = w;          w += align_up((size_t)rows * 3 * E * es);
    void* ctx = w;          w += align_up((size_t)rows * E * es);
    float* x1 = reinterpret_cast<float*>(w); w += align_up((size_t)rows * E * 4);
    void* f = w;            w += align_up((size_t)rows * F * es);
    const float scale = 1.0f / sqrtf((float)hd);
    int rc;
    const DecodePlan d = plan_decode(dtype, rows, Sq, E, F);
    if (d.ok) {
        // ---- decode step: split-K weight streaming; every GEMM leaves fp32 partial sums that its consumer adds up
        // Alternative route (QKV and FFN-up through eavqa_gemm's M <= 64 tile kernels with the bias / activation in the epilogue, the
        // decode attention appending K / V itself: 7 kernels per layer instead of 9).  MEASURED SLOWER in the step (OPT-2.7B, B = 32:
        // 3.42 against 3.16 ms per step) although the tile GEMM wins on cache-resident weights (33 against 111 us for the mapper's
        // second Linear): with every layer's weights coming cold from HBM the 96-128 workgroups of a 128 x 80 tiling keep too few
        // bytes in flight, while the split-K kernels spread each product over 500+ workgroups.  Kept selectable for that experiment
        // (the compile-time constant below), not taken.
        constexpr bool kFusedDecode = false;
        const bool fused = kFusedDecode && (E % 64) == 0 && (hd % 8) == 0 && hd <= 128 && Sk <= 3584;
        const bool attn_from_partials = (hd % 8) == 0 && hd <= 128 && Sk <= 3584 && (E % 8) == 0;
        float* part = reinterpret_cast<float*>(w);
        float* part2 = reinterpret_cast<float*>(w + (d.part_bytes - align_up((size_t)d.ks_fc2 * rows * E * 4)));
        for (int l = 0; l < n_layer; ++l) {
            const eavqa_lm_layer_t& L = layers[l];
            // x = x1 + b_fc2 + sum(FFN-down partials of the previous layer); a = LN1(x)
            if (l == 0) rc = eavqa_layernorm_splitk(dtype, rows, E, x, E, nullptr, 0, nullptr, nullptr, 0, L.ln1_g, L.ln1_b, eps, a, E, stream);
            else rc = eavqa_layernorm_splitk(dtype, rows, E, x1, E, part2, d.ks_fc2, layers[l - 1].b_fc2, x, E, L.ln1_g, L.ln1_b, eps, a, E, stream);
            if (rc) return rc;
            if (fused) {
                // wide N, short K: no split - the weight-streaming tile GEMM (eavqa_gemm's M <= 64 route) adds the bias itself, and
                // the decode attention takes the new K / V rows straight from its output and appends them to the cache (7 kernels
                // per layer instead of 9: no finish pass, no partial sums for this product)
                if ((rc = eavqa_gemm(dtype, 1, 1, rows, 3 * E, E, a, E, L.w_qkv, E, qkv, 3 * E, 0, 1.f, L.b_qkv, EAVQA_ACT_NONE, nullptr, nullptr, 0,
                                     nullptr, 0, stream))) return rc;
                if ((rc = eavqa_attention_decode(dtype, B, H, Sk, hd, qkv, 3 * E, L.k_cache, E, L.v_cache, E, S_max, qkv + (size_t)E * es,
                                                 qkv + (size_t)2 * E * es, 3 * E, ctx, E, key_mask, ld_mask, scale, stream))) return rc;
            } else if (attn_from_partials) {
                // the decode attention sums the QKV partial sums itself (q, and the new K / V rows, which it appends to the cache)
                if ((rc = eavqa_gemm_splitk(dtype, rows, 3 * E, E, a, E, L.w_qkv, E, part, d.ks_qkv, stream))) return rc;
                if ((rc = eavqa_attention_decode_splitk(dtype, B, H, Sk, hd, part, d.ks_qkv, L.b_qkv, L.k_cache, E, L.v_cache, E, S_max, ctx, E,
                                                        key_mask, ld_mask, scale, stream))) return rc;
            } else {
                if ((rc = eavqa_gemm_splitk(dtype, rows, 3 * E, E, a, E, L.w_qkv, E, part, d.ks_qkv, stream))) return rc;
                // q -> qkv[:, :E]; k, v -> cache rows (sample m at row m * S_max + row0)
                if ((rc = eavqa_splitk_finish(dtype, rows, 3 * E, part, d.ks_qkv, L.b_qkv, EAVQA_ACT_NONE, nullptr, 0, 0, 3, qkv, 3 * E,
                                              static_cast<char*>(L.k_cache) + (size_t)row0 * E * es, (int64_t)S_max * E,
                                              static_cast<char*>(L.v_cache) + (size_t)row0 * E * es, (int64_t)S_max * E, stream))) return rc;
                if ((rc = eavqa_attention_fwd(dtype, B, H, Sq, Sk, hd, qkv, 3 * E, L.k_cache, E, L.v_cache, E, ctx, E, Sq, S_max, key_mask, ld_mask,
                                              nullptr, 1, scale, nullptr, stream))) return rc;
            }
            if ((rc = eavqa_gemm_splitk(dtype, rows, E, E, ctx, E, L.w_o, E, part, d.ks_o, stream))) return rc;
            // x1 = x + b_o + sum(partials); a = LN2(x1)
            if ((rc = eavqa_layernorm_splitk(dtype, rows, E, x, E, part, d.ks_o, L.b_o, x1, E, L.ln2_g, L.ln2_b, eps, a, E, stream))) return rc;
            if (fused) {
                if ((rc = eavqa_gemm(dtype, 1, 1, rows, F, E, a, E, L.w_fc1, E, f, F, 0, 1.f, L.b_fc1, act, nullptr, nullptr, 0, nullptr, 0,
                                     stream))) return rc;
            } else {
                if ((rc = eavqa_gemm_splitk(dtype, rows, F, E, a, E, L.w_fc1, E, part, d.ks_fc1, stream))) return rc;
                if ((rc = eavqa_splitk_finish(dtype, rows, F, part, d.ks_fc1, L.b_fc1, act, nullptr, 0, 0, 1, f, F, nullptr, 0, nullptr, 0, stream))) return rc;
            }
            if ((rc = eavqa_gemm_splitk(dtype, rows, E, F, f, F, L.w_fc2, F, part2, d.ks_fc2, stream))) return rc;
        }
        // x = x1 + b_fc2 + sum(last FFN-down partials)
        return eavqa_splitk_finish(dtype, rows, E, part2, d.ks_fc2, layers[n_layer - 1].b_fc2, EAVQA_ACT_NONE, x1, E, 1, 1, x, E, nullptr, 0,
                                   nullptr, 0, stream);
    }
    for (int l = 0; l < n_layer; ++l) {
        const eavqa_lm_layer_t& L = layers[l];
        if ((rc = eavqa_layernorm_fwd(dtype, 1, rows, E, x, E, L.ln1_g, L.ln1_b, eps, a, E, nullptr, nullptr, stream))) return rc;
        if ((rc = eavqa_gemm(dtype, 1, 1, rows, 3 * E, E, a, E, L.w_qkv, E, qkv, 3 * E, dtype == EAVQA_F32, 1.f, L.b_qkv, EAVQA_ACT_NONE,
                             nullptr, nullptr, 0, nullptr, 0, stream))) return rc;
        // append K / V of the new positions to the cache [B, S_max, E]
        if ((rc = eavqa_copy_rows(dtype, B, Sq, E, qkv + (size_t)E * es, 3 * E, Sq, L.k_cache, E, S_max, row0, stream))) return rc;
        if ((rc = eavqa_copy_rows(dtype, B, Sq, E, qkv + (size_t)2 * E * es, 3 * E, Sq, L.v_cache, E, S_max, row0, stream))) return rc;
        if ((rc = eavqa_attention_fwd(dtype, B, H, Sq, Sk, hd, qkv, 3 * E, L.k_cache, E, L.v_cache, E, ctx, E, Sq, S_max, key_mask, ld_mask,
                                      nullptr, 1, scale, nullptr, stream))) return rc;
        if ((rc = eavqa_gemm(dtype, 1, 1, rows, E, E, ctx, E, L.w_o, E, x1, E, 1, 1.f, L.b_o, EAVQA_ACT_NONE, nullptr, nullptr, 0, x, E,
                             stream))) return rc;
        if ((rc = eavqa_layernorm_fwd(dtype, 1, rows, E, x1, E, L.ln2_g, L.ln2_b, eps, a, E, nullptr, nullptr, stream))) return rc;
        if ((rc = eavqa_gemm(dtype, 1, 1, rows, F, E, a, E, L.w_fc1, E, f, F, dtype == EAVQA_F32, 1.f, L.b_fc1, act, nullptr, nullptr, 0,
                             nullptr, 0, stream))) return rc;
        if ((rc = eavqa_gemm(dtype, 1, 1, rows, E, F, f, F, L.w_fc2, F, x, E, 1, 1.f, L.b_fc2, EAVQA_ACT_NONE, nullptr, nullptr, 0, x1, E,
                             stream))) return rc;
    }
    return EAVQA_OK;
}

extern "C" int eavqa_lm_block_forward(int dtype, int n_layer, const eavqa_lm_layer_t* layers, int E, int H, int F, int act,
                                      float eps, int B, int Sq, int row0, int S_max, float* x, const int32_t* key_mask,
                                      int64_t ld_mask, void* workspace, int64_t workspace_bytes, void* stream) {
    return lm_block_forward_impl(dtype, n_layer, layers, E, H, F, act, eps, B, Sq, row0, S_max, x, key_mask, ld_mask, workspace, workspace_bytes,
                                 stream, 0);
}

extern "C" int eavqa_lm_block_forward_ex(int dtype, int n_layer, const eavqa_lm_layer_t* layers, int E, int H, int F, int act,
                                         float eps, int B, int Sq, int row0, int S_max, float* x, const int32_t* key_mask,
                                         int64_t ld_mask, void* workspace, int64_t workspace_bytes, void* stream, int route) {
    return lm_block_forward_impl(dtype, n_layer, layers, E, H, F, act, eps, B, Sq, row0, S_max, x, key_mask, ld_mask, workspace, workspace_bytes,
                                 stream, route);
}


// ------------------------------------------------------------------------------------------------ frozen LM held in e4m3 (BASELINE configs[4])
// The same driver for weights in e4m3 (models/lm.py Fp8Weight: bytes + one scale per tensor).  Every Linear is what the training / re-forward
// path computes - rows quantised to e4m3 with their own scale (eavqa_quantize_rows_fp8), product on the fp8 matrix cores, scales in the
// epilogue - so a cached generation reproduces `use_cache=False` (the reference's own loop, src/models/clipcap.py:414-419, through
// eavqa_gemm_fp8) up to fp32 summation order.  Decode steps stream HALF the weight bytes of the bf16 step: split-K over e4m3 weights
// (eavqa_gemm_fp8_splitk), the LayerNorm pass emits the quantised operand of the next projection itself (eavqa_layernorm_splitk_fp8); the
// attention output and the FFN activation take the stand-alone row quantiser (two more short kernels per layer).
namespace {
struct Fp8Plan { int ks_qkv, ks_o, ks_fc1, ks_fc2; bool ok; size_t part_bytes; };
inline Fp8Plan plan_decode_fp8(int rows, int Sq, int E, int F) {
    Fp8Plan d{0, 0, 0, 0, false, 0};
    if (Sq != 1 || rows > 64 || E % 16 || F % 16) return d;
    d.ks_qkv = eavqa_gemm_fp8_splitk_plan(rows, 3 * E, E);
    d.ks_o = eavqa_gemm_fp8_splitk_plan(rows, E, E);
    d.ks_fc1 = eavqa_gemm_fp8_splitk_plan(rows, F, E);
    d.ks_fc2 = eavqa_gemm_fp8_splitk_plan(rows, E, F);
    d.ok = d.ks_qkv > 0 && d.ks_o > 0 && d.ks_fc1 > 0 && d.ks_fc2 > 0;
    size_t a = (size_t)d.ks_qkv * rows * 3 * E, b = (size_t)d.ks_o * rows * E, c = (size_t)d.ks_fc1 * rows * F;
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    d.part_bytes = align_up(m * 4) + align_up((size_t)d.ks_fc2 * rows * E * 4);
    return d;
}
}

extern "C" int64_t eavqa_lm_block_fp8_workspace_bytes(int rows, int E, int F) {
    size_t b = (size_t)eavqa_lm_block_workspace_bytes(EAVQA_BF16, rows, E, F);       // the bf16 activations of a layer (+ bf16-route partials: unused)
    b += align_up((size_t)rows * (F > 3 * E ? F : 3 * E));                           // a quantised operand (e4m3 bytes)
    b += align_up((size_t)rows * 4);                                                  // its row scales
    const Fp8Plan d = plan_decode_fp8(rows, 1, E, F);
    if (d.ok) b += d.part_bytes;
    return (int64_t)b;
}

extern "C" int eavqa_lm_block_forward_fp8(int n_layer, const eavqa_lm_layer_t* layers, const eavqa_lm_layer_scales_t* scales, int E, int H, int F, int act,
                                          float eps, int B, int Sq, int row0, int S_max, float* x, const int32_t* key_mask, int64_t ld_mask,
                                          void* workspace, int64_t workspace_bytes, void* stream) {
    if (!layers || !scales || !x || !workspace || n_layer <= 0 || B <= 0 || Sq <= 0 || row0 < 0 || S_max < row0 + Sq) return EAVQA_E_ARG;
    if (E % H || E % 128 || F % 128) return EAVQA_E_SHAPE;                            // eavqa_gemm_fp8: K % 128
    const int rows = B * Sq, hd = E / H, Sk = row0 + Sq;
    if (workspace_bytes < eavqa_lm_block_fp8_workspace_bytes(rows, E, F)) return EAVQA_E_ARG;
    const int dtype = EAVQA_BF16;
    const size_t es = 2;
    char* w = static_cast<char*>(workspace);
    void* a = w;            w += align_up((size_t)rows * E * es);
    char* qkv = w;          w += align_up((size_t)rows * 3 * E * es);
    void* ctx = w;          w += align_up((size_t)rows * E * es);
    float* x1 = reinterpret_cast<float*>(w); w += align_up((size_t)rows * E * 4);
    void* f = w;            w += align_up((size_t)rows * F * es);
    {   // skip what eavqa_lm_block_workspace_bytes reserved for the bf16 route's partial sums
        const size_t used = (size_t)(w - static_cast<char*>(workspace));
        w = static_cast<char*>(workspace) + (size_t)eavqa_lm_block_workspace_bytes(EAVQA_BF16, rows, E, F);
        if ((size_t)(w - static_cast<char*>(workspace)) < used) return EAVQA_E_ARG;
    }
    void* aq = w;           w += align_up((size_t)rows * (F > 3 * E ? F : 3 * E));
    float* asc = reinterpret_cast<float*>(w); w += align_up((size_t)rows * 4);
    const float scale = 1.0f / sqrtf((float)hd);
    int rc;
    const Fp8Plan d = plan_decode_fp8(rows, Sq, E, F);
    const bool attn_from_partials = (hd % 8) == 0 && hd <= 128 && Sk <= 3584;
    if (d.ok && attn_from_partials) {
        float* part = reinterpret_cast<float*>(w);
        float* part2 = reinterpret_cast<float*>(w + (d.part_bytes - align_up((size_t)d.ks_fc2 * rows * E * 4)));
        for (int l = 0; l < n_layer; ++l) {
            const eavqa_lm_layer_t& L = layers[l];
            const eavqa_lm_layer_scales_t& S = scales[l];
            // x = x1 + b_fc2 + sum(FFN-down partials of the previous layer); aq = e4m3(LN1(x))
            if (l == 0) rc = eavqa_layernorm_splitk_fp8(rows, E, x, E, nullptr, 0, nullptr, nullptr, 0, L.ln1_g, L.ln1_b, eps, aq, E, asc, stream);
            else rc = eavqa_layernorm_splitk_fp8(rows, E, x1, E, part2, d.ks_fc2, layers[l - 1].b_fc2, x, E, L.ln1_g, L.ln1_b, eps, aq, E, asc, stream);
            if (rc) return rc;
            if ((rc = eavqa_gemm_fp8_splitk(rows, 3 * E, E, aq, E, asc, L.w_qkv, E, S.s_qkv, part, d.ks_qkv, stream))) return rc;
            if ((rc = eavqa_attention_decode_splitk(dtype, B, H, Sk, hd, part, d.ks_qkv, L.b_qkv, L.k_cache, E, L.v_cache, E, S_max, ctx, E,
                                                    key_mask, ld_mask, scale, stream))) return rc;
            if ((rc = eavqa_quantize_rows_fp8(dtype, rows, E, ctx, E, aq, E, asc, stream))) return rc;
            if ((rc = eavqa_gemm_fp8_splitk(rows, E, E, aq, E, asc, L.w_o, E, S.s_o, part, d.ks_o, stream))) return rc;
            // x1 = x + b_o + sum(partials); aq = e4m3(LN2(x1))
            if ((rc = eavqa_layernorm_splitk_fp8(rows, E, x, E, part, d.ks_o, L.b_o, x1, E, L.ln2_g, L.ln2_b, eps, aq, E, asc, stream))) return rc;
            if ((rc = eavqa_gemm_fp8_splitk(rows, F, E, aq, E, asc, L.w_fc1, E, S.s_fc1, part, d.ks_fc1, stream))) return rc;
            if ((rc = eavqa_splitk_finish(dtype, rows, F, part, d.ks_fc1, L.b_fc1, act, nullptr, 0, 0, 1, f, F, nullptr, 0, nullptr, 0, stream))) return rc;
            if ((rc = eavqa_quantize_rows_fp8(dtype, rows, F, f, F, aq, F, asc, stream))) return rc;
            if ((rc = eavqa_gemm_fp8_splitk(rows, E, F, aq, F, asc, L.w_fc2, F, S.s_fc2, part2, d.ks_fc2, stream))) return rc;
        }
        return eavqa_splitk_finish(dtype, rows, E, part2, d.ks_fc2, layers[n_layer - 1].b_fc2, EAVQA_ACT_NONE, x1, E, 1, 1, x, E, nullptr, 0,
                                   nullptr, 0, stream);
    }
    // prefill (or a shape without a plan): the calls of FrozenCausalLM.forward's `linear()` - quantise the rows, multiply on the fp8 cores
    for (int l = 0; l < n_layer; ++l) {
        const eavqa_lm_layer_t& L = layers[l];
        const eavqa_lm_layer_scales_t& S = scales[l];
        if ((rc = eavqa_layernorm_fwd(dtype, 1, rows, E, x, E, L.ln1_g, L.ln1_b, eps, a, E, nullptr, nullptr, stream))) return rc;
        if ((rc = eavqa_quantize_rows_fp8(dtype, rows, E, a, E, aq, E, asc, stream))) return rc;
        if ((rc = eavqa_gemm_fp8(rows, 3 * E, E, aq, E, asc, L.w_qkv, E, S.s_qkv, qkv, 3 * E, 0, 1.f, L.b_qkv, EAVQA_ACT_NONE, nullptr, nullptr, 0,
                                 nullptr, 0, stream, 0))) return rc;
        if ((rc = eavqa_copy_rows(dtype, B, Sq, E, qkv + (size_t)E * es, 3 * E, Sq, L.k_cache, E, S_max, row0, stream))) return rc;
        if ((rc = eavqa_copy_rows(dtype, B, Sq, E, qkv + (size_t)2 * E * es, 3 * E, Sq, L.v_cache, E, S_max, row0, stream))) return rc;
        if ((rc = eavqa_attention_fwd(dtype, B, H, Sq, Sk, hd, qkv, 3 * E, L.k_cache, E, L.v_cache, E, ctx, E, Sq, S_max, key_mask, ld_mask,
                                      nullptr, 1, scale, nullptr, stream))) return rc;
        if ((rc = eavqa_quantize_rows_fp8(dtype, rows, E, ctx, E, aq, E, asc, stream))) return rc;
        if ((rc = eavqa_gemm_fp8(rows, E, E, aq, E, asc, L.w_o, E, S.s_o, x1, E, 1, 1.f, L.b_o, EAVQA_ACT_NONE, nullptr, nullptr, 0, x, E, stream, 0))) return rc;
        if ((rc = eavqa_layernorm_fwd(dtype, 1, rows, E, x1, E, L.ln2_g, L.ln2_b, eps, a, E, nullptr, nullptr, stream))) return rc;
        if ((rc = eavqa_quantize_rows_fp8(dtype, rows, E, a, E, aq, E, asc, stream))) return rc;
        if ((rc = eavqa_gemm_fp8(rows, F, E, aq, E, asc, L.w_fc1, E, S.s_fc1, f, F, 0, 1.f, L.b_fc1, act, nullptr, nullptr, 0, nullptr, 0, stream, 0))) return rc;
        if ((rc = eavqa_quantize_rows_fp8(dtype, rows, F, f, F, aq, F, asc, stream))) return rc;
        if ((rc = eavqa_gemm_fp8(rows, E, F, aq, F, asc, L.w_fc2, F, S.s_fc2, x, E, 1, 1.f, L.b_fc2, EAVQA_ACT_NONE, nullptr, nullptr, 0, x1, E, stream, 0))) return rc;
    }
    return EAVQA_OK;
}
