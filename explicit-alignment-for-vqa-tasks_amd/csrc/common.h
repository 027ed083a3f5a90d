// Shared device/host helpers for the gfx950 kernels behind include/eavqa.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <atomic>
#include "eavqa.h"
#include "eavqa_test.h"      // the *_ex entry points are defined next to the public ones: both declarations carry default visibility

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define EAVQA_WAVE 64

#define EAVQA_LAUNCH_CHECK()                                   \
    do {                                                       \
        if (hipGetLastError() != hipSuccess) return EAVQA_E_LAUNCH; \
    } while (0)

static inline bool eavqa_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- scalar load/store through float, for kernels templated on the storage type ----
template <typename T> struct elem;
template <> struct elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    // 4 consecutive elements
    static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return (float)*p; }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = (bf16_t)v; }
    static __device__ __forceinline__ float4 ld4(const bf16_t* p) {
        bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
        return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
    }
    static __device__ __forceinline__ void st4(bf16_t* p, float4 v) {
        bf16x4 t;
        t[0] = (bf16_t)v.x; t[1] = (bf16_t)v.y; t[2] = (bf16_t)v.z; t[3] = (bf16_t)v.w;
        *reinterpret_cast<bf16x4*>(p) = t;
    }
};

// IEEE half: a STORAGE type of the frozen CLIP tower's residual stream only (EAVQA_F16: LayerNorm input / output, GEMM residual / C);
// no kernel multiplies in it.
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
template <> struct elem<f16_t> {
    static __device__ __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    static __device__ __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }
    static __device__ __forceinline__ float4 ld4(const f16_t* p) {
        f16x4 t = *reinterpret_cast<const f16x4*>(p);
        return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
    }
    static __device__ __forceinline__ void st4(f16_t* p, float4 v) {
        f16x4 t;
        t[0] = (f16_t)v.x; t[1] = (f16_t)v.y; t[2] = (f16_t)v.z; t[3] = (f16_t)v.w;
        *reinterpret_cast<f16x4*>(p) = t;
    }
};

// ---- activations (forward value and derivative w.r.t. the pre-activation) ----
// tanh through one v_exp + one v_rcp: tanh(x) = 1 - 2 / (exp(2x) + 1).  |error| <= ~2e-7 absolute (exp2-based
// __expf is within 2 ulp, the form is stable at both tails: exp -> inf gives 1, exp -> 0 gives -1).  The libm
// tanhf costs ~10x more VALU time, which showed up as +60 % on the gelu_new GEMM epilogues.
// The reciprocal is v_rcp_f32 (1 ulp) through the builtin: `__fdividef` is a plain `/` on this toolchain and compiles to the IEEE
// division sequence (v_div_scale x2, v_rcp, four FMAs, v_div_fmas, v_div_fixup: ~12 VALU instructions per element) - round 3 found
// it made the QuickGELU epilogue of the CLIP tower's FFN-up cost +35 % of the whole GEMM (168 M sigmoid evaluations per launch).
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __expf(2.f * x);
    return 1.f - 2.f * fast_rcp(e + 1.f);
}
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.f + __expf(-x)); }

__device__ __forceinline__ float act_fwd(int act, float x) {
    switch (act) {
        case EAVQA_ACT_TANH: return fast_tanh(x);
        case EAVQA_ACT_RELU: return x > 0.f ? x : 0.f;
        case EAVQA_ACT_GELU_NEW: {
            const float c = 0.7978845608028654f;  // sqrt(2/pi)
            return 0.5f * x * (1.f + fast_tanh(c * (x + 0.044715f * x * x * x)));
        }
        case EAVQA_ACT_QUICK_GELU: return x * fast_sigmoid(1.702f * x);
        default: return x;
    }
}
__device__ __forceinline__ float act_bwd(int act, float x) {
    switch (act) {
        case EAVQA_ACT_TANH: { float t = fast_tanh(x); return 1.f - t * t; }
        case EAVQA_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case EAVQA_ACT_GELU_NEW: {
            const float c = 0.7978845608028654f;
            float inner = c * (x + 0.044715f * x * x * x);
            float t = fast_tanh(inner);
            float dinner = c * (1.f + 3.f * 0.044715f * x * x);
            return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * dinner;
        }
        case EAVQA_ACT_QUICK_GELU: {
            float s = fast_sigmoid(1.702f * x);
            return s * (1.f + 1.702f * x * (1.f - s));
        }
        default: return 1.f;
    }
}

// ---- wave / block reductions (wave = 64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
