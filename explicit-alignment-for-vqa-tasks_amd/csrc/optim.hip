// Fused AdamW over one flat float32 parameter buffer (eavqa_adamw in include/eavqa.h).
// HBM-bound: per element 16 B read (param, grad, m, v) + 12 B written (+ 2 B bf16 shadow).
#include "common.h"

namespace {

template <typename S, bool HAS_SHADOW>
__global__ __launch_bounds__(256) void adamw_kernel(int64_t n4, int64_t n, float* param, const float* grad, float* m, float* v,
                                                    float lr, float beta1, float beta2, float eps, float decay_mul,
                                                    float inv_bc1, float inv_sqrt_bc2, float grad_scale, S* shadow) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto update = [&](float4& p, const float4 g, float4& mm, float4& vv) {
        float pa[4] = {p.x, p.y, p.z, p.w}, ga[4] = {g.x, g.y, g.z, g.w};
        float ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gj = ga[j] * grad_scale;
            pa[j] *= decay_mul;                                   // p *= 1 - lr*wd  (decoupled decay first)
            ma[j] = beta1 * ma[j] + (1.f - beta1) * gj;
            va[j] = beta2 * va[j] + (1.f - beta2) * gj * gj;
            const float denom = sqrtf(va[j]) * inv_sqrt_bc2 + eps;
            pa[j] -= (lr * inv_bc1) * (ma[j] / denom);
        }
        p = make_float4(pa[0], pa[1], pa[2], pa[3]);
        mm = make_float4(ma[0], ma[1], ma[2], ma[3]);
        vv = make_float4(va[0], va[1], va[2], va[3]);
    };
    // two grid strides per iteration: eight 16-byte loads in flight per thread before the first use (the update is element-wise, so
    // the order in which elements are visited does not change any result)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
        const int64_t i2 = i + stride;
        const bool two = i2 < n4;
        const int64_t j2 = two ? i2 : i;
        float4 p0 = reinterpret_cast<float4*>(param)[i], p1 = reinterpret_cast<float4*>(param)[j2];
        const float4 g0 = reinterpret_cast<const float4*>(grad)[i], g1 = reinterpret_cast<const float4*>(grad)[j2];
        float4 m0 = reinterpret_cast<float4*>(m)[i], m1 = reinterpret_cast<float4*>(m)[j2];
        float4 v0 = reinterpret_cast<float4*>(v)[i], v1 = reinterpret_cast<float4*>(v)[j2];
        update(p0, g0, m0, v0);
        reinterpret_cast<float4*>(param)[i] = p0;
        reinterpret_cast<float4*>(m)[i] = m0;
        reinterpret_cast<float4*>(v)[i] = v0;
        if (HAS_SHADOW) elem<S>::st4(shadow + 4 * i, p0);
        if (two) {
            update(p1, g1, m1, v1);
            reinterpret_cast<float4*>(param)[i2] = p1;
            reinterpret_cast<float4*>(m)[i2] = m1;
            reinterpret_cast<float4*>(v)[i2] = v1;
            if (HAS_SHADOW) elem<S>::st4(shadow + 4 * i2, p1);
        }
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            const float gj = grad[i] * grad_scale;
            float pj = param[i] * decay_mul;
            const float mj = beta1 * m[i] + (1.f - beta1) * gj;
            const float vj = beta2 * v[i] + (1.f - beta2) * gj * gj;
            pj -= (lr * inv_bc1) * (mj / (sqrtf(vj) * inv_sqrt_bc2 + eps));
            param[i] = pj; m[i] = mj; v[i] = vj;
            if (HAS_SHADOW) elem<S>::st(shadow + i, pj);
        }
    }
}

}  // namespace

extern "C" int eavqa_adamw(int64_t n, float* param, const float* grad, float* m, float* v, int step, float lr, float beta1,
                           float beta2, float eps, float weight_decay, float grad_scale, int shadow_dtype, void* shadow,
                           void* stream) {
    if (n <= 0 || !param || !grad || !m || !v || step < 1) return EAVQA_E_ARG;
    if (!eavqa_aligned16(param) || !eavqa_aligned16(grad) || !eavqa_aligned16(m) || !eavqa_aligned16(v)) return EAVQA_E_ALIGN;
    // bias corrections in double on the host, as torch does with Python floats
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float inv_bc1 = (float)(1.0 / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const float decay_mul = (float)(1.0 - (double)lr * (double)weight_decay);
    const int64_t n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define EAVQA_ADAMW(S, HAS)                                                                                          \
    hipLaunchKernelGGL((adamw_kernel<S, HAS>), dim3(blocks), dim3(256), 0, s, n4, n, param, grad, m, v, lr, beta1, \
                       beta2, eps, decay_mul, inv_bc1, inv_sqrt_bc2, grad_scale, reinterpret_cast<S*>(shadow))
    if (!shadow) EAVQA_ADAMW(float, false);
    else if (shadow_dtype == EAVQA_BF16) EAVQA_ADAMW(bf16_t, true);
    else if (shadow_dtype == EAVQA_F32) EAVQA_ADAMW(float, true);
    else return EAVQA_E_DTYPE;
#undef EAVQA_ADAMW
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}
