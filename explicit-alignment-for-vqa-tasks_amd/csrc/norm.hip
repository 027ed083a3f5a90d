// LayerNorm forward / backward (eavqa_layernorm_fwd / _bwd in include/eavqa.h).
// HBM-bound: one wavefront (64 lanes) owns one row, the row lives in registers between the
// statistics passes, every global access is a 16-byte (fp32) or 8-byte (bf16) vector.
// Algorithmic bytes per row: fwd = cols*(sizeof x + sizeof y); bwd = cols*(x + dy + dres + dx).
#include "common.h"

namespace {

constexpr int LN_MAX_V4 = 32;  // float4 per lane: cols <= 64 * 4 * 32 = 8192
constexpr int LN_WAVES = 4;    // rows per 256-thread block

// kind: 0 = T, 1 = float32, 2 = bfloat16, 3 = half (wave-uniform)
template <typename T>
__device__ __forceinline__ float4 ldrow4(const void* base, int64_t off, int kind) {
    if (kind == 1) return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
    if (kind == 2) return elem<bf16_t>::ld4(reinterpret_cast<const bf16_t*>(base) + off);
    if (kind == 3) return elem<f16_t>::ld4(reinterpret_cast<const f16_t*>(base) + off);
    return elem<T>::ld4(reinterpret_cast<const T*>(base) + off);
}

template <typename T> __device__ __forceinline__ float4 round_to(const float4 o) {
    return make_float4((float)(T)o.x, (float)(T)o.y, (float)(T)o.z, (float)(T)o.w);
}

// The row-wise e4m3 quantisation of eavqa_quantize_rows_fp8 on a row that one wave holds in registers (lane l: float4 l, l + 64, ...;
// entries beyond nv are zeros): scale = amax / 448 (1 for an all-zero row), q = cvt(v / scale) - the same arithmetic, so the bytes and the
// scale are those of the separate kernel.  (eavqa_layernorm_fwd_fp8 / _bwd_fp8: the quantiser fused into its row-complete producers.)
template <int NV>
__device__ __forceinline__ void quantize_row_e4m3(const float4 (&v)[NV], int nv, int lane, unsigned char* q, float* scale_out) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < nv) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
    amax = wave_max(amax);
    const float scale = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
    const float inv = 1.f / scale;
    if (lane == 0) *scale_out = scale;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].x * inv, v[i].y * inv, 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].z * inv, v[i].w * inv, pk, true);
            *reinterpret_cast<int*>(q + 4 * c) = pk;
        }
    }
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(int x_f32, int rows, int cols, const void* x, int64_t ldx,
                                                     const float* gamma, const float* beta, float eps,
                                                     T* y, int64_t ldy, float* mean, float* rstd,
                                                     unsigned char* yq = nullptr, int64_t ldq = 0, float* yq_scale = nullptr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = cols >> 2;
    float4 v[NV], gm[NV], bt[NV];
    float s = 0.f;
    // gamma / beta are fetched together with the row: one memory round trip instead of two dependent ones
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            v[i] = ldrow4<T>(x, (int64_t)row * ldx + 4 * c, x_f32);
            gm[i] = gamma ? *reinterpret_cast<const float4*>(gamma + 4 * c) : make_float4(1.f, 1.f, 1.f, 1.f);
            bt[i] = beta ? *reinterpret_cast<const float4*>(beta + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float4 o;
            o.x = (v[i].x - mu) * rs * gm[i].x + bt[i].x;
            o.y = (v[i].y - mu) * rs * gm[i].y + bt[i].y;
            o.z = (v[i].z - mu) * rs * gm[i].z + bt[i].z;
            o.w = (v[i].w - mu) * rs * gm[i].w + bt[i].w;
            if (y) elem<T>::st4(y + (int64_t)row * ldy + 4 * c, o);
            if (yq) v[i] = round_to<T>(o);            // what eavqa_quantize_rows_fp8 would read back: the values in the storage type
        }
    }
    if (yq) quantize_row_e4m3<NV>(v, nv, lane, yq + (int64_t)row * ldq, yq_scale + row);
}

// dx = dres + rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat))
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(int x_f32, int rows, int cols, const void* x, int64_t ldx,
                                                     const T* dy, int64_t lddy, const float* gamma,
                                                     const float* mean, const float* rstd, const float* dres,
                                                     float* dx, int64_t lddx, float* dgamma, float* dbeta,
                                                     T* dx_lowp, int64_t ld_lowp,
                                                     unsigned char* dxq = nullptr, int64_t ldq = 0, float* dxq_scale = nullptr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = cols >> 2;
    const float mu = mean[row], rs = rstd[row];
    float4 xh[NV], gd[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float4 xv = ldrow4<T>(x, (int64_t)row * ldx + 4 * c, x_f32);
            float4 d = elem<T>::ld4(dy + (int64_t)row * lddy + 4 * c);
            float4 g = gamma ? *reinterpret_cast<const float4*>(gamma + 4 * c) : make_float4(1.f, 1.f, 1.f, 1.f);
            xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            gd[i] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
            s1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
            s2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
        } else {
            xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            gd[i] = xh[i];
        }
    }
    const float m1 = wave_sum(s1) / (float)cols;
    const float m2 = wave_sum(s2) / (float)cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float4 r = dres ? *reinterpret_cast<const float4*>(dres + (int64_t)row * lddx + 4 * c)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o;
            o.x = r.x + rs * (gd[i].x - m1 - xh[i].x * m2);
            o.y = r.y + rs * (gd[i].y - m1 - xh[i].y * m2);
            o.z = r.z + rs * (gd[i].z - m1 - xh[i].z * m2);
            o.w = r.w + rs * (gd[i].w - m1 - xh[i].w * m2);
            *reinterpret_cast<float4*>(dx + (int64_t)row * lddx + 4 * c) = o;
            if (dx_lowp) elem<T>::st4(dx_lowp + (int64_t)row * ld_lowp + 4 * c, o);
            if (dxq) gd[i] = round_to<T>(o);          // (gd is dead from here on)
        }
    }
    if (dxq) quantize_row_e4m3<NV>(gd, nv, lane, dxq + (int64_t)row * ldq, dxq_scale + row);
}

// dgamma[c] += sum_r dy[r,c] * xhat[r,c], dbeta[c] += sum_r dy[r,c] (the mapper's LayerNorms only: the LM is frozen).  A block owns 64
// columns x 64 rows, sums them in registers / LDS and issues ONE atomic per column: rows / 64 atomics per column instead of one per
// element (2 048 x 4 096 elements cost 440-540 us through the atomic units, MI355X_MICROARCH.md "Global float atomics").
template <typename T>
__global__ __launch_bounds__(256) void ln_dparam_kernel(int x_f32, int rows, int cols, const void* x, int64_t ldx, const T* dy, int64_t lddy,
                                                        const float* mean, const float* rstd, float* dgamma, float* dbeta) {
    __shared__ float sg[4][64], sb[4][64];
    const int tid = threadIdx.x, cl = tid & 63, rs = tid >> 6;
    const int col = blockIdx.x * 64 + cl;
    float g = 0.f, b = 0.f;
    if (col < cols) {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int row = blockIdx.y * 64 + rs + 4 * i;
            if (row < rows) {
                const int64_t off = (int64_t)row * ldx + col;
                const float xv = x_f32 ? reinterpret_cast<const float*>(x)[off] : elem<T>::ld(reinterpret_cast<const T*>(x) + off);
                const float d = elem<T>::ld(dy + (int64_t)row * lddy + col);
                g += d * (xv - mean[row]) * rstd[row];
                b += d;
            }
        }
    }
    sg[rs][cl] = g; sb[rs][cl] = b;
    __syncthreads();
    if (tid < 64 && col < cols) {
        if (dgamma) atomicAdd(dgamma + col, (sg[0][cl] + sg[1][cl]) + (sg[2][cl] + sg[3][cl]));
        if (dbeta) atomicAdd(dbeta + col, (sb[0][cl] + sb[1][cl]) + (sb[2][cl] + sb[3][cl]));
    }
}

template <typename T>
int ln_fwd_dispatch(int x_f32, int rows, int cols, const void* x, int64_t ldx, const float* gamma, const float* beta,
                    float eps, void* y, int64_t ldy, float* mean, float* rstd, hipStream_t s, void* yq_ = nullptr, int64_t ldq = 0,
                    float* yq_scale = nullptr) {
    unsigned char* yq = reinterpret_cast<unsigned char*>(yq_);
    const int nv = (cols / 4 + 63) / 64;
    dim3 grid((rows + LN_WAVES - 1) / LN_WAVES), block(256);
#define EAVQA_LN_FWD(NV)                                                                                         \
    hipLaunchKernelGGL((ln_fwd_kernel<T, NV>), grid, block, 0, s, x_f32, rows, cols, x, ldx, gamma, beta, eps, \
                       reinterpret_cast<T*>(y), ldy, mean, rstd, yq, ldq, yq_scale)
    if (nv <= 2) EAVQA_LN_FWD(2);
    else if (nv <= 4) EAVQA_LN_FWD(4);
    else if (nv <= 8) EAVQA_LN_FWD(8);
    else if (nv <= 16) EAVQA_LN_FWD(16);
    else EAVQA_LN_FWD(32);
#undef EAVQA_LN_FWD
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

template <typename T>
int ln_bwd_dispatch(int x_f32, int rows, int cols, const void* x, int64_t ldx, const void* dy, int64_t lddy,
                    const float* gamma, const float* mean, const float* rstd, const float* dres, float* dx,
                    int64_t lddx, float* dgamma, float* dbeta, void* dx_lowp, int64_t ld_lowp, hipStream_t s, void* dxq_ = nullptr,
                    int64_t ldq = 0, float* dxq_scale = nullptr) {
    unsigned char* dxq = reinterpret_cast<unsigned char*>(dxq_);
    const int nv = (cols / 4 + 63) / 64;
    dim3 grid((rows + LN_WAVES - 1) / LN_WAVES), block(256);
#define EAVQA_LN_BWD(NV)                                                                                   \
    hipLaunchKernelGGL((ln_bwd_kernel<T, NV>), grid, block, 0, s, x_f32, rows, cols, x, ldx,               \
                       reinterpret_cast<const T*>(dy), lddy, gamma, mean, rstd, dres, dx, lddx, dgamma, dbeta,  \
                       reinterpret_cast<T*>(dx_lowp), ld_lowp, dxq, ldq, dxq_scale)
    if (nv <= 2) EAVQA_LN_BWD(2);
    else if (nv <= 4) EAVQA_LN_BWD(4);
    else if (nv <= 8) EAVQA_LN_BWD(8);
    else if (nv <= 16) EAVQA_LN_BWD(16);
    else return EAVQA_E_SHAPE;  // 2 x 32 float4 per lane would spill; cols <= 4096 in backward
#undef EAVQA_LN_BWD
    if (dgamma || dbeta)
        hipLaunchKernelGGL((ln_dparam_kernel<T>), dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, s, x_f32, rows, cols, x, ldx,
                           reinterpret_cast<const T*>(dy), lddy, mean, rstd, dgamma, dbeta);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

// T5LayerNorm (HF:models/t5/modeling_t5.py:50-72): y = w * x * rsqrt(mean(x^2) + eps) - no mean subtraction, no bias.  Same row-per-wave
// layout as ln_fwd_kernel.  Backward (frozen weight): dx = dres + rstd * (g - xhat * mean(g * xhat)), g = w * dy, xhat = x * rstd.
template <typename T, int NV>
__global__ __launch_bounds__(256) void rms_fwd_kernel(int x_kind, int rows, int cols, const void* x, int64_t ldx, const float* gamma, float eps,
                                                      T* y, int64_t ldy, float* rstd) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = cols >> 2;
    float4 v[NV], gm[NV];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            v[i] = ldrow4<T>(x, (int64_t)row * ldx + 4 * c, x_kind);
            gm[i] = gamma ? *reinterpret_cast<const float4*>(gamma + 4 * c) : make_float4(1.f, 1.f, 1.f, 1.f);
            q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
    if (lane == 0 && rstd) rstd[row] = rs;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) elem<T>::st4(y + (int64_t)row * ldy + 4 * c, make_float4(v[i].x * rs * gm[i].x, v[i].y * rs * gm[i].y, v[i].z * rs * gm[i].z, v[i].w * rs * gm[i].w));
    }
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void rms_bwd_kernel(int x_kind, int rows, int cols, const void* x, int64_t ldx, const T* dy, int64_t lddy,
                                                      const float* gamma, const float* rstd, const float* dres, float* dx, int64_t lddx,
                                                      T* dx_lowp, int64_t ld_lowp) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = cols >> 2;
    const float rs = rstd[row];
    float4 xh[NV], gd[NV];
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 xv = ldrow4<T>(x, (int64_t)row * ldx + 4 * c, x_kind);
            const float4 d = elem<T>::ld4(dy + (int64_t)row * lddy + 4 * c);
            const float4 g = gamma ? *reinterpret_cast<const float4*>(gamma + 4 * c) : make_float4(1.f, 1.f, 1.f, 1.f);
            xh[i] = make_float4(xv.x * rs, xv.y * rs, xv.z * rs, xv.w * rs);
            gd[i] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
            s2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
        } else {
            xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            gd[i] = xh[i];
        }
    }
    const float m2 = wave_sum(s2) / (float)cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 r = dres ? *reinterpret_cast<const float4*>(dres + (int64_t)row * lddx + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o;
            o.x = r.x + rs * (gd[i].x - xh[i].x * m2);
            o.y = r.y + rs * (gd[i].y - xh[i].y * m2);
            o.z = r.z + rs * (gd[i].z - xh[i].z * m2);
            o.w = r.w + rs * (gd[i].w - xh[i].w * m2);
            *reinterpret_cast<float4*>(dx + (int64_t)row * lddx + 4 * c) = o;
            if (dx_lowp) elem<T>::st4(dx_lowp + (int64_t)row * ld_lowp + 4 * c, o);
        }
    }
}

template <typename T>
int rms_fwd_dispatch(int x_kind, int rows, int cols, const void* x, int64_t ldx, const float* gamma, float eps, void* y, int64_t ldy, float* rstd,
                     hipStream_t s) {
    const int nv = (cols / 4 + 63) / 64;
    dim3 grid((rows + LN_WAVES - 1) / LN_WAVES), block(256);
#define EAVQA_RMS_FWD(NV) hipLaunchKernelGGL((rms_fwd_kernel<T, NV>), grid, block, 0, s, x_kind, rows, cols, x, ldx, gamma, eps, reinterpret_cast<T*>(y), ldy, rstd)
    if (nv <= 2) EAVQA_RMS_FWD(2); else if (nv <= 4) EAVQA_RMS_FWD(4); else if (nv <= 8) EAVQA_RMS_FWD(8); else if (nv <= 16) EAVQA_RMS_FWD(16); else EAVQA_RMS_FWD(32);
#undef EAVQA_RMS_FWD
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

template <typename T>
int rms_bwd_dispatch(int x_kind, int rows, int cols, const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* gamma, const float* rstd,
                     const float* dres, float* dx, int64_t lddx, void* dx_lowp, int64_t ld_lowp, hipStream_t s) {
    const int nv = (cols / 4 + 63) / 64;
    dim3 grid((rows + LN_WAVES - 1) / LN_WAVES), block(256);
#define EAVQA_RMS_BWD(NV) hipLaunchKernelGGL((rms_bwd_kernel<T, NV>), grid, block, 0, s, x_kind, rows, cols, x, ldx, reinterpret_cast<const T*>(dy), lddy, \
                                             gamma, rstd, dres, dx, lddx, reinterpret_cast<T*>(dx_lowp), ld_lowp)
    if (nv <= 2) EAVQA_RMS_BWD(2); else if (nv <= 4) EAVQA_RMS_BWD(4); else if (nv <= 8) EAVQA_RMS_BWD(8); else if (nv <= 16) EAVQA_RMS_BWD(16);
    else return EAVQA_E_SHAPE;
#undef EAVQA_RMS_BWD
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

}  // namespace

extern "C" int eavqa_layernorm_fwd(int dtype, int x_kind, int rows, int cols, const void* x, int64_t ldx,
                                   const float* gamma, const float* beta, float eps, void* y, int64_t ldy,
                                   float* mean, float* rstd, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0) return EAVQA_E_ARG;
    if (cols % 4 || cols > 64 * 4 * LN_MAX_V4) return EAVQA_E_SHAPE;
    if (ldx % 4 || ldy % 4) return EAVQA_E_ALIGN;
    if (x_kind < 0 || x_kind > 3) return EAVQA_E_DTYPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_F32) return ln_fwd_dispatch<float>(x_kind == 0 ? 1 : x_kind, rows, cols, x, ldx, gamma, beta, eps, y, ldy, mean, rstd, s);
    if (dtype == EAVQA_BF16) return ln_fwd_dispatch<bf16_t>(x_kind, rows, cols, x, ldx, gamma, beta, eps, y, ldy, mean, rstd, s);
    if (dtype == EAVQA_F16) return ln_fwd_dispatch<f16_t>(x_kind, rows, cols, x, ldx, gamma, beta, eps, y, ldy, mean, rstd, s);
    return EAVQA_E_DTYPE;
}

extern "C" int eavqa_layernorm_bwd(int dtype, int x_f32, int rows, int cols, const void* x, int64_t ldx,
                                   const void* dy, int64_t lddy, const float* gamma, const float* mean,
                                   const float* rstd, const float* dres, float* dx, int64_t lddx,
                                   float* dgamma, float* dbeta, void* dx_lowp, int64_t ld_lowp, void* stream) {
    if (!x || !dy || !dx || !mean || !rstd || rows <= 0 || cols <= 0) return EAVQA_E_ARG;
    if (cols % 4) return EAVQA_E_SHAPE;
    if (ldx % 4 || lddy % 4 || lddx % 4 || (dx_lowp && ld_lowp % 4)) return EAVQA_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_F32)
        return ln_bwd_dispatch<float>(1, rows, cols, x, ldx, dy, lddy, gamma, mean, rstd, dres, dx, lddx, dgamma, dbeta, dx_lowp, ld_lowp, s);
    if (dtype == EAVQA_BF16)
        return ln_bwd_dispatch<bf16_t>(x_f32, rows, cols, x, ldx, dy, lddy, gamma, mean, rstd, dres, dx, lddx, dgamma, dbeta, dx_lowp, ld_lowp, s);
    return EAVQA_E_DTYPE;
}

extern "C" int eavqa_layernorm_fwd_fp8(int x_kind, int rows, int cols, const void* x, int64_t ldx, const float* gamma, const float* beta,
                                       float eps, void* yq, int64_t ldq, float* row_scale, float* mean, float* rstd, void* stream) {
    if (!x || !yq || !row_scale || rows <= 0 || cols <= 0) return EAVQA_E_ARG;
    if (cols % 4 || cols > 64 * 4 * LN_MAX_V4) return EAVQA_E_SHAPE;
    if (ldx % 4 || ldq % 4) return EAVQA_E_ALIGN;
    if (x_kind < 1 || x_kind > 3) return EAVQA_E_DTYPE;
    return ln_fwd_dispatch<bf16_t>(x_kind, rows, cols, x, ldx, gamma, beta, eps, nullptr, 0, mean, rstd, reinterpret_cast<hipStream_t>(stream),
                                   yq, ldq, row_scale);
}

extern "C" int eavqa_layernorm_bwd_fp8(int x_f32, int rows, int cols, const void* x, int64_t ldx, const void* dy, int64_t lddy,
                                       const float* gamma, const float* mean, const float* rstd, const float* dres, float* dx, int64_t lddx,
                                       void* dxq, int64_t ldq, float* row_scale, void* stream) {
    if (!x || !dy || !dx || !mean || !rstd || !dxq || !row_scale || rows <= 0 || cols <= 0) return EAVQA_E_ARG;
    if (cols % 4) return EAVQA_E_SHAPE;
    if (ldx % 4 || lddy % 4 || lddx % 4 || ldq % 4) return EAVQA_E_ALIGN;
    return ln_bwd_dispatch<bf16_t>(x_f32, rows, cols, x, ldx, dy, lddy, gamma, mean, rstd, dres, dx, lddx, nullptr, nullptr, nullptr, 0,
                                   reinterpret_cast<hipStream_t>(stream), dxq, ldq, row_scale);
}

extern "C" int eavqa_rmsnorm_fwd(int dtype, int x_kind, int rows, int cols, const void* x, int64_t ldx, const float* gamma, float eps,
                                 void* y, int64_t ldy, float* rstd, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0) return EAVQA_E_ARG;
    if (cols % 4 || cols > 64 * 4 * LN_MAX_V4) return EAVQA_E_SHAPE;
    if (ldx % 4 || ldy % 4) return EAVQA_E_ALIGN;
    if (x_kind < 0 || x_kind > 3) return EAVQA_E_DTYPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_F32) return rms_fwd_dispatch<float>(x_kind == 0 ? 1 : x_kind, rows, cols, x, ldx, gamma, eps, y, ldy, rstd, s);
    if (dtype == EAVQA_BF16) return rms_fwd_dispatch<bf16_t>(x_kind, rows, cols, x, ldx, gamma, eps, y, ldy, rstd, s);
    return EAVQA_E_DTYPE;
}

extern "C" int eavqa_rmsnorm_bwd(int dtype, int x_kind, int rows, int cols, const void* x, int64_t ldx, const void* dy, int64_t lddy,
                                 const float* gamma, const float* rstd, const float* dres, float* dx, int64_t lddx, void* dx_lowp,
                                 int64_t ld_lowp, void* stream) {
    if (!x || !dy || !dx || !rstd || rows <= 0 || cols <= 0) return EAVQA_E_ARG;
    if (cols % 4) return EAVQA_E_SHAPE;
    if (ldx % 4 || lddy % 4 || lddx % 4 || (dx_lowp && ld_lowp % 4)) return EAVQA_E_ALIGN;
    if (x_kind < 0 || x_kind > 3) return EAVQA_E_DTYPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_F32) return rms_bwd_dispatch<float>(x_kind == 0 ? 1 : x_kind, rows, cols, x, ldx, dy, lddy, gamma, rstd, dres, dx, lddx, dx_lowp, ld_lowp, s);
    if (dtype == EAVQA_BF16) return rms_bwd_dispatch<bf16_t>(x_kind, rows, cols, x, ldx, dy, lddy, gamma, rstd, dres, dx, lddx, dx_lowp, ld_lowp, s);
    return EAVQA_E_DTYPE;
}
