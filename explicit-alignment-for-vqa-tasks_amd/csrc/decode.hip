// Decode-step kernels (one new token per sample, M = batch <= 64 rows): the step is a weight-streaming problem, HBM bound
// on the 12 E^2 weights of every layer, and a chain of short dependent kernels.  Three pieces (eavqa.h):
//
//   eavqa_gemm_splitk      P[s][m][n] = sum_{k in slice s} A[m,k] B[n,k]        (bf16 in, fp32 partial sums out)
//   eavqa_splitk_finish    out = act(sum_s P[s] + bias) (+ residual), columns cut into up to 3 destination segments
//                          (q | k-cache row | v-cache row: the K/V append is the finish of the QKV projection)
//   eavqa_layernorm_splitk x = x_in + bias + sum_s P[s];  y = LayerNorm(x)      (residual add + finish + LN in one pass)
//
// Why split K over workgroups: with M <= 64 every workgroup needs the whole activation slab A, and the first skinny
// kernel (16 columns x all of K per workgroup) pulled 2 bytes of A through the CU's L1 for every byte of weights: the
// L2 -> CU path (about 55 GB/s per CU), not HBM, set its 2.4 TB/s.  Here a workgroup owns 128 columns x one K slice: its
// A slice is staged ONCE in LDS (LDS-DMA, swizzled 64-byte rows as in the tiled GEMMs) and shared by the eight waves, each
// of which streams its own 16 weight rows straight into VGPRs through a rolling window of 8 / 16 loads in flight.  A-bytes
// per weight byte: 16 MF / 128 = 0.25 (MF = 2).  The partial sums (ks x M x N fp32, 5-20 % of the weight bytes) are summed
// in a fixed order by the consumer, so the result does not depend on scheduling.  A pure-load kernel with the same access
// pattern reads these matrices at 4.3-4.7 TB/s (tools/hbm_probe.hip; 5.7 TB/s for 250+ MB): that is the floor this kernel is
// measured against, not 8 TB/s.
#include "common.h"

namespace {

__device__ __forceinline__ int fswz(int row, int kc) { return row * 64 + ((kc ^ ((-(row >> 2)) & 3)) << 4); }


// MF: 16-row fragments of A (M <= 16 MF); NF: 16-column fragments per wave; NW: waves per workgroup (columns per workgroup =
// 16 NF NW: the A slice is staged once per workgroup, so A bytes per weight byte = 16 MF / (16 NF NW)); U: k-steps of weight loads in
// flight per wave.
template <int MF, int NF, int NW, int U>
__global__ __launch_bounds__(64 * NW) void gemm_bf16_splitk_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                   const bf16_t* __restrict__ B, int64_t ldb,
                                                                   float* __restrict__ P, int M, int N, int KS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 16 * MF * 64;                      // bytes of one [16 MF rows][32 k] A tile
    constexpr int COLS = 16 * NF * NW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * COLS + wave * 16 * NF, slice = blockIdx.y, k0 = slice * KS;
    const int nsteps = KS >> 5;

    // ---- this wave's 16 NF weight rows: the first U k-steps go out before anything else (they are the HBM round trip that
    //      bounds the workgroup's life; the A slice below comes from L2)
    const int x = lane & 15, g = lane >> 4;
    const bf16_t* bp[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) bp[j] = B + (int64_t)min(n0 + 16 * j + x, N - 1) * ldb + k0 + 8 * g;
    bf16x8 bf[U][NF];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NF; ++j) bf[u][j] = *reinterpret_cast<const bf16x8*>(bp[j] + 32 * min(u, nsteps - 1));

    // ---- stage the A slice: chunk c -> tile c / (64 MF), row (c % (64 MF)) >> 2, physical slot c & 3
    const int total = nsteps * 64 * MF;
    for (int base = wave * 64; base < total; base += 64 * NW) {
        const int c = base + lane;
        if (c < total) {
            const int t = c / (64 * MF), within = c % (64 * MF);
            const int row = within >> 2, pc = within & 3;
            const bf16_t* src = A + (int64_t)min(row, M - 1) * lda + k0 + t * 32 + ((pc ^ ((-(row >> 2)) & 3)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + base * 16), 16, 0, 0);
        }
    }
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int a_off = fswz(x, g);
    __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00);     // vmcnt(0): A slice (and the first weights) have landed
    __syncthreads();

    // rolling window: the registers of k-step s are refilled with k-step s + U as soon as its MFMAs are issued, so U loads
    // per fragment column stay in flight for the whole slice
    for (int s0 = 0; s0 < nsteps; s0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bf16x8 w[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) w[j] = bf[u][j];
            if (s0 + U + u < nsteps) {
#pragma unroll
                for (int j = 0; j < NF; ++j) bf[u][j] = *reinterpret_cast<const bf16x8*>(bp[j] + 32 * (s0 + U + u));
            }
            if (s0 + u < nsteps) {
                const char* tile = smem + (s0 + u) * TILE;
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(tile + a_off + i * 1024);
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, w[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }

#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = n0 + 16 * j + x;
        if (n < N) {
            float* out = P + (int64_t)slice * M * N + n;
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * i + 4 * g + r;
                    if (m < M) out[(int64_t)m * N] = acc[i][j][r];
                }
        }
    }
}

// The same kernel for a frozen LM held in e4m3 (BASELINE configs[4]): A = the activation rows quantised to e4m3 with one scale per row
// (eavqa_quantize_rows_fp8 / the fp8 output of eavqa_layernorm_splitk), B = e4m3 weights with one scale per tensor.  A k-step is 128 values
// = one whole 128-byte line per row and ONE v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales - the instruction eavqa_gemm_fp8 uses,
// so the partial sums differ from its accumulator by fp32 summation order only (the plain v_mfma_f32_16x16x32_fp8_fp8 adds its products with
// fewer guard bits: 2e-5 relative against float64 - enough, through the e4m3 rounding of every following activation, to move logits by
// percents between the cached and the re-forward generation).  Lane (x, g) holds the 16-byte chunks g and g + 4 of its row's step for both
// operands (gemm_fp8.hip's fragment layout).  The row and tensor scales are applied before the partial sums are written: every consumer of
// bf16 partial sums (LayerNorm pass, finish, decode attention) takes them unchanged.  Half the weight bytes per step of the bf16 decode.
typedef __attribute__((ext_vector_type(8))) int dec_i32x8;
template <int MF, int NF, int NW, int U>
__global__ __launch_bounds__(64 * NW) void gemm_fp8_splitk_kernel(const unsigned char* __restrict__ A, int64_t lda, const float* __restrict__ a_scale,
                                                                  const unsigned char* __restrict__ B, int64_t ldb, float b_scale,
                                                                  float* __restrict__ P, int M, int N, int KS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 16 * MF * 128;                     // bytes of one [16 MF rows][128 k] A tile
    constexpr int COLS = 16 * NF * NW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * COLS + wave * 16 * NF, slice = blockIdx.y, k0 = slice * KS;
    const int nsteps = KS >> 7;
    const int x = lane & 15, g = lane >> 4;
    const unsigned char* bp[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) bp[j] = B + (int64_t)min(n0 + 16 * j + x, N - 1) * ldb + k0 + 16 * g;
    uint4 blo[U][NF], bhi[U][NF];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            blo[u][j] = *reinterpret_cast<const uint4*>(bp[j] + 128 * min(u, nsteps - 1));
            bhi[u][j] = *reinterpret_cast<const uint4*>(bp[j] + 128 * min(u, nsteps - 1) + 64);
        }
    // A slice: chunk c -> tile c / (128 MF), row (c % (128 MF)) >> 3, physical slot c & 7 holds logical chunk slot ^ (row & 7)
    const int total = nsteps * 128 * MF;
    for (int base = wave * 64; base < total; base += 64 * NW) {
        const int c = base + lane;
        if (c < total) {
            const int t = c / (128 * MF), within = c % (128 * MF);
            const int row = within >> 3, pc = within & 7;
            const unsigned char* src = A + (int64_t)min(row, M - 1) * lda + k0 + t * 128 + ((pc ^ (row & 7)) << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + base * 16), 16, 0, 0);
        }
    }
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int a_off = x * 128 + ((g ^ (x & 7)) << 4);       // chunk g; chunk g + 4 sits at a_off ^ 64
    __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00);
    __syncthreads();
    auto frag = [](const uint4& lo, const uint4& hi) {
        return (dec_i32x8){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    };
    for (int s0 = 0; s0 < nsteps; s0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dec_i32x8 w[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) w[j] = frag(blo[u][j], bhi[u][j]);
            if (s0 + U + u < nsteps) {
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    blo[u][j] = *reinterpret_cast<const uint4*>(bp[j] + 128 * (s0 + U + u));
                    bhi[u][j] = *reinterpret_cast<const uint4*>(bp[j] + 128 * (s0 + U + u) + 64);
                }
            }
            if (s0 + u < nsteps) {
                const char* tile = smem + (s0 + u) * TILE;
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    const dec_i32x8 af = frag(*reinterpret_cast<const uint4*>(tile + a_off + i * 2048),
                                              *reinterpret_cast<const uint4*>(tile + (a_off ^ 64) + i * 2048));
#pragma unroll
                    for (int j = 0; j < NF; ++j)      // cbsz = blgp = 0: both operands e4m3; block scales 2^0 (E8M0 byte 127)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, w[j], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = n0 + 16 * j + x;
        if (n < N) {
            float* out = P + (int64_t)slice * M * N + n;
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * i + 4 * g + r;
                    if (m < M) out[(int64_t)m * N] = acc[i][j][r] * (a_scale[m] * b_scale);
                }
        }
    }
}

struct FinishSeg { void* dst; int64_t ld; };

// out[m, n] = act(sum_s P[s][m][n] + bias[n]) (+ residual[m, n]); 4 columns per thread
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(int M, int N, const float* __restrict__ P, int ks, const float* __restrict__ bias,
                                                            int act, const float* __restrict__ residual, int64_t ldr, int out_f32,
                                                            int seg_cols, FinishSeg s0, FinishSeg s1, FinishSeg s2) {
    const int nq = N >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * nq) return;
    const int m = idx / nq, n = (idx % nq) * 4;
    float4 v = *reinterpret_cast<const float4*>(P + (int64_t)m * N + n);
    for (int s = 1; s < ks; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(P + ((int64_t)s * M + m) * N + n);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    v.x = act_fwd(act, v.x); v.y = act_fwd(act, v.y); v.z = act_fwd(act, v.z); v.w = act_fwd(act, v.w);
    if (residual) {
        const float4 r = *reinterpret_cast<const float4*>(residual + (int64_t)m * ldr + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    const int seg = n / seg_cols, nl = n - seg * seg_cols;
    const FinishSeg d = seg == 0 ? s0 : (seg == 1 ? s1 : s2);
    if (out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.dst) + (int64_t)m * d.ld + nl) = v;
    else elem<T>::st4(reinterpret_cast<T*>(d.dst) + (int64_t)m * d.ld + nl, v);
}

// h[m, c] = act(sum_s P[s][m][c]) * sum_s P[s][m][F + c]: the finish of T5's [wi_0; wi_1] projection (HF:t5 :97-123); 4 columns per thread
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_gated_kernel(int M, int F, const float* __restrict__ P, int ks, int act, T* __restrict__ out,
                                                                  int64_t ld) {
    const int nq = F >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * nq) return;
    const int m = idx / nq, n = (idx % nq) * 4;
    const int64_t N2 = 2 * (int64_t)F;
    float4 u = *reinterpret_cast<const float4*>(P + m * N2 + n), w = *reinterpret_cast<const float4*>(P + m * N2 + F + n);
    for (int s = 1; s < ks; ++s) {
        const float* ps = P + ((int64_t)s * M + m) * N2 + n;
        const float4 a = *reinterpret_cast<const float4*>(ps), b = *reinterpret_cast<const float4*>(ps + F);
        u.x += a.x; u.y += a.y; u.z += a.z; u.w += a.w;
        w.x += b.x; w.y += b.y; w.z += b.z; w.w += b.w;
    }
    float4 v;
    v.x = act_fwd(act, u.x) * w.x; v.y = act_fwd(act, u.y) * w.y; v.z = act_fwd(act, u.z) * w.z; v.w = act_fwd(act, u.w) * w.w;
    elem<T>::st4(out + (int64_t)m * ld + n, v);
}

// one 512-thread workgroup per row (a decode step has at most 64 rows: parallelism must come from inside the row), the
// partial-sum loads of eight slices in flight at once: x = x_in + bias + sum_s P[s]; y = LN(x) * gamma + beta
constexpr int LNS_NT = 512;
__device__ __forceinline__ float block_sum8(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

template <typename T, int NV>
__global__ __launch_bounds__(LNS_NT) void ln_splitk_kernel(int rows, int cols, const float* __restrict__ x_in, int64_t ldx,
                                                           const float* __restrict__ P, int ks, const float* __restrict__ bias,
                                                           float* __restrict__ x_out, int64_t ldxo, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, T* __restrict__ y, int64_t ldy, int rms,
                                                           unsigned char* __restrict__ yq, int64_t ldq, float* __restrict__ yq_scale) {
    __shared__ float red[LNS_NT / 64];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int nv = cols >> 2;
    float4 v[NV], gm[NV], bt[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid + LNS_NT * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nv) {
            v[i] = *reinterpret_cast<const float4*>(x_in + (int64_t)row * ldx + 4 * c);
            gm[i] = *reinterpret_cast<const float4*>(gamma + 4 * c);
            bt[i] = rms ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(beta + 4 * c);
            if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + 4 * c); v[i].x += b.x; v[i].y += b.y; v[i].z += b.z; v[i].w += b.w; }
        }
    }
    const int64_t slice = (int64_t)rows * cols;
    const float* prow = P + (int64_t)row * cols;
    int sl = 0;
    for (; sl + 8 <= ks; sl += 8) {
        float4 t[8][NV];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = tid + LNS_NT * i;
                t[u][i] = c < nv ? *reinterpret_cast<const float4*>(prow + (sl + u) * slice + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int u = 0; u < 8; ++u)          // slices are added in index order: the sum does not depend on scheduling
#pragma unroll
            for (int i = 0; i < NV; ++i) { v[i].x += t[u][i].x; v[i].y += t[u][i].y; v[i].z += t[u][i].z; v[i].w += t[u][i].w; }
    }
    if (sl < ks) {
        float4 t[8][NV];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = tid + LNS_NT * i;
                t[u][i] = (c < nv && sl + u < ks) ? *reinterpret_cast<const float4*>(prow + (sl + u) * slice + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (sl + u < ks) { v[i].x += t[u][i].x; v[i].y += t[u][i].y; v[i].z += t[u][i].z; v[i].w += t[u][i].w; }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid + LNS_NT * i;
        if (c < nv) {
            if (x_out) *reinterpret_cast<float4*>(x_out + (int64_t)row * ldxo + 4 * c) = v[i];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    // rms (T5LayerNorm, HF:t5 :50-72): no mean subtraction, no bias - y = gamma * x * rsqrt(mean(x^2) + eps)
    const float mu = rms ? 0.f : block_sum8(s, red) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid + LNS_NT * i;
        if (c < nv) {
            const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rs = rsqrtf(block_sum8(q, red) / (float)cols + eps);
    float4 o[NV];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid + LNS_NT * i;
        if (c < nv) {
            o[i].x = (v[i].x - mu) * rs * gm[i].x + bt[i].x;
            o[i].y = (v[i].y - mu) * rs * gm[i].y + bt[i].y;
            o[i].z = (v[i].z - mu) * rs * gm[i].z + bt[i].z;
            o[i].w = (v[i].w - mu) * rs * gm[i].w + bt[i].w;
            if (y) elem<T>::st4(y + (int64_t)row * ldy + 4 * c, o[i]);
            if (yq) {          // what eavqa_quantize_rows_fp8 would see: the values rounded to the storage type first
                o[i].x = (float)(T)o[i].x; o[i].y = (float)(T)o[i].y; o[i].z = (float)(T)o[i].z; o[i].w = (float)(T)o[i].w;
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o[i].x), fabsf(o[i].y)), fmaxf(fabsf(o[i].z), fabsf(o[i].w))));
            }
        }
    }
    if (yq) {
        // the row-wise e4m3 quantisation of eavqa_quantize_rows_fp8, fused: scale = amax / 448 (1 for an all-zero row), same arithmetic
        amax = wave_max(amax);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
        const float scale = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
        const float inv = 1.f / scale;
        if (tid == 0) yq_scale[row] = scale;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = tid + LNS_NT * i;
            if (c < nv) {
                int pk = __builtin_amdgcn_cvt_pk_fp8_f32(o[i].x * inv, o[i].y * inv, 0, false);
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(o[i].z * inv, o[i].w * inv, pk, true);
                *reinterpret_cast<int*>(yq + (int64_t)row * ldq + 4 * c) = pk;
            }
        }
    }
}

inline int splitk_mf(int M) { return M <= 16 ? 1 : (M <= 32 ? 2 : 4); }

}  // namespace

// Plan (measured on OPT-2.7B's four decode shapes at M = 32, tools/decode_bench.py --sweep, profiles/round2_decode.md): 8-wave
// workgroups of 128 columns (the A slice is staged once per 128 columns: a quarter of a byte of A per weight byte), ONE workgroup per
// CU where the shape allows it - the largest ks with (N / 128) x ks <= 256 and slices of at least 256 k-values - and a 16-deep
// weight-load window when the slice has 16+ steps.  Slices stay <= 1024 k-values (64 KiB of LDS at M <= 32).
extern "C" int eavqa_gemm_splitk_plan(int M, int N, int K) {
    if (M <= 0 || M > 64 || N <= 0 || K <= 0 || K % 32) return 0;
    const int mf = splitk_mf(M), groups = (N + 127) / 128;
    int best = 0, one_round = 0, fine = 0;
    for (int ks = 1; ks <= 32; ++ks) {
        if (K % (32 * ks)) continue;
        const int KS = K / ks;
        if (KS * mf * 32 > 64 * 1024) continue;          // staged A slice
        if (!best) best = ks;                            // smallest admissible split
        if (groups * ks <= 256 && KS >= 256) one_round = ks;
        if (KS >= 512) fine = ks;
    }
    // no split fits one workgroup per CU (FFN-up of OPT-2.7B: 80 column groups, K = 2560 admits ks >= 4): several workgroups share a CU
    // anyway, and slices of 512 k-values balance better than the smallest split (round-3 sweep: ks = 5 15.4 us against ks = 4 17.7 us)
    return one_round ? one_round : (fine > best ? fine : best);
}

namespace {
int gemm_splitk_impl(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, float* partials, int ks,
                     void* stream, int unroll) {
    if (dtype != EAVQA_BF16) return EAVQA_E_DTYPE;
    if (!A || !B || !partials || M <= 0 || M > 64 || N <= 0 || K <= 0 || ks <= 0) return EAVQA_E_ARG;
    if (K % (32 * ks) || lda < K || ldb < K) return EAVQA_E_SHAPE;
    if (lda % 8 || ldb % 8 || !eavqa_aligned16(A) || !eavqa_aligned16(B)) return EAVQA_E_ALIGN;
    const int mf = splitk_mf(M), KS = K / ks;
    const int lds = KS * mf * 32;
    if (lds > 150 * 1024) return EAVQA_E_SHAPE;
    // selector: bits [7:0] U (8 / 16), [11:8] NW (4 / 8), [15:12] NF (1 / 2); 0 fields = defaults
    const int U = (unroll & 0xFF) ? (unroll & 0xFF) : (KS >= 512 ? 16 : 8);
    const int NW = ((unroll >> 8) & 0xF) ? ((unroll >> 8) & 0xF) : 8, NF = ((unroll >> 12) & 0xF) ? ((unroll >> 12) & 0xF) : 1;
    if ((U != 8 && U != 16) || (NW != 4 && NW != 8) || (NF != 1 && NF != 2)) return EAVQA_E_ARG;
    typedef void (*kernel_t)(const bf16_t*, int64_t, const bf16_t*, int64_t, float*, int, int, int);
#define EAVQA_SK_ROW(MFV) {{{gemm_bf16_splitk_kernel<MFV, 1, 4, 8>, gemm_bf16_splitk_kernel<MFV, 1, 4, 16>},   \
                            {gemm_bf16_splitk_kernel<MFV, 1, 8, 8>, gemm_bf16_splitk_kernel<MFV, 1, 8, 16>}},  \
                           {{gemm_bf16_splitk_kernel<MFV, 2, 4, 8>, gemm_bf16_splitk_kernel<MFV, 2, 4, 16>},   \
                            {gemm_bf16_splitk_kernel<MFV, 2, 8, 8>, gemm_bf16_splitk_kernel<MFV, 2, 8, 16>}}}
    static const kernel_t kernels[3][2][2][2] = {EAVQA_SK_ROW(1), EAVQA_SK_ROW(2), EAVQA_SK_ROW(4)};      // [mf][NF][NW][U]
#undef EAVQA_SK_ROW
    static std::atomic<bool> configured[3][2][2][2];
    const int im = mf == 1 ? 0 : (mf == 2 ? 1 : 2), in = NF - 1, iw = NW == 8, iu = U == 16;
    const kernel_t kernel = kernels[im][in][iw][iu];
    if (!configured[im][in][iw][iu].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return EAVQA_E_LAUNCH;
        configured[im][in][iw][iu].store(true, std::memory_order_release);
    }
    const int cols = 16 * NF * NW;
    hipLaunchKernelGGL(kernel, dim3((N + cols - 1) / cols, ks), dim3(64 * NW), lds, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16_t*>(A), lda, reinterpret_cast<const bf16_t*>(B), ldb, partials, M, N, KS);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}
}  // namespace

extern "C" int eavqa_gemm_splitk(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb,
                                 float* partials, int ks, void* stream) {
    return gemm_splitk_impl(dtype, M, N, K, A, lda, B, ldb, partials, ks, stream, 0);
}

extern "C" int eavqa_gemm_splitk_ex(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb,
                                    float* partials, int ks, void* stream, int unroll) {
    return gemm_splitk_impl(dtype, M, N, K, A, lda, B, ldb, partials, ks, stream, unroll);
}

// fp8 plan: the bf16 rules in bytes - a k-step is 128 values, the staged A slice KS x 16 MF bytes, slices of at least 512 values
extern "C" int eavqa_gemm_fp8_splitk_plan(int M, int N, int K) {
    if (M <= 0 || M > 64 || N <= 0 || K <= 0 || K % 128) return 0;
    const int mf = splitk_mf(M), groups = (N + 127) / 128;
    int best = 0, one_round = 0, fine = 0;
    for (int ks = 1; ks <= 32; ++ks) {
        if (K % (128 * ks)) continue;
        const int KS = K / ks;
        if (KS * mf * 16 > 64 * 1024) continue;
        if (!best) best = ks;
        if (groups * ks <= 256 && KS >= 512) one_round = ks;
        if (KS >= 1024) fine = ks;
    }
    return one_round ? one_round : (fine > best ? fine : best);
}

extern "C" int eavqa_gemm_fp8_splitk(int M, int N, int K, const void* A, int64_t lda, const float* a_row_scale, const void* B, int64_t ldb,
                                     float b_scale, float* partials, int ks, void* stream) {
    if (!A || !B || !a_row_scale || !partials || M <= 0 || M > 64 || N <= 0 || K <= 0 || ks <= 0) return EAVQA_E_ARG;
    if (K % (128 * ks) || lda < K || ldb < K) return EAVQA_E_SHAPE;
    if (lda % 16 || ldb % 16 || !eavqa_aligned16(A) || !eavqa_aligned16(B)) return EAVQA_E_ALIGN;
    const int mf = splitk_mf(M), KS = K / ks;
    const int lds = KS * mf * 16;
    if (lds > 150 * 1024) return EAVQA_E_SHAPE;
    const int U = KS >= 1024 ? 8 : 4;                       // 128-byte steps: 8 of them = the 16 loads per lane of the bf16 kernel's deep window
    typedef void (*kernel_t)(const unsigned char*, int64_t, const float*, const unsigned char*, int64_t, float, float*, int, int, int);
    static const kernel_t kernels[3][2] = {{gemm_fp8_splitk_kernel<1, 1, 8, 4>, gemm_fp8_splitk_kernel<1, 1, 8, 8>},
                                           {gemm_fp8_splitk_kernel<2, 1, 8, 4>, gemm_fp8_splitk_kernel<2, 1, 8, 8>},
                                           {gemm_fp8_splitk_kernel<4, 1, 8, 4>, gemm_fp8_splitk_kernel<4, 1, 8, 8>}};
    static std::atomic<bool> configured[3][2];
    const int im = mf == 1 ? 0 : (mf == 2 ? 1 : 2), iu = U == 8;
    const kernel_t kernel = kernels[im][iu];
    if (!configured[im][iu].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return EAVQA_E_LAUNCH;
        configured[im][iu].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kernel, dim3((N + 127) / 128, ks), dim3(512), lds, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned char*>(A), lda, a_row_scale, reinterpret_cast<const unsigned char*>(B), ldb, b_scale, partials,
                       M, N, KS);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_splitk_finish(int dtype, int M, int N, const float* partials, int ks, const float* bias, int act,
                                   const float* residual, int64_t ld_residual, int out_f32, int n_seg,
                                   void* out0, int64_t ld0, void* out1, int64_t ld1, void* out2, int64_t ld2, void* stream) {
    if (dtype != EAVQA_BF16 && dtype != EAVQA_F32) return EAVQA_E_DTYPE;
    if (!partials || !out0 || M <= 0 || N <= 0 || ks <= 0 || n_seg < 1 || n_seg > 3) return EAVQA_E_ARG;
    if ((n_seg > 1 && !out1) || (n_seg > 2 && !out2)) return EAVQA_E_ARG;
    if (N % (4 * n_seg) || ld0 % 4 || (n_seg > 1 && ld1 % 4) || (n_seg > 2 && ld2 % 4) || (residual && ld_residual % 4)) return EAVQA_E_SHAPE;
    if (act < EAVQA_ACT_NONE || act > EAVQA_ACT_QUICK_GELU) return EAVQA_E_DTYPE;
    const int seg_cols = N / n_seg;
    const FinishSeg s0{out0, ld0}, s1{out1, ld1}, s2{out2, ld2};
    const int threads = M * (N / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_BF16)
        hipLaunchKernelGGL(splitk_finish_kernel<bf16_t>, dim3((threads + 255) / 256), dim3(256), 0, s, M, N, partials, ks, bias, act,
                           residual, ld_residual, out_f32, seg_cols, s0, s1, s2);
    else
        hipLaunchKernelGGL(splitk_finish_kernel<float>, dim3((threads + 255) / 256), dim3(256), 0, s, M, N, partials, ks, bias, act,
                           residual, ld_residual, 1, seg_cols, s0, s1, s2);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_splitk_finish_gated(int dtype, int M, int F, const float* partials, int ks, int act, void* out, int64_t ld, void* stream) {
    if (dtype != EAVQA_BF16 && dtype != EAVQA_F32) return EAVQA_E_DTYPE;
    if (!partials || !out || M <= 0 || F <= 0 || ks <= 0) return EAVQA_E_ARG;
    if (F % 4 || ld % 4) return EAVQA_E_SHAPE;
    if (act < EAVQA_ACT_NONE || act > EAVQA_ACT_QUICK_GELU) return EAVQA_E_DTYPE;
    const int threads = M * (F / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_BF16)
        hipLaunchKernelGGL(splitk_finish_gated_kernel<bf16_t>, dim3((threads + 255) / 256), dim3(256), 0, s, M, F, partials, ks, act,
                           reinterpret_cast<bf16_t*>(out), ld);
    else
        hipLaunchKernelGGL(splitk_finish_gated_kernel<float>, dim3((threads + 255) / 256), dim3(256), 0, s, M, F, partials, ks, act,
                           reinterpret_cast<float*>(out), ld);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

static int norm_splitk_impl(int dtype, int rows, int cols, const float* x_in, int64_t ldx, const float* partials, int ks,
                            const float* bias, float* x_out, int64_t ld_out, const float* gamma, const float* beta,
                            float eps, void* y, int64_t ldy, void* stream, int rms, void* yq = nullptr, int64_t ldq = 0,
                            float* yq_scale = nullptr) {
    if (dtype != EAVQA_BF16 && dtype != EAVQA_F32) return EAVQA_E_DTYPE;
    if (!x_in || !gamma || !beta || (!y && !yq) || rows <= 0 || cols <= 0 || ks < 0 || (ks > 0 && !partials)) return EAVQA_E_ARG;
    if (yq && (!yq_scale || ldq % 4)) return EAVQA_E_ARG;
    if (cols % 4 || cols > 64 * 4 * 16) return EAVQA_E_SHAPE;
    if (ldx % 4 || ldy % 4 || (x_out && ld_out % 4)) return EAVQA_E_ALIGN;
    const int nv = (cols / 4 + LNS_NT - 1) / LNS_NT;
    const dim3 grid(rows), block(LNS_NT);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define EAVQA_LNS(T, NV)                                                                                                    \
    hipLaunchKernelGGL((ln_splitk_kernel<T, NV>), grid, block, 0, s, rows, cols, x_in, ldx, partials, ks, bias, x_out, ld_out, \
                       gamma, beta, eps, reinterpret_cast<T*>(y), ldy, rms, reinterpret_cast<unsigned char*>(yq), ldq, yq_scale)
    if (dtype == EAVQA_BF16) {
        if (nv <= 1) EAVQA_LNS(bf16_t, 1); else if (nv <= 2) EAVQA_LNS(bf16_t, 2); else EAVQA_LNS(bf16_t, 4);
    } else {
        if (nv <= 1) EAVQA_LNS(float, 1); else if (nv <= 2) EAVQA_LNS(float, 2); else EAVQA_LNS(float, 4);
    }
#undef EAVQA_LNS
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_layernorm_splitk(int dtype, int rows, int cols, const float* x_in, int64_t ldx, const float* partials, int ks,
                                      const float* bias, float* x_out, int64_t ld_out, const float* gamma, const float* beta,
                                      float eps, void* y, int64_t ldy, void* stream) {
    if (!beta) return EAVQA_E_ARG;
    return norm_splitk_impl(dtype, rows, cols, x_in, ldx, partials, ks, bias, x_out, ld_out, gamma, beta, eps, y, ldy, stream, 0);
}

extern "C" int eavqa_layernorm_splitk_fp8(int rows, int cols, const float* x_in, int64_t ldx, const float* partials, int ks, const float* bias,
                                          float* x_out, int64_t ld_out, const float* gamma, const float* beta, float eps, void* yq, int64_t ldq,
                                          float* row_scale, void* stream) {
    if (!beta || !yq) return EAVQA_E_ARG;
    return norm_splitk_impl(EAVQA_BF16, rows, cols, x_in, ldx, partials, ks, bias, x_out, ld_out, gamma, beta, eps, nullptr, 4, stream, 0, yq, ldq, row_scale);
}

extern "C" int eavqa_rmsnorm_splitk(int dtype, int rows, int cols, const float* x_in, int64_t ldx, const float* partials, int ks,
                                    float* x_out, int64_t ld_out, const float* gamma, float eps, void* y, int64_t ldy, void* stream) {
    return norm_splitk_impl(dtype, rows, cols, x_in, ldx, partials, ks, nullptr, x_out, ld_out, gamma, gamma, eps, y, ldy, stream, 1);
}
