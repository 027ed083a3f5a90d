// Attention forward / backward, fp32 arithmetic on the vector ALU, generic over the storage
// type (eavqa_attention_fwd / _bwd in include/eavqa.h).
//
// This is the exact-arithmetic path used for parity in both dtypes; the sequences on this hot
// path are short (LM: S = 42..200, ViT: N = 50..577, mapper: 20), so each workgroup streams
// 64-key (or 64-query) tiles of one (batch, head) through LDS as fp32 and every query row
// (key row in the dK/dV pass) is owned by LPR adjacent lanes that each hold hd/LPR dims in
// registers.  Softmax is online (running max / sum per row), scores are processed in chunks of
// 8 keys so that the accumulator rescale is paid once per chunk.
//
// Masking: a masked score is REPLACED by -FLT_MAX (HF adds finfo.min to a score that is
// negligible against it), so a fully masked row degrades to the uniform average, never NaN.
#include "common.h"
#include "attention_mfma.hip"   // bf16 matrix-core kernels (same translation unit)

namespace {


constexpr int CK = 8;  // keys per softmax chunk

struct AttnParams {
    const void* q; const void* k; const void* v; const void* o; const void* d_o;
    void* out;   // fwd: o
    void* dq; void* dk; void* dv;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    const int32_t* key_mask;
    const int32_t* cu;   // packed self-attention: sample b owns rows [cu[b], cu[b+1]) of q/k/v/o (Sq = Sk = that length)
    float* lse; float* delta;
    int B, H, Sq, Sk, hd, causal, tile;
    int64_t ld_mask;
    int stat_ld;         // row pitch of lse / delta: [B, H, stat_ld]
    int64_t bsq, bsk;   // rows between consecutive batches of q/o/do/dq and of k/v/dk/dv
    float scale;
    // T5 relative-position bias (eavqa_attention_fwd_rel / _bwd_rel): score(i, j) += rel_bias[h * rel_ld + (j - (i + Sk - Sq)) + rel_zero]
    const float* rel_bias; int64_t rel_ld; int rel_zero;
};

__device__ __forceinline__ float rel_term(const AttnParams& p, int h, int i_plus_off, int j) {
    if (!p.rel_bias) return 0.f;
    const int idx = min(max((j - i_plus_off) + p.rel_zero, 0), (int)p.rel_ld - 1);      // (rows / keys beyond the sequence: clamped, unused)
    return p.rel_bias[(int64_t)h * p.rel_ld + idx];
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// cooperative load of `nrows` rows (row index r0.., bound rmax) of one head into LDS as fp32 [tile][hd]
template <typename T>
__device__ __forceinline__ void stage_rows(float* dst, const T* src, int64_t ld, int64_t base_row, int r0, int rmax,
                                           int tile, int hd, int head_off) {
    const int per_row = hd >> 2;
    const int total = tile * per_row;
    for (int c = threadIdx.x; c < total; c += blockDim.x) {
        const int r = c / per_row, d4 = c - r * per_row;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + r < rmax) val = elem<T>::ld4(src + (base_row + r0 + r) * ld + head_off + 4 * d4);
        *reinterpret_cast<float4*>(dst + r * hd + 4 * d4) = val;
    }
}

// ------------------------------------------------------------------ forward
template <typename T, int LPR, int DP4>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Ks = reinterpret_cast<float*>(smem_raw);
    float* Vs = Ks + p.tile * p.hd;
    int* valid = reinterpret_cast<int*>(Vs + p.tile * p.hd);

    constexpr int ROWS = 256 / LPR;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int r_local = threadIdx.x / LPR, part = threadIdx.x % LPR;
    const int r0 = blockIdx.x * ROWS;
    if (p.cu) {                                 // packed: this sample's own length and row base
        const int base = p.cu[b], len = p.cu[b + 1] - base;
        if (r0 >= len) return;
        p.Sq = p.Sk = len;
        p.bsq = p.bsk = 0;
        const int64_t skip_q = (int64_t)base;
        p.q = reinterpret_cast<const T*>(p.q) + skip_q * p.ldq;
        p.k = reinterpret_cast<const T*>(p.k) + skip_q * p.ldk;
        p.v = reinterpret_cast<const T*>(p.v) + skip_q * p.ldv;
        p.out = reinterpret_cast<T*>(p.out) + skip_q * p.ldo;
    }
    const int i = r0 + r_local;                 // query row
    const bool active = i < p.Sq;
    const int off = p.Sk - p.Sq;                // causal: key j visible iff j <= i + off
    const int head_off = h * p.hd;
    const int dbase = part * (DP4 * 4);
    const T* Q = reinterpret_cast<const T*>(p.q);
    const T* K = reinterpret_cast<const T*>(p.k);
    const T* V = reinterpret_cast<const T*>(p.v);

    float4 qv[DP4], acc[DP4];
#pragma unroll
    for (int d = 0; d < DP4; ++d) {
        qv[d] = active ? elem<T>::ld4(Q + ((int64_t)b * p.bsq + i) * p.ldq + head_off + dbase + 4 * d)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        acc[d] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float m = -FLT_MAX, l = 0.f;

    // last key any row of this block may see
    int k_end = p.Sk;
    if (p.causal) { const int last = min(p.Sq, r0 + ROWS) - 1 + off; k_end = min(p.Sk, last + 1); }
    if (k_end < 1) k_end = min(p.Sk, 1);

    for (int k0 = 0; k0 < k_end; k0 += p.tile) {
        __syncthreads();
        stage_rows<T>(Ks, K, p.ldk, (int64_t)b * p.bsk, k0, p.Sk, p.tile, p.hd, head_off);
        stage_rows<T>(Vs, V, p.ldv, (int64_t)b * p.bsk, k0, p.Sk, p.tile, p.hd, head_off);
        for (int c = threadIdx.x; c < p.tile; c += blockDim.x)
            valid[c] = (k0 + c < p.Sk) && (!p.key_mask || p.key_mask[(int64_t)b * p.ld_mask + k0 + c] != 0);
        __syncthreads();
        const int nkeys = min(p.tile, p.Sk - k0);
        for (int c0 = 0; c0 < nkeys; c0 += CK) {
            float sc[CK];
            float cmax = -FLT_MAX;
#pragma unroll
            for (int c = 0; c < CK; ++c) {
                const int j = c0 + c;
                float s = 0.f;
                if (j < nkeys) {
                    const float* kr = Ks + j * p.hd + dbase;
#pragma unroll
                    for (int d = 0; d < DP4; ++d) {
                        const float4 kv = *reinterpret_cast<const float4*>(kr + 4 * d);
                        s += qv[d].x * kv.x + qv[d].y * kv.y + qv[d].z * kv.z + qv[d].w * kv.w;
                    }
                }
                s = group_sum<LPR>(s) * p.scale;
                if (j < nkeys) s += rel_term(p, h, i + off, k0 + j);
                const bool vis = (j < nkeys) && valid[j] && (!p.causal || (k0 + j) <= i + off);
                sc[c] = (j < nkeys) ? (vis ? s : -FLT_MAX) : -INFINITY;   // -inf: key does not exist
                cmax = fmaxf(cmax, sc[c]);
            }
            const float m_new = fmaxf(m, cmax);
            const float corr = expf(m - m_new);
            l *= corr;
#pragma unroll
            for (int d = 0; d < DP4; ++d) { acc[d].x *= corr; acc[d].y *= corr; acc[d].z *= corr; acc[d].w *= corr; }
#pragma unroll
            for (int c = 0; c < CK; ++c) {
                const int j = c0 + c;
                if (j < nkeys) {
                    const float pj = expf(sc[c] - m_new);
                    l += pj;
                    const float* vr = Vs + j * p.hd + dbase;
#pragma unroll
                    for (int d = 0; d < DP4; ++d) {
                        const float4 vv = *reinterpret_cast<const float4*>(vr + 4 * d);
                        acc[d].x += pj * vv.x; acc[d].y += pj * vv.y; acc[d].z += pj * vv.z; acc[d].w += pj * vv.w;
                    }
                }
            }
            m = m_new;
        }
    }
    if (active) {
        const float inv = 1.f / l;
        T* O = reinterpret_cast<T*>(p.out);
#pragma unroll
        for (int d = 0; d < DP4; ++d)
            elem<T>::st4(O + ((int64_t)b * p.bsq + i) * p.ldo + head_off + dbase + 4 * d,
                         make_float4(acc[d].x * inv, acc[d].y * inv, acc[d].z * inv, acc[d].w * inv));
        if (p.lse && part == 0) p.lse[((int64_t)b * p.H + h) * p.stat_ld + i] = m + logf(l);
    }
}

// ------------------------------------------------------------------ backward, dQ (+ delta)
template <typename T, int LPR, int DP4>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Ks = reinterpret_cast<float*>(smem_raw);
    float* Vs = Ks + p.tile * p.hd;
    int* valid = reinterpret_cast<int*>(Vs + p.tile * p.hd);

    constexpr int ROWS = 256 / LPR;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int r_local = threadIdx.x / LPR, part = threadIdx.x % LPR;
    const int r0 = blockIdx.x * ROWS;
    if (p.cu) {
        const int base = p.cu[b], len = p.cu[b + 1] - base;
        if (r0 >= len) return;
        p.Sq = p.Sk = len;
        p.bsq = p.bsk = 0;
        const int64_t skip = (int64_t)base;
        p.q = reinterpret_cast<const T*>(p.q) + skip * p.ldq;
        p.k = reinterpret_cast<const T*>(p.k) + skip * p.ldk;
        p.v = reinterpret_cast<const T*>(p.v) + skip * p.ldv;
        p.o = reinterpret_cast<const T*>(p.o) + skip * p.ldo;
        p.d_o = reinterpret_cast<const T*>(p.d_o) + skip * p.lddo;
        p.dq = reinterpret_cast<T*>(p.dq) + skip * p.lddq;
    }
    const int i = r0 + r_local;
    const bool active = i < p.Sq;
    const int off = p.Sk - p.Sq;
    const int head_off = h * p.hd;
    const int dbase = part * (DP4 * 4);
    const T* Q = reinterpret_cast<const T*>(p.q);
    const T* K = reinterpret_cast<const T*>(p.k);
    const T* V = reinterpret_cast<const T*>(p.v);
    const T* O = reinterpret_cast<const T*>(p.o);
    const T* DO = reinterpret_cast<const T*>(p.d_o);

    float4 qv[DP4], dov[DP4], dq[DP4];
    float dsum = 0.f;
#pragma unroll
    for (int d = 0; d < DP4; ++d) {
        const int64_t row = (int64_t)b * p.bsq + i;
        const int col = head_off + dbase + 4 * d;
        qv[d] = active ? elem<T>::ld4(Q + row * p.ldq + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        dov[d] = active ? elem<T>::ld4(DO + row * p.lddo + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 ov = active ? elem<T>::ld4(O + row * p.ldo + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        dsum += dov[d].x * ov.x + dov[d].y * ov.y + dov[d].z * ov.z + dov[d].w * ov.w;
        dq[d] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float delta = group_sum<LPR>(dsum);
    const int64_t stat = ((int64_t)b * p.H + h) * p.stat_ld + i;
    const float lse = active ? p.lse[stat] : 0.f;
    if (active && part == 0) p.delta[stat] = delta;

    int k_end = p.Sk;
    if (p.causal) { const int last = min(p.Sq, r0 + ROWS) - 1 + off; k_end = min(p.Sk, last + 1); }
    if (k_end < 1) k_end = min(p.Sk, 1);

    for (int k0 = 0; k0 < k_end; k0 += p.tile) {
        __syncthreads();
        stage_rows<T>(Ks, K, p.ldk, (int64_t)b * p.bsk, k0, p.Sk, p.tile, p.hd, head_off);
        stage_rows<T>(Vs, V, p.ldv, (int64_t)b * p.bsk, k0, p.Sk, p.tile, p.hd, head_off);
        for (int c = threadIdx.x; c < p.tile; c += blockDim.x)
            valid[c] = (k0 + c < p.Sk) && (!p.key_mask || p.key_mask[(int64_t)b * p.ld_mask + k0 + c] != 0);
        __syncthreads();
        const int nkeys = min(p.tile, p.Sk - k0);
        for (int j = 0; j < nkeys; ++j) {
            const float* kr = Ks + j * p.hd + dbase;
            const float* vr = Vs + j * p.hd + dbase;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < DP4; ++d) {
                const float4 kv = *reinterpret_cast<const float4*>(kr + 4 * d);
                const float4 vv = *reinterpret_cast<const float4*>(vr + 4 * d);
                s += qv[d].x * kv.x + qv[d].y * kv.y + qv[d].z * kv.z + qv[d].w * kv.w;
                dp += dov[d].x * vv.x + dov[d].y * vv.y + dov[d].z * vv.z + dov[d].w * vv.w;
            }
            s = group_sum<LPR>(s) * p.scale + rel_term(p, h, i + off, k0 + j);
            dp = group_sum<LPR>(dp);
            const bool vis = valid[j] && (!p.causal || (k0 + j) <= i + off);
            const float pj = expf((vis ? s : -FLT_MAX) - lse);
            const float ds = pj * (dp - delta) * p.scale;
#pragma unroll
            for (int d = 0; d < DP4; ++d) {
                const float4 kv = *reinterpret_cast<const float4*>(kr + 4 * d);
                dq[d].x += ds * kv.x; dq[d].y += ds * kv.y; dq[d].z += ds * kv.z; dq[d].w += ds * kv.w;
            }
        }
    }
    if (active) {
        T* DQ = reinterpret_cast<T*>(p.dq);
#pragma unroll
        for (int d = 0; d < DP4; ++d)
            elem<T>::st4(DQ + ((int64_t)b * p.bsq + i) * p.lddq + head_off + dbase + 4 * d, dq[d]);
    }
}

// ------------------------------------------------------------------ backward, dK and dV
template <typename T, int LPR, int DP4>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Qs = reinterpret_cast<float*>(smem_raw);
    float* DOs = Qs + p.tile * p.hd;
    float* stats = DOs + p.tile * p.hd;  // [tile] lse then [tile] delta

    constexpr int ROWS = 256 / LPR;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int r_local = threadIdx.x / LPR, part = threadIdx.x % LPR;
    const int j0 = blockIdx.x * ROWS;
    if (p.cu) {
        const int base = p.cu[b], len = p.cu[b + 1] - base;
        if (j0 >= len) return;
        p.Sq = p.Sk = len;
        p.bsq = p.bsk = 0;
        const int64_t skip = (int64_t)base;
        p.q = reinterpret_cast<const T*>(p.q) + skip * p.ldq;
        p.k = reinterpret_cast<const T*>(p.k) + skip * p.ldk;
        p.v = reinterpret_cast<const T*>(p.v) + skip * p.ldv;
        p.d_o = reinterpret_cast<const T*>(p.d_o) + skip * p.lddo;
        p.dk = reinterpret_cast<T*>(p.dk) + skip * p.lddk;
        p.dv = reinterpret_cast<T*>(p.dv) + skip * p.lddv;
    }
    const int j = j0 + r_local;                 // key row
    const bool active = j < p.Sk;
    const int off = p.Sk - p.Sq;
    const int head_off = h * p.hd;
    const int dbase = part * (DP4 * 4);
    const T* Q = reinterpret_cast<const T*>(p.q);
    const T* K = reinterpret_cast<const T*>(p.k);
    const T* V = reinterpret_cast<const T*>(p.v);
    const T* DO = reinterpret_cast<const T*>(p.d_o);

    float4 kv[DP4], vv[DP4], dk[DP4], dv[DP4];
#pragma unroll
    for (int d = 0; d < DP4; ++d) {
        const int64_t row = (int64_t)b * p.bsk + j;
        const int col = head_off + dbase + 4 * d;
        kv[d] = active ? elem<T>::ld4(K + row * p.ldk + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        vv[d] = active ? elem<T>::ld4(V + row * p.ldv + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        dk[d] = make_float4(0.f, 0.f, 0.f, 0.f);
        dv[d] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool kvalid = active && (!p.key_mask || p.key_mask[(int64_t)b * p.ld_mask + j] != 0);

    // first query any key of this block is visible to: i >= j - off
    int q_begin = 0;
    if (p.causal) q_begin = max(0, j0 - off);
    q_begin = (q_begin / p.tile) * p.tile;

    for (int q0 = q_begin; q0 < p.Sq; q0 += p.tile) {
        __syncthreads();
        stage_rows<T>(Qs, Q, p.ldq, (int64_t)b * p.bsq, q0, p.Sq, p.tile, p.hd, head_off);
        stage_rows<T>(DOs, DO, p.lddo, (int64_t)b * p.bsq, q0, p.Sq, p.tile, p.hd, head_off);
        for (int c = threadIdx.x; c < p.tile; c += blockDim.x) {
            const bool in = q0 + c < p.Sq;
            const int64_t st = ((int64_t)b * p.H + h) * p.stat_ld + q0 + c;
            stats[c] = in ? p.lse[st] : 0.f;
            stats[p.tile + c] = in ? p.delta[st] : 0.f;
        }
        __syncthreads();
        const int nq = min(p.tile, p.Sq - q0);
        for (int c = 0; c < nq; ++c) {
            const int i = q0 + c;
            const float* qr = Qs + c * p.hd + dbase;
            const float* dor = DOs + c * p.hd + dbase;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < DP4; ++d) {
                const float4 qq = *reinterpret_cast<const float4*>(qr + 4 * d);
                const float4 dd = *reinterpret_cast<const float4*>(dor + 4 * d);
                s += qq.x * kv[d].x + qq.y * kv[d].y + qq.z * kv[d].z + qq.w * kv[d].w;
                dp += dd.x * vv[d].x + dd.y * vv[d].y + dd.z * vv[d].z + dd.w * vv[d].w;
            }
            s = group_sum<LPR>(s) * p.scale + rel_term(p, h, i + off, j);
            dp = group_sum<LPR>(dp);
            const bool vis = kvalid && (!p.causal || j <= i + off);
            const float pj = expf((vis ? s : -FLT_MAX) - stats[c]);
            const float ds = pj * (dp - stats[p.tile + c]) * p.scale;
#pragma unroll
            for (int d = 0; d < DP4; ++d) {
                const float4 qq = *reinterpret_cast<const float4*>(qr + 4 * d);
                const float4 dd = *reinterpret_cast<const float4*>(dor + 4 * d);
                dv[d].x += pj * dd.x; dv[d].y += pj * dd.y; dv[d].z += pj * dd.z; dv[d].w += pj * dd.w;
                dk[d].x += ds * qq.x; dk[d].y += ds * qq.y; dk[d].z += ds * qq.z; dk[d].w += ds * qq.w;
            }
        }
    }
    if (active) {
        T* DK = reinterpret_cast<T*>(p.dk);
        T* DV = reinterpret_cast<T*>(p.dv);
#pragma unroll
        for (int d = 0; d < DP4; ++d) {
            const int64_t row = (int64_t)b * p.bsk + j;
            const int col = head_off + dbase + 4 * d;
            elem<T>::st4(DK + row * p.lddk + col, dk[d]);
            elem<T>::st4(DV + row * p.lddv + col, dv[d]);
        }
    }
}

enum { K_FWD = 0, K_DQ = 1, K_DKV = 2 };

template <typename T, int LPR, int DP4>
int launch_cfg(int which, AttnParams& p, hipStream_t s) {
    constexpr int ROWS = 256 / LPR;
    // tile rows so that two fp32 [tile][hd] images fit in 64 KiB
    int tile = 64;
    while (tile > 8 && (size_t)tile * p.hd * 8 > 60 * 1024) tile >>= 1;
    p.tile = tile;
    const size_t lds = (size_t)tile * p.hd * 8 + (size_t)tile * 8;
    if (which == K_FWD) {
        dim3 grid((p.Sq + ROWS - 1) / ROWS, p.B * p.H);
        hipLaunchKernelGGL((attn_fwd_kernel<T, LPR, DP4>), grid, dim3(256), lds, s, p);
    } else if (which == K_DQ) {
        dim3 grid((p.Sq + ROWS - 1) / ROWS, p.B * p.H);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, LPR, DP4>), grid, dim3(256), lds, s, p);
    } else {
        dim3 grid((p.Sk + ROWS - 1) / ROWS, p.B * p.H);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, LPR, DP4>), grid, dim3(256), lds, s, p);
    }
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

// lanes per row / float4 per lane for a head dim (hd = LPR * DP4 * 4)
template <typename T>
int dispatch(int which, AttnParams& p, hipStream_t s) {
    const int hd = p.hd;
#define EAVQA_ATTN_CASE(L, D) if (hd == (L) * (D) * 4) return launch_cfg<T, L, D>(which, p, s)
    EAVQA_ATTN_CASE(4, 4);   // 64
    EAVQA_ATTN_CASE(4, 5);   // 80
    EAVQA_ATTN_CASE(4, 6);   // 96
    EAVQA_ATTN_CASE(4, 8);   // 128
    EAVQA_ATTN_CASE(8, 5);   // 160
    EAVQA_ATTN_CASE(8, 8);   // 256
    EAVQA_ATTN_CASE(16, 5);  // 320
    EAVQA_ATTN_CASE(16, 8);  // 512
    EAVQA_ATTN_CASE(4, 1);   // 16
    EAVQA_ATTN_CASE(4, 2);   // 32
    EAVQA_ATTN_CASE(4, 3);   // 48
    EAVQA_ATTN_CASE(2, 1);   // 8
    EAVQA_ATTN_CASE(1, 1);   // 4
#undef EAVQA_ATTN_CASE
    return EAVQA_E_SHAPE;
}

int check_common(int dtype, int B, int H, int Sq, int Sk, int hd) {
    if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || hd <= 0) return EAVQA_E_ARG;
    if (dtype != EAVQA_F32 && dtype != EAVQA_BF16) return EAVQA_E_DTYPE;
    if (hd % 4) return EAVQA_E_SHAPE;
    if ((int64_t)B * H > 65535) return EAVQA_E_SHAPE;
    return EAVQA_OK;
}


// ------------------------------------------------------------ decode (Sq = 1) ---
// One new query per sample against a cache of Sk keys: the work is reading K and V once, and at decode batch sizes the
// kernel is a chain of dependent HBM round trips (measured 19.7 us at B = 8 and 23.3 us at B = 32 with one wave walking
// all keys of a head: latency, not bytes).  So the walk is cut short instead: a workgroup = 4 neighbouring heads of one
// sample (their K / V slices are adjacent in the [S, E] cache rows) x DEC_WPH waves per head, each wave taking every
// DEC_WPH-th group of keys; a wave gives LPK lanes to a key (16 bytes = 8 head dims each) and walks 64 / LPK keys per load
// instruction, DEC_U instructions in flight - 160 keys are ONE batch of loads per wave for K and one for V.  Scores go
// to LDS; every wave of a head reduces max / sum over all of them itself (no second exchange); the partial outputs of the
// DEC_WPH waves are summed through LDS in wave order.  Same masking rule as the tiled kernels: masked scores become
// -FLT_MAX (a fully masked row averages all keys).
// DEC_WPH: 4 when the grid fits the chip once (one 16-wave workgroup per CU), fewer for larger batches, where several smaller
// workgroups per CU overlap their latency chains instead.
// VLDS: V does not depend on the scores, so its bytes should be on their way while K is being scored - but a second batch of loads
// held in registers does not fit a 16-wave workgroup's 128 VGPRs (measured: 41 spilled registers, 26.5 us against 18 us).  So the
// whole V slice of the workgroup ([Sk keys][4 heads x hd], 100 KiB at Sk = 160, hd = 80) is fetched by LDS-DMA at the very start,
// costs no registers, and P.V reads it from LDS: the kernel is ONE HBM round trip.  Taken when the image fits (<= 128 KiB) and
// the grid is one workgroup per CU.
// Phase timestamps of workgroup (0, 0), one row per wave: only in the profiling build (tools/attn_stamps.sh, -DEAVQA_ATTN_STAMPS); the
// shipped library compiles EAVQA_STAMP to nothing.
#ifdef EAVQA_ATTN_STAMPS
__device__ unsigned long long eavqa_attn_stamps[16 * 16];
#define EAVQA_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) eavqa_attn_stamps[(threadIdx.x >> 6) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define EAVQA_STAMP(i) do { } while (0)
#endif

// VMODE 2 (round 4): BOTH batches in registers and in flight from the first instruction on - K, then V - on workgroups of 4 heads x 1 or 2
// waves (at most 8 waves per CU: 256 VGPRs each, room for 2 x DEC_U x 4 data registers).  tools/attn_stamps.py showed where the LDS-image
// kernel's 20 us go: ISSUING the ~100 LDS-DMA instructions of a CU takes 4.3 us for its first wave and 9.6 us for its sixteenth (an LDS-DMA
// holds the CU's issue for ~40 ns), the K loads queue behind them, and every wave then waits at the barrier for the last one's scores
// (15.4 us).  Plain loads issue in ~1 us and the kernel is one HBM round trip.  DEC_WPH == 1 (a head's keys fit one wave's batch: T5's
// decoder self-attention, <= 80 keys) also drops the cross-wave exchange of partial outputs and its barrier.
template <int LPK, int DEC_WPH, int DEC_U, int VMODE>
__global__ __launch_bounds__(256 * DEC_WPH) void attn_decode_kernel(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k,
                                                          int64_t ldk, const bf16_t* __restrict__ v, int64_t ldv, bf16_t* __restrict__ out,
                                                          int64_t ldo, int64_t bsq, int64_t bsk, const int32_t* __restrict__ key_mask,
                                                          int64_t ld_mask, float* __restrict__ lse, int H, int Sk, int hd, float scale,
                                                          const bf16_t* __restrict__ k_new, const bf16_t* __restrict__ v_new, int64_t ld_new,
                                                          const float* __restrict__ qkv_part, int ks, const float* __restrict__ qkv_bias,
                                                          int part_cols, int part_kv, const float* __restrict__ rel_bias, int64_t rel_ld, int rel_zero) {
    // k_new / v_new (eavqa_attention_decode): the K / V rows of the NEW position (key Sk - 1) still sit in the QKV projection's
    // output; the lanes that own that key take them from there and append them to the cache on the way (each 16-byte piece of a
    // cache row has exactly one owner lane), which saves the separate append pass of the decode step.
    // qkv_part (eavqa_attention_decode_splitk): q and the new K / V rows do not exist yet - the QKV projection left `ks` fp32 partial
    // sums [ks][B][3 E]; every lane adds up the 8 values it needs (slices in index order, then the bias, then rounded to bf16: exactly
    // what eavqa_splitk_finish would have stored), which also saves the finish pass.  part_cols = columns per row of the partial sums
    // (3 H hd: q | k | v, part_kv != 0; H hd: a cross-attention's q alone, part_kv == 0 - nothing to append).
    // rel_bias (T5, HF:t5 :217-279): score(j) += rel_bias[h * rel_ld + (j - (Sk - 1)) + rel_zero] - the one query sits at position Sk - 1.
    extern __shared__ float dec_sc[];                 // [4 heads][Sk] scores, then [4][DEC_WPH][128] partial outputs, then the V image
    EAVQA_STAMP(0);
    constexpr bool VLDS = VMODE == 1, VREG = VMODE == 2;
    constexpr int KPI = 64 / LPK;
    char* vimg = reinterpret_cast<char*>(dec_sc + 4 * Sk + 4 * DEC_WPH * 128);      // VLDS: [Sk][4 heads x hd] bf16
    const int cpk = hd >> 1;                          // 16-byte pieces per key in the image (4 heads x hd / 8)
    const bool appended = (qkv_part && part_kv) || k_new;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hh = wave / DEC_WPH, part = wave % DEC_WPH;
    const int b = blockIdx.x, h = blockIdx.y * 4 + hh;
    const bool head_ok = h < H;
    const int sub = lane / LPK, dl = lane % LPK;
    const bool active = head_ok && 8 * dl < hd;
    float* sc = dec_sc + hh * Sk;
    float* opart = dec_sc + 4 * Sk + (hh * DEC_WPH + part) * 128;
    constexpr int STEP = KPI * DEC_WPH * DEC_U;
    const int E3 = part_cols;
    // bf16(sum_s P[s][b][col .. col+7] + bias[col ..]) - the value eavqa_splitk_finish stores
    auto from_part = [&](int col) -> bf16x8 {
        const float* p0 = qkv_part + (int64_t)b * E3 + col;
        const int64_t slice = (int64_t)gridDim.x * E3;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
        for (int s0 = 0; s0 < ks; s0 += 4) {              // four slices' loads in flight, added in index order
            float4 ta[4], tc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* ps = p0 + min(s0 + i, ks - 1) * slice;
                ta[i] = *reinterpret_cast<const float4*>(ps);
                tc[i] = *reinterpret_cast<const float4*>(ps + 4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (s0 + i == 0) { a = ta[0]; c = tc[0]; }
                else if (s0 + i < ks) {
                    a.x += ta[i].x; a.y += ta[i].y; a.z += ta[i].z; a.w += ta[i].w;
                    c.x += tc[i].x; c.y += tc[i].y; c.z += tc[i].z; c.w += tc[i].w;
                }
            }
        }
        if (qkv_bias) {
            const float4 a2 = *reinterpret_cast<const float4*>(qkv_bias + col), c2 = *reinterpret_cast<const float4*>(qkv_bias + col + 4);
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w; c.x += c2.x; c.y += c2.y; c.z += c2.z; c.w += c2.w;
        }
        bf16x8 r;
        r[0] = (bf16_t)a.x; r[1] = (bf16_t)a.y; r[2] = (bf16_t)a.z; r[3] = (bf16_t)a.w;
        r[4] = (bf16_t)c.x; r[5] = (bf16_t)c.y; r[6] = (bf16_t)c.z; r[7] = (bf16_t)c.w;
        return r;
    };
    const bf16_t* kb = k + (int64_t)b * bsk * ldk + h * hd + 8 * dl;
    const bf16_t* vb = v + (int64_t)b * bsk * ldv + h * hd + 8 * dl;
    if (VLDS) {
        const int n_keys = appended ? Sk - 1 : Sk;    // the new key's row is not in the cache yet: its owner lanes write the image
        const int total = n_keys * cpk;
        const int valid_pieces = min(cpk, ((H - blockIdx.y * 4) * hd) >> 3);      // a last group of < 4 heads: stay inside the row
        const bf16_t* vsrc = v + (int64_t)b * bsk * ldv + blockIdx.y * 4 * hd;
        for (int base = __builtin_amdgcn_readfirstlane(wave) * 64; base < total; base += 64 * 4 * DEC_WPH) {
            const int c = base + lane;
            const int key = c / cpk, piece = c - key * cpk;
            if (c < total && piece < valid_pieces)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + (int64_t)key * ldv + piece * 8),
                                                 (__attribute__((address_space(3))) void*)(vimg + base * 16), 16, 0, 0);
        }
    }
    // key of (batch start j0, slot u): groups of KPI keys are dealt round-robin to the DEC_WPH waves of the head
    auto key_of = [&](int j0, int u) { return j0 + (u * DEC_WPH + part) * KPI + sub; };
    // the first batch of K goes out before anything that has to wait for the previous kernel's results (q and the new K / V row
    // below): those L2 round trips then run under the HBM round trip instead of in front of it
    bf16x8 kv0[DEC_U];
#pragma unroll
    for (int u = 0; u < DEC_U; ++u) {
        const int j = key_of(0, u);
        kv0[u] = (bf16x8){};
        if (active && j < Sk) kv0[u] = *reinterpret_cast<const bf16x8*>(kb + (int64_t)j * ldk);    // row Sk-1 may be stale: patched below
    }
    bf16x8 vpre[VREG ? DEC_U : 1];                    // VMODE 2: the first batch of V right behind it (same patch for row Sk-1)
    if (VREG) {
#pragma unroll
        for (int u = 0; u < DEC_U; ++u) {
            const int j = key_of(0, u);
            vpre[u] = (bf16x8){};
            if (active && j < Sk) vpre[u] = *reinterpret_cast<const bf16x8*>(vb + (int64_t)j * ldv);
        }
    }
    EAVQA_STAMP(1);
    // the new position: its one owner lane per 16-byte piece fetches (or sums up) the row and appends it to the cache
    bf16x8 knew = {}, vnew = {};
    bool own_new = false;
    if (appended) {
        const int rem = (Sk - 1) % STEP, grp = rem / KPI;
        own_new = active && (rem % KPI) == sub && (grp % DEC_WPH) == part;
        if (own_new) {
            if (qkv_part) {
                knew = from_part(H * hd + h * hd + 8 * dl);
                vnew = from_part(2 * H * hd + h * hd + 8 * dl);
            } else {
                knew = *reinterpret_cast<const bf16x8*>(k_new + (int64_t)b * ld_new + h * hd + 8 * dl);
                vnew = *reinterpret_cast<const bf16x8*>(v_new + (int64_t)b * ld_new + h * hd + 8 * dl);
            }
            *reinterpret_cast<bf16x8*>(const_cast<bf16_t*>(kb) + (int64_t)(Sk - 1) * ldk) = knew;
            *reinterpret_cast<bf16x8*>(const_cast<bf16_t*>(vb) + (int64_t)(Sk - 1) * ldv) = vnew;
            if (VLDS) *reinterpret_cast<bf16x8*>(vimg + ((Sk - 1) * cpk + hh * (hd >> 3) + dl) * 16) = vnew;
        }
    }
    EAVQA_STAMP(2);
    float qf[8];
    {
        bf16x8 t = {};
        if (active) t = qkv_part ? from_part(h * hd + 8 * dl) : *reinterpret_cast<const bf16x8*>(q + (int64_t)b * bsq * ldq + h * hd + 8 * dl);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[e] = (float)t[e];
    }
    auto load_k = [&](int j) -> bf16x8 {
        if (own_new && j == Sk - 1) return knew;
        return *reinterpret_cast<const bf16x8*>(kb + (int64_t)j * ldk);
    };
    auto load_v = [&](int j) -> bf16x8 {
        if (own_new && j == Sk - 1) return vnew;
        return *reinterpret_cast<const bf16x8*>(vb + (int64_t)j * ldv);
    };
    const int32_t* mrow = key_mask ? key_mask + (int64_t)b * ld_mask : nullptr;
    EAVQA_STAMP(3);

    auto score = [&](const bf16x8 (&kv)[DEC_U], int j0) {
#pragma unroll
        for (int u = 0; u < DEC_U; ++u) {
            const int j = key_of(j0, u);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d += qf[e] * (float)kv[u][e];
#pragma unroll
            for (int o = LPK >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
            if (head_ok && dl == 0 && j < Sk)
                sc[j] = (mrow && mrow[j] == 0) ? -FLT_MAX : d * scale + (rel_bias ? rel_bias[(int64_t)h * rel_ld + (j - (Sk - 1)) + rel_zero] : 0.f);
        }
    };
    // first batch: the registers loaded at kernel start, the stale new row patched in place (no copy: with 2 x 20 loads held a second
    // set of K registers spilled 137 VGPRs)
#pragma unroll
    for (int u = 0; u < DEC_U; ++u)
        if (own_new && key_of(0, u) == Sk - 1) kv0[u] = knew;
    score(kv0, 0);
    for (int j0 = STEP; j0 < Sk; j0 += STEP) {
        bf16x8 kv[DEC_U];
#pragma unroll
        for (int u = 0; u < DEC_U; ++u) {
            const int j = key_of(j0, u);
            kv[u] = (bf16x8){};
            if (active && j < Sk) kv[u] = load_k(j);
        }
        score(kv, j0);
    }
    EAVQA_STAMP(4);
    // without the image, the first batch of V is fetched under the exchange and the softmax
    bf16x8 v0[VMODE == 0 ? DEC_U : 1];
    if (VMODE == 0) {
#pragma unroll
        for (int u = 0; u < DEC_U; ++u) {
            const int j = key_of(0, u);
            v0[u] = (bf16x8){};
            if (active && j < Sk) v0[u] = load_v(j);
        }
    } else if (VLDS) {
        __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00);      // vmcnt(0): this wave's share of the V image has landed
    }
    EAVQA_STAMP(5);
    __syncthreads();
    EAVQA_STAMP(6);
    float mx = -FLT_MAX;
    for (int j = lane; j < Sk; j += 64) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < Sk; j += 64) sum += __expf(sc[j] - mx);
    sum = wave_sum(sum);

    EAVQA_STAMP(7);
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto accumulate = [&](const bf16x8 (&vv)[DEC_U], int j0) {
#pragma unroll
        for (int u = 0; u < DEC_U; ++u) {
            const int j = key_of(j0, u);
            const float pj = (active && j < Sk) ? __expf(sc[j] - mx) : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += pj * (float)vv[u][e];
        }
    };
    if (VMODE == 0) accumulate(reinterpret_cast<const bf16x8 (&)[DEC_U]>(v0), 0);
    if (VREG) {
#pragma unroll
        for (int u = 0; u < DEC_U; ++u)
            if (own_new && key_of(0, u) == Sk - 1) vpre[u] = vnew;
        accumulate(reinterpret_cast<const bf16x8 (&)[DEC_U]>(vpre), 0);
    }
    for (int j0 = VLDS ? 0 : STEP; j0 < Sk; j0 += STEP) {
        bf16x8 vv[DEC_U];
#pragma unroll
        for (int u = 0; u < DEC_U; ++u) {
            const int j = key_of(j0, u);
            vv[u] = (bf16x8){};
            if (active && j < Sk)
                vv[u] = VLDS ? *reinterpret_cast<const bf16x8*>(vimg + (j * cpk + hh * (hd >> 3) + dl) * 16) : load_v(j);
        }
        accumulate(vv, j0);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int off = LPK; off < 64; off <<= 1) o[e] += __shfl_xor(o[e], off, 64);
    EAVQA_STAMP(8);
    if (DEC_WPH == 1) {                               // the wave holds its head's whole output: no exchange
        if (sub == 0 && active) {
            const float inv = 1.f / sum;
            bf16x8 r;
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = (bf16_t)(o[e] * inv);
            *reinterpret_cast<bf16x8*>(out + (int64_t)b * bsq * ldo + h * hd + 8 * dl) = r;
        }
        if (lse && head_ok && lane == 0) lse[(int64_t)b * H + h] = mx + __logf(sum);
        EAVQA_STAMP(9);
        return;
    }
    __syncthreads();                                  // every wave is done reading the scores: reuse nothing of theirs
    if (sub == 0 && active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) opart[8 * dl + e] = o[e];
    }
    __syncthreads();
    if (part == 0 && sub == 0 && active) {
        const float* p0 = dec_sc + 4 * Sk + hh * DEC_WPH * 128 + 8 * dl;
        const float inv = 1.f / sum;
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = p0[e];
#pragma unroll
            for (int w = 1; w < DEC_WPH; ++w) t += p0[w * 128 + e];
            r[e] = (bf16_t)(t * inv);
        }
        *reinterpret_cast<bf16x8*>(out + (int64_t)b * bsq * ldo + h * hd + 8 * dl) = r;
    }
    if (lse && head_ok && part == 0 && lane == 0) lse[(int64_t)b * H + h] = mx + __logf(sum);
    EAVQA_STAMP(9);
}

bool decode_supported(int dtype, int Sq, int Sk, int hd, const int32_t* cu, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo) {
    return dtype == EAVQA_BF16 && Sq == 1 && !cu && hd % 8 == 0 && hd <= 128 && Sk <= 3584 && (ldq % 8 == 0) && (ldk % 8 == 0) &&
           (ldv % 8 == 0) && (ldo % 8 == 0);
}

}  // namespace

static int attention_fwd_impl(int dtype, int B, int H, int Sq, int Sk, int hd,
                              const void* q, int64_t ldq, const void* k, int64_t ldk,
                              const void* v, int64_t ldv, void* o, int64_t ldo,
                              int64_t q_batch_rows, int64_t kv_batch_rows,
                              const int32_t* key_mask, int64_t ld_mask, const int32_t* cu_seqlens, int causal,
                              float scale, float* lse, void* stream, int path,
                              const void* k_new, const void* v_new, int64_t ld_new,
                              const float* qkv_part = nullptr, int ks = 0, const float* qkv_bias = nullptr, int part_cols = 0,
                              const float* rel_bias = nullptr, int64_t rel_ld = 0, int rel_zero = 0) {
    const bool g_force_valu = (path & 1) != 0;      // include/eavqa_test.h: bf16 on the vector-ALU kernels
    if ((!q && !qkv_part) || !k || !v || !o) return EAVQA_E_ARG;
    int rc = check_common(dtype, B, H, Sq, Sk, hd);
    if (rc) return rc;
    if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4) return EAVQA_E_ALIGN;
    AttnParams p = {};
    p.q = q; p.k = k; p.v = v; p.out = o; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.key_mask = key_mask; p.lse = lse; p.B = B; p.H = H; p.Sq = Sq; p.Sk = Sk; p.hd = hd;
    p.causal = causal; p.scale = scale;
    p.cu = cu_seqlens; p.stat_ld = Sq;
    if (cu_seqlens && (key_mask || Sq != Sk)) return EAVQA_E_ARG;
    p.ld_mask = ld_mask > 0 ? ld_mask : Sk;
    if (p.ld_mask < Sk) return EAVQA_E_ARG;
    p.bsq = q_batch_rows > 0 ? q_batch_rows : Sq;
    p.bsk = kv_batch_rows > 0 ? kv_batch_rows : Sk;
    if (p.bsq < Sq || p.bsk < Sk) return EAVQA_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (decode_supported(dtype, Sq, Sk, hd, cu_seqlens, ldq, ldk, ldv, ldo) && !g_force_valu && (qkv_part || eavqa_aligned16(q)) && eavqa_aligned16(k) &&
        eavqa_aligned16(v) && eavqa_aligned16(o)) {
        const dim3 grid(B, (H + 3) / 4);
        const int blocks = B * ((H + 3) / 4);
        const int kpi = hd <= 64 ? 8 : 4;
#define EAVQA_DEC2(LPK, WPH, U, VM)                                                                                          \
    hipLaunchKernelGGL((attn_decode_kernel<LPK, WPH, U, VM>), grid, dim3(256 * WPH), lds, s, reinterpret_cast<const bf16_t*>(q), ldq,  \
                       reinterpret_cast<const bf16_t*>(k), ldk, reinterpret_cast<const bf16_t*>(v), ldv,                      \
                       reinterpret_cast<bf16_t*>(o), ldo, p.bsq, p.bsk, key_mask, p.ld_mask, lse, H, Sk, hd, scale,                  \
                       reinterpret_cast<const bf16_t*>(k_new), reinterpret_cast<const bf16_t*>(v_new), ld_new, qkv_part, ks, qkv_bias,        \
                       part_cols ? part_cols : 3 * H * hd, part_cols == 0 || part_cols == 3 * H * hd, rel_bias, rel_ld, rel_zero)
        // one workgroup per CU and a head's keys within two batches of one or two waves: everything in registers, one HBM round trip
        // (path bit 4, include/eavqa_test.h: keep the round-3 LDS-image kernel for A / B measurements and its parity tests)
        // Taken where it measured faster (profiles/round4_decode_attention.md): one wave per head (<= 80 / 40 keys: 9.5 -> 6.6 us) and two
        // waves x 10 loads (T0-3B cross-attention, 150 keys x 64: 16.9 -> 13.9 us).  Two waves x 20 loads (OPT-2.7B, 160 keys x 80) landed
        // its 204 KB per CU no sooner than the LDS-image kernel (22.6 against 21.1 us): path bit 5 selects it for measurements only.
        if (blocks <= 256 && Sk <= kpi * 2 * ((path & 32) ? 20 : 10) && !(path & 16)) {
            const int wph = Sk <= kpi * 10 ? 1 : 2, u = Sk <= kpi * wph * 10 ? 10 : 20;
            const size_t lds = ((size_t)4 * Sk + 4 * wph * 128) * sizeof(float);
            if (hd <= 64) {
                if (wph == 1) EAVQA_DEC2(8, 1, 10, 2); else if (u == 10) EAVQA_DEC2(8, 2, 10, 2); else EAVQA_DEC2(8, 2, 20, 2);
            } else {
                if (wph == 1) EAVQA_DEC2(16, 1, 10, 2); else if (u == 10) EAVQA_DEC2(16, 2, 10, 2); else EAVQA_DEC2(16, 2, 20, 2);
            }
            EAVQA_LAUNCH_CHECK();
            return EAVQA_OK;
        }
        const int wph = blocks <= 256 ? 4 : (blocks <= 512 ? 2 : 1);
        const size_t v_image = (size_t)Sk * 4 * hd * 2;
        // the V image rides in LDS only when the WHOLE request (scores + per-wave scratch + image) fits the 150 KiB the kernel opts into;
        // otherwise the register route (small head dims at long Sk: hd = 16, Sk ~ 1024 asked for 152 KiB and failed the launch)
        const size_t lds_base = ((size_t)4 * Sk + 4 * wph * 128) * sizeof(float);
        const bool vlds = wph == 4 && lds_base + v_image <= 150 * 1024;
        const size_t lds = lds_base + (vlds ? v_image : 0);
        if (vlds) {
            static std::atomic<bool> configured[2];              // zero-initialised; concurrent first calls only repeat an idempotent call
            const int slot = hd <= 64 ? 0 : 1;
            if (!configured[slot].load(std::memory_order_acquire)) {
                const void* fn = hd <= 64 ? reinterpret_cast<const void*>(attn_decode_kernel<8, 4, 10, 1>)
                                          : reinterpret_cast<const void*>(attn_decode_kernel<16, 4, 10, 1>);
                if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return EAVQA_E_LAUNCH;
                configured[slot].store(true, std::memory_order_release);
            }
        }
#define EAVQA_DEC(LPK)                                                                                                       \
    if (vlds) EAVQA_DEC2(LPK, 4, 10, 1); else if (wph == 4) EAVQA_DEC2(LPK, 4, 10, 0); else if (wph == 2) EAVQA_DEC2(LPK, 2, 10, 0); \
    else EAVQA_DEC2(LPK, 1, 10, 0)
        if (hd <= 64) { EAVQA_DEC(8); } else { EAVQA_DEC(16); }
#undef EAVQA_DEC
#undef EAVQA_DEC2
        EAVQA_LAUNCH_CHECK();
        return EAVQA_OK;
    }
    if (k_new || v_new || qkv_part) return EAVQA_E_SHAPE;          // the append forms exist for the decode kernel only
    if (rel_bias) return EAVQA_E_SHAPE;                            // (eavqa_attention_fwd_rel routes a bias to the tiled kernels itself)
    const bool wide = eavqa_attn_mfma::supported_wide(hd, Sq, Sk) && !(ldq % 8 || ldk % 8 || ldv % 8);
    if (dtype == EAVQA_BF16 && (eavqa_attn_mfma::supported(hd) || wide) && !g_force_valu) {
        eavqa_attn_mfma::Params m = {};
        m.q = q; m.k = k; m.v = v; m.out = o; m.ldq = ldq; m.ldk = ldk; m.ldv = ldv; m.ldo = ldo;
        m.key_mask = key_mask; m.ld_mask = p.ld_mask; m.cu = cu_seqlens; m.lse = lse;
        m.B = B; m.H = H; m.Sq = Sq; m.Sk = Sk; m.hd = hd; m.causal = causal; m.stat_ld = Sq;
        m.bsq = p.bsq; m.bsk = p.bsk; m.scale = scale;
        if (wide) return eavqa_attn_mfma::run_wide(0, m, s);
        // K / V resident in LDS (the CLIP tower: one workgroup per (image, head)); path bit 2 keeps the streamed-tile kernel (A / B, tests)
        if (!(path & 4) && (Sk > 64 || (path & 8)) && eavqa_attn_mfma::resident_supported(m)) return eavqa_attn_mfma::run_resident(m, s);
        return eavqa_attn_mfma::run(0, m, s);
    }
    return dtype == EAVQA_F32 ? dispatch<float>(K_FWD, p, s) : dispatch<bf16_t>(K_FWD, p, s);
}

extern "C" int eavqa_attention_fwd_ex(int dtype, int B, int H, int Sq, int Sk, int hd,
                                   const void* q, int64_t ldq, const void* k, int64_t ldk,
                                   const void* v, int64_t ldv, void* o, int64_t ldo,
                                   int64_t q_batch_rows, int64_t kv_batch_rows,
                                   const int32_t* key_mask, int64_t ld_mask, const int32_t* cu_seqlens, int causal,
                                   float scale, float* lse, void* stream, int path) {
    return attention_fwd_impl(dtype, B, H, Sq, Sk, hd, q, ldq, k, ldk, v, ldv, o, ldo, q_batch_rows, kv_batch_rows, key_mask, ld_mask,
                              cu_seqlens, causal, scale, lse, stream, path, nullptr, nullptr, 0);
}

extern "C" int eavqa_attention_decode(int dtype, int B, int H, int Sk, int hd, const void* q, int64_t ldq, void* k_cache, int64_t ldk,
                                      void* v_cache, int64_t ldv, int64_t kv_batch_rows, const void* k_new, const void* v_new,
                                      int64_t ld_new, void* o, int64_t ldo, const int32_t* key_mask, int64_t ld_mask, float scale,
                                      void* stream) {
    if (!k_new || !v_new) return EAVQA_E_ARG;
    if (ld_new % 8 || !eavqa_aligned16(k_new) || !eavqa_aligned16(v_new)) return EAVQA_E_ALIGN;
    return attention_fwd_impl(dtype, B, H, 1, Sk, hd, q, ldq, k_cache, ldk, v_cache, ldv, o, ldo, 1, kv_batch_rows, key_mask, ld_mask,
                              nullptr, 1, scale, nullptr, stream, 0, k_new, v_new, ld_new);
}

extern "C" int eavqa_attention_decode_splitk(int dtype, int B, int H, int Sk, int hd, const float* qkv_partials, int ks, const float* qkv_bias,
                                             void* k_cache, int64_t ldk, void* v_cache, int64_t ldv, int64_t kv_batch_rows, void* o, int64_t ldo,
                                             const int32_t* key_mask, int64_t ld_mask, float scale, void* stream) {
    if (!qkv_partials || ks <= 0) return EAVQA_E_ARG;
    if ((H * hd) % 4 || !eavqa_aligned16(qkv_partials) || (qkv_bias && !eavqa_aligned16(qkv_bias))) return EAVQA_E_ALIGN;
    return attention_fwd_impl(dtype, B, H, 1, Sk, hd, nullptr, 8, k_cache, ldk, v_cache, ldv, o, ldo, 1, kv_batch_rows, key_mask, ld_mask,
                              nullptr, 1, scale, nullptr, stream, 0, nullptr, nullptr, 0, qkv_partials, ks, qkv_bias);
}

#ifdef EAVQA_ATTN_STAMPS
extern "C" __attribute__((visibility("default"))) int eavqa_attn_stamps_read(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(eavqa_attn_stamps), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -5;
}
#endif

extern "C" int eavqa_attention_decode_splitk_rel(int dtype, int B, int H, int Sk, int hd, const float* partials, int ks, int part_cols,
                                                 void* k, int64_t ldk, void* v, int64_t ldv, int64_t kv_batch_rows, void* o, int64_t ldo,
                                                 const int32_t* key_mask, int64_t ld_mask, float scale, const float* rel_bias, int64_t rel_ld,
                                                 int rel_zero, void* stream) {
    if (!partials || ks <= 0) return EAVQA_E_ARG;
    if (part_cols != H * hd && part_cols != 3 * H * hd) return EAVQA_E_SHAPE;
    if ((H * hd) % 4 || !eavqa_aligned16(partials)) return EAVQA_E_ALIGN;
    if (rel_bias && (rel_zero < Sk - 1 || rel_ld < rel_zero + 1)) return EAVQA_E_ARG;     // offsets -(Sk - 1) .. 0 are read
    return attention_fwd_impl(dtype, B, H, 1, Sk, hd, nullptr, 8, k, ldk, v, ldv, o, ldo, 1, kv_batch_rows, key_mask, ld_mask,
                              nullptr, 1, scale, nullptr, stream, 0, nullptr, nullptr, 0, partials, ks, nullptr, part_cols, rel_bias, rel_ld, rel_zero);
}

extern "C" int eavqa_attention_fwd(int dtype, int B, int H, int Sq, int Sk, int hd,
                                   const void* q, int64_t ldq, const void* k, int64_t ldk,
                                   const void* v, int64_t ldv, void* o, int64_t ldo,
                                   int64_t q_batch_rows, int64_t kv_batch_rows,
                                   const int32_t* key_mask, int64_t ld_mask, const int32_t* cu_seqlens, int causal,
                                   float scale, float* lse, void* stream) {
    return eavqa_attention_fwd_ex(dtype, B, H, Sq, Sk, hd, q, ldq, k, ldk, v, ldv, o, ldo, q_batch_rows, kv_batch_rows, key_mask,
                                  ld_mask, cu_seqlens, causal, scale, lse, stream, 0);
}

extern "C" int eavqa_attention_bwd_ex(int dtype, int B, int H, int Sq, int Sk, int hd,
                                   const void* q, int64_t ldq, const void* k, int64_t ldk,
                                   const void* v, int64_t ldv, const void* o, int64_t ldo,
                                   const void* d_o, int64_t lddo,
                                   void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                                   const int32_t* key_mask, const int32_t* cu_seqlens, int causal, float scale,
                                   const float* lse, float* delta, void* stream, int path) {
    const bool g_force_valu = (path & 1) != 0;      // bf16 on the vector-ALU kernels
    const bool g_split_bwd = (path & 2) != 0;       // two-kernel backward even when the problem is one tile
    if (!q || !k || !v || !o || !d_o || !dq || !dk || !dv || !lse || !delta) return EAVQA_E_ARG;
    int rc = check_common(dtype, B, H, Sq, Sk, hd);
    if (rc) return rc;
    if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4 || lddo % 4 || lddq % 4 || lddk % 4 || lddv % 4) return EAVQA_E_ALIGN;
    AttnParams p = {};
    p.q = q; p.k = k; p.v = v; p.o = o; p.d_o = d_o; p.dq = dq; p.dk = dk; p.dv = dv;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.key_mask = key_mask; p.lse = const_cast<float*>(lse); p.delta = delta;
    p.B = B; p.H = H; p.Sq = Sq; p.Sk = Sk; p.hd = hd; p.causal = causal; p.scale = scale;
    p.bsq = Sq; p.bsk = Sk; p.ld_mask = Sk; p.cu = cu_seqlens; p.stat_ld = Sq;
    if (cu_seqlens && (key_mask || Sq != Sk)) return EAVQA_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool wide = eavqa_attn_mfma::supported_wide(hd, Sq, Sk) && !(ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || lddo % 8);
    if (dtype == EAVQA_BF16 && (eavqa_attn_mfma::supported(hd) || wide) && !g_force_valu) {
        eavqa_attn_mfma::Params m = {};
        m.q = q; m.k = k; m.v = v; m.o = o; m.d_o = d_o; m.dq = dq; m.dk = dk; m.dv = dv;
        m.ldq = ldq; m.ldk = ldk; m.ldv = ldv; m.ldo = ldo; m.lddo = lddo; m.lddq = lddq; m.lddk = lddk; m.lddv = lddv;
        m.key_mask = key_mask; m.ld_mask = Sk; m.cu = cu_seqlens; m.lse = const_cast<float*>(lse); m.delta = delta;
        m.B = B; m.H = H; m.Sq = Sq; m.Sk = Sk; m.hd = hd; m.causal = causal; m.stat_ld = Sq;
        m.bsq = Sq; m.bsk = Sk; m.scale = scale;
        if (wide) return eavqa_attn_mfma::run_wide(3, m, s);
        m.fused_padded = (path & 4) != 0;               // path bit 2: the round-2 padded-pitch one-tile kernel also for hd = 64
        if (Sq <= eavqa_attn_mfma::TILE && Sk <= eavqa_attn_mfma::TILE && !g_split_bwd) return eavqa_attn_mfma::run(3, m, s);
        rc = eavqa_attn_mfma::run(1, m, s);
        if (rc) return rc;
        return eavqa_attn_mfma::run(2, m, s);
    }
    rc = dtype == EAVQA_F32 ? dispatch<float>(K_DQ, p, s) : dispatch<bf16_t>(K_DQ, p, s);
    if (rc) return rc;
    return dtype == EAVQA_F32 ? dispatch<float>(K_DKV, p, s) : dispatch<bf16_t>(K_DKV, p, s);
}

extern "C" int eavqa_attention_bwd(int dtype, int B, int H, int Sq, int Sk, int hd,
                                   const void* q, int64_t ldq, const void* k, int64_t ldk,
                                   const void* v, int64_t ldv, const void* o, int64_t ldo,
                                   const void* d_o, int64_t lddo,
                                   void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                                   const int32_t* key_mask, const int32_t* cu_seqlens, int causal, float scale,
                                   const float* lse, float* delta, void* stream) {
    return eavqa_attention_bwd_ex(dtype, B, H, Sq, Sk, hd, q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, dq, lddq, dk, lddk, dv, lddv,
                                  key_mask, cu_seqlens, causal, scale, lse, delta, stream, 0);
}

// ---------------------------------------------------------------------------------------------------- T5 relative-position bias
// eavqa_attention_fwd / _bwd with an additive per-head bias that depends on (key position - query position) only: T5's relative
// attention bias (HF:models/t5/modeling_t5.py:217-279 - compute_bias: values[h][q][k] = table[bucket(k - q)][h], shared by every layer
// of a stack) handed over as rel_bias[h * rel_ld + (k - q) + rel_zero], float32, q counted from the END of the keys when Sq < Sk (a
// cached decode step: query i sits at position i + Sk - Sq).  fp32 arithmetic on the vector-ALU kernels for both storage types.
extern "C" int eavqa_attention_fwd_rel(int dtype, int B, int H, int Sq, int Sk, int hd, const void* q, int64_t ldq, const void* k, int64_t ldk,
                                       const void* v, int64_t ldv, void* o, int64_t ldo, int64_t q_batch_rows, int64_t kv_batch_rows,
                                       const int32_t* key_mask, int64_t ld_mask, int causal, float scale, const float* rel_bias,
                                       int64_t rel_ld, int rel_zero, float* lse, void* stream) {
    if (!q || !k || !v || !o) return EAVQA_E_ARG;
    int rc = check_common(dtype, B, H, Sq, Sk, hd);
    if (rc) return rc;
    if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4) return EAVQA_E_ALIGN;
    if (rel_bias && (rel_zero < Sk - 1 || rel_ld < rel_zero + Sk)) return EAVQA_E_ARG;       // the table must span -(Sk - 1) .. Sk - 1
    AttnParams p = {};
    p.q = q; p.k = k; p.v = v; p.out = o; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.key_mask = key_mask; p.lse = lse; p.B = B; p.H = H; p.Sq = Sq; p.Sk = Sk; p.hd = hd; p.causal = causal; p.scale = scale;
    p.stat_ld = Sq; p.ld_mask = ld_mask > 0 ? ld_mask : Sk;
    if (p.ld_mask < Sk) return EAVQA_E_ARG;
    p.bsq = q_batch_rows > 0 ? q_batch_rows : Sq;
    p.bsk = kv_batch_rows > 0 ? kv_batch_rows : Sk;
    if (p.bsq < Sq || p.bsk < Sk) return EAVQA_E_ARG;
    p.rel_bias = rel_bias; p.rel_ld = rel_ld; p.rel_zero = rel_zero;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // bf16 at an MFMA head size: the streamed-tile MFMA kernel adds the bias to its score tile (T0_3B few-shot: the vector-ALU kernel
    // was 27 % of the GPU time, 71 us per call); without a bias this is eavqa_attention_fwd (T5's cross-attention)
    if (dtype == EAVQA_BF16 && !rel_bias)
        return attention_fwd_impl(dtype, B, H, Sq, Sk, hd, q, ldq, k, ldk, v, ldv, o, ldo, q_batch_rows, kv_batch_rows, key_mask, ld_mask,
                                  nullptr, causal, scale, lse, stream, 0, nullptr, nullptr, 0);
    if (dtype == EAVQA_BF16 && eavqa_attn_mfma::supported(hd) && !(ldq % 8 || ldk % 8 || ldv % 8) && eavqa_aligned16(q) && eavqa_aligned16(k) &&
        eavqa_aligned16(v)) {
        eavqa_attn_mfma::Params m = {};
        m.q = q; m.k = k; m.v = v; m.out = o; m.ldq = ldq; m.ldk = ldk; m.ldv = ldv; m.ldo = ldo;
        m.key_mask = key_mask; m.ld_mask = p.ld_mask; m.cu = nullptr; m.lse = lse;
        m.B = B; m.H = H; m.Sq = Sq; m.Sk = Sk; m.hd = hd; m.causal = causal; m.stat_ld = Sq;
        m.bsq = p.bsq; m.bsk = p.bsk; m.scale = scale;
        m.rel_bias = rel_bias; m.rel_ld = rel_ld; m.rel_zero = rel_zero;
        return eavqa_attn_mfma::run(0, m, s);
    }
    return dtype == EAVQA_F32 ? dispatch<float>(K_FWD, p, s) : dispatch<bf16_t>(K_FWD, p, s);
}

extern "C" int eavqa_attention_bwd_rel(int dtype, int B, int H, int Sq, int Sk, int hd, const void* q, int64_t ldq, const void* k, int64_t ldk,
                                       const void* v, int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                                       void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                                       const int32_t* key_mask, int causal, float scale, const float* rel_bias, int64_t rel_ld,
                                       int rel_zero, const float* lse, float* delta, void* stream) {
    if (!q || !k || !v || !o || !d_o || !dq || !dk || !dv || !lse || !delta) return EAVQA_E_ARG;
    int rc = check_common(dtype, B, H, Sq, Sk, hd);
    if (rc) return rc;
    if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4 || lddo % 4 || lddq % 4 || lddk % 4 || lddv % 4) return EAVQA_E_ALIGN;
    if (rel_bias && (rel_zero < Sk - 1 || rel_ld < rel_zero + Sk)) return EAVQA_E_ARG;
    AttnParams p = {};
    p.q = q; p.k = k; p.v = v; p.o = o; p.d_o = d_o; p.dq = dq; p.dk = dk; p.dv = dv;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.key_mask = key_mask; p.lse = const_cast<float*>(lse); p.delta = delta;
    p.B = B; p.H = H; p.Sq = Sq; p.Sk = Sk; p.hd = hd; p.causal = causal; p.scale = scale;
    p.bsq = Sq; p.bsk = Sk; p.ld_mask = Sk; p.stat_ld = Sq;
    p.rel_bias = rel_bias; p.rel_ld = rel_ld; p.rel_zero = rel_zero;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // bf16 at an MFMA head size: the matrix-core backward kernels recompute P with the bias added (round 4; the vector-ALU kernels below
    // were 12 % of a T0_3B training step).  Without a bias this is eavqa_attention_bwd (T5's cross-attention).
    if (dtype == EAVQA_BF16 && eavqa_attn_mfma::supported(hd) && !(ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || lddo % 8 || lddq % 8 || lddk % 8 || lddv % 8) &&
        eavqa_aligned16(q) && eavqa_aligned16(k) && eavqa_aligned16(v) && eavqa_aligned16(o) && eavqa_aligned16(d_o) && eavqa_aligned16(dq) &&
        eavqa_aligned16(dk) && eavqa_aligned16(dv)) {
        eavqa_attn_mfma::Params m = {};
        m.q = q; m.k = k; m.v = v; m.o = o; m.d_o = d_o; m.dq = dq; m.dk = dk; m.dv = dv;
        m.ldq = ldq; m.ldk = ldk; m.ldv = ldv; m.ldo = ldo; m.lddo = lddo; m.lddq = lddq; m.lddk = lddk; m.lddv = lddv;
        m.key_mask = key_mask; m.ld_mask = Sk; m.cu = nullptr; m.lse = const_cast<float*>(lse); m.delta = delta;
        m.B = B; m.H = H; m.Sq = Sq; m.Sk = Sk; m.hd = hd; m.causal = causal; m.stat_ld = Sq;
        m.bsq = Sq; m.bsk = Sk; m.scale = scale;
        m.rel_bias = rel_bias; m.rel_ld = rel_ld; m.rel_zero = rel_zero;
        if (Sq <= eavqa_attn_mfma::TILE && Sk <= eavqa_attn_mfma::TILE) return eavqa_attn_mfma::run(3, m, s);
        rc = eavqa_attn_mfma::run(1, m, s);
        if (rc) return rc;
        return eavqa_attn_mfma::run(2, m, s);
    }
    rc = dtype == EAVQA_F32 ? dispatch<float>(K_DQ, p, s) : dispatch<bf16_t>(K_DQ, p, s);
    if (rc) return rc;
    return dtype == EAVQA_F32 ? dispatch<float>(K_DKV, p, s) : dispatch<bf16_t>(K_DKV, p, s);
}
