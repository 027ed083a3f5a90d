// One decode step of ALL layers in ONE persistent kernel (eavqa_lm_decode_persistent, used by eavqa_lm_block_forward for Sq = 1).
//
// Why: a decode step is 8 dependent kernels per layer, each 5-20 us; every one pays launch + ramp + drain and cannot start
// streaming its weights before its predecessor has drained, although the weights do not depend on it.  Here
// 256 co-resident workgroups (one per CU, hipLaunchCooperativeKernel) walk the phases of every layer with a device-wide barrier in
// between, and a workgroup issues the first 16 KiB-per-wave of its NEXT GEMM's weights BEFORE it waits at the barrier.
//
// What makes the barrier affordable (tools/barrier_probe.hip, profiles/round2_decode.md section 4): the eight XCDs' L2s are not
// coherent; the agent-scope fences that make them so cost 15 us per barrier.  Relaxed agent-scope ATOMIC loads / stores of the
// exchanged words (served at the level all XCDs share) are correct WITHOUT fences: 4.8 us per barrier including 4 KiB of data per
// workgroup.  So every buffer one workgroup writes and another reads inside this kernel (LayerNorm outputs, partial sums, the
// attention output, the FFN activation) is accessed ONLY through cld* / cst* below; weights and the K / V cache rows of earlier
// steps are ordinary loads.
//
// STATUS: correct (tests/test_model_gpu.py::test_persistent_decode_step_matches_the_kernel_per_phase_route) but SLOWER than the
// kernel-per-phase route - 6.6 against 2.93 ms per OPT-2.7B step at B = 32.  Timeline of one layer as workgroup 0 sees it
// (profiles/round2_decode.md section 5): the four GEMM phases take 8 / 6 / 28 (two units) / 17 us, attention 26 us, but the eight
// barriers cost 7-18 us each once 256 workgroups with uneven work arrive at them (70 us per layer), and a LayerNorm row read through
// coherent 8-byte loads takes 17 us (34 us per layer).  A barrier costs as much as the kernel boundary it replaces, so the weight
// prefetch across it has nothing to win back.  Reached only through eavqa_lm_block_forward_ex(route = 2).
//
// Arithmetic: the same split-K plans, MFMA order and fixed-order reductions as the kernels of decode.hip / attention.hip, so the
// GEMM / LayerNorm / finish phases reproduce them bit for bit; the attention phase runs 8 waves per (sample, 4 heads) instead of
// 16, which changes the order of its fp32 P.V partial sums (differences at the 1e-7 level before the bf16 rounding).
#include "common.h"
#include "decode_layer.h"

namespace {
typedef unsigned long long u64;
typedef uint32_t u32;

__device__ __forceinline__ u64 cld64(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cst64(void* p, u64 v) {
    __hip_atomic_store(reinterpret_cast<u64*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cstf(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<u32*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 cld4f(const float* p) {          // 16-byte aligned
    const u64 a = cld64(p), b = cld64(p + 2);
    return make_float4(__uint_as_float((u32)a), __uint_as_float((u32)(a >> 32)), __uint_as_float((u32)b), __uint_as_float((u32)(b >> 32)));
}
__device__ __forceinline__ void cst4f(float* p, const float4& v) {
    cst64(p, (u64)__float_as_uint(v.x) | ((u64)__float_as_uint(v.y) << 32));
    cst64(p + 2, (u64)__float_as_uint(v.z) | ((u64)__float_as_uint(v.w) << 32));
}
__device__ __forceinline__ u64 pack4bf(const float4& v) {         // the conversion elem<bf16_t>::st4 performs
    union { bf16_t h[4]; u64 u; } t;
    t.h[0] = (bf16_t)v.x; t.h[1] = (bf16_t)v.y; t.h[2] = (bf16_t)v.z; t.h[3] = (bf16_t)v.w;
    return t.u;
}

struct PL {
    const eavqa_lm_layer_t* layers;
    int n_layer, E, H, F, act, B, row0, S_max, Sk, hd;
    float eps, scale;
    float* x;                           // [B, E] fp32 residual stream in / out
    const int32_t* key_mask; int64_t ld_mask;
    float* x1; bf16_t* a; bf16_t* ctx; bf16_t* f; float* part; float* part2;
    unsigned* sync;                     // [0] arrivals, [1] error flag
    int ks_qkv, ks_o, ks_fc1, ks_fc2;
};

constexpr int NT = 512, NWAVE = 8, GU = 16, COLS = 128;

// ---- device-wide barrier: every wave has its stores acknowledged, one arrival per workgroup, bounded spin (never hangs)
__device__ __forceinline__ bool grid_barrier(const PL& p, unsigned& epoch, int* flag) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    epoch += 1;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(p.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * gridDim.x;
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(p.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 21) || ((spins & 1023u) == 0 && __hip_atomic_load(p.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { ok = 0; break; }
        }
        if (!ok) __hip_atomic_store(p.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;
}

__device__ __forceinline__ float block_sum8(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

// ---- x = x_in + bias + sum_s P[s]; y = LayerNorm(x) (ln_splitk_kernel of decode.hip, same element-to-thread map and summation order)
__device__ __noinline__ void ln_row(int row, int rows, int cols, const float* x_in, const float* P, int ks, const float* bias, float* x_out,
                       const float* gamma, const float* beta, float eps, bf16_t* y, float* red) {
    constexpr int NV = 2;                                           // cols <= 4096
    const int tid = threadIdx.x, nv = cols >> 2;
    float4 v[NV], gm[NV], bt[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid + NT * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        gm[i] = v[i]; bt[i] = v[i];
        if (c < nv) {
            v[i] = cld4f(x_in + (int64_t)row * cols + 4 * c);
            gm[i] = *reinterpret_cast<const float4*>(gamma + 4 * c);
            bt[i] = *reinterpret_cast<const float4*>(beta + 4 * c);
            if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + 4 * c); v[i].x += b.x; v[i].y += b.y; v[i].z += b.z; v[i].w += b.w; }
        }
    }
    const int64_t slice = (int64_t)rows * cols;
    const float* prow = P + (int64_t)row * cols;
    for (int sl = 0; sl < ks; sl += 8) {
        float4 t[8][NV];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = tid + NT * i;
                t[u][i] = (c < nv && sl + u < ks) ? cld4f(prow + (sl + u) * slice + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        const int c = tid + NT * i;
        if (c < nv) {
            if (x_out) cst4f(x_out + (int64_t)row * cols + 4 * c, v[i]);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mu = block_sum8(s, red) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid + NT * i;
        if (c < nv) {
            const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rs = rsqrtf(block_sum8(q, red) / (float)cols + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid + NT * i;
        if (c < nv) {
            float4 o;
            o.x = (v[i].x - mu) * rs * gm[i].x + bt[i].x;
            o.y = (v[i].y - mu) * rs * gm[i].y + bt[i].y;
            o.z = (v[i].z - mu) * rs * gm[i].z + bt[i].z;
            o.w = (v[i].w - mu) * rs * gm[i].w + bt[i].w;
            cst64(y + (int64_t)row * cols + 4 * c, pack4bf(o));
        }
    }
}

// ---- split-K GEMM unit: 128 columns x one K slice, M <= 32 rows (gemm_bf16_splitk_kernel<2, 1, 8, 16> of decode.hip)
__device__ __forceinline__ int fswz(int row, int kc) { return row * 64 + ((kc ^ ((-(row >> 2)) & 3)) << 4); }

struct GemmUnit { const bf16_t* bp; int nsteps, n, slice; bool valid; };

// first GU k-steps of this wave's 16 weight rows: issued before the barrier in front of the phase
__device__ __forceinline__ GemmUnit gemm_issue(bf16x8 (&bf)[GU], const bf16_t* W, int N, int K, int ks, int unit, int units) {
    GemmUnit g;
    g.valid = unit < units;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = g.valid ? unit / ks : 0;
    g.slice = g.valid ? unit - grp * ks : 0;
    const int KS = K / ks;
    g.nsteps = KS >> 5;
    g.n = grp * COLS + wave * 16 + (lane & 15);
    g.bp = W + (int64_t)min(g.n, N - 1) * K + g.slice * KS + 8 * (lane >> 4);
    if (g.valid) {
#pragma unroll
        for (int u = 0; u < GU; ++u) bf[u] = *reinterpret_cast<const bf16x8*>(g.bp + 32 * min(u, g.nsteps - 1));
    }
    return g;
}

__device__ void gemm_run(const GemmUnit& g, bf16x8 (&bf)[GU], const bf16_t* A, int M, int N, int K, int ks, float* P, char* smem) {
    if (!g.valid) return;
    constexpr int MF = 2, TILE = 16 * MF * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int KS = K / ks, k0 = g.slice * KS, nsteps = g.nsteps;
    // the A slice comes from a buffer another workgroup has just written: coherent 8-byte loads, then the swizzled LDS image
    const int total = nsteps * 64 * MF;
    for (int c = tid; c < total; c += NT) {
        const int t = c / (64 * MF), within = c % (64 * MF);
        const int row = within >> 2, pc = within & 3;
        const bf16_t* src = A + (int64_t)min(row, M - 1) * K + k0 + t * 32 + ((pc ^ ((-(row >> 2)) & 3)) << 3);
        const u64 lo = cld64(src), hi = cld64(src + 4);
        *reinterpret_cast<uint4*>(smem + c * 16) = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
    }
    f32x4 acc[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int x = lane & 15, gq = lane >> 4;
    const int a_off = fswz(x, gq);
    __syncthreads();
    for (int s0 = 0; s0 < nsteps; s0 += GU) {
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const bf16x8 w = bf[u];
            if (s0 + GU + u < nsteps) bf[u] = *reinterpret_cast<const bf16x8*>(g.bp + 32 * (s0 + GU + u));
            if (s0 + u < nsteps) {
                const char* tile = smem + (s0 + u) * TILE;
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(tile + a_off + i * 1024);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, w, acc[i], 0, 0, 0);
                }
            }
        }
    }
    if (g.n < N) {
        float* out = P + (int64_t)g.slice * M * N + g.n;
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * i + 4 * gq + r;
                if (m < M) cstf(out + (int64_t)m * N, acc[i][r]);
            }
    }
    __syncthreads();                                   // the LDS image may be overwritten by the next unit / phase
}

// every workgroup's units of one GEMM phase (at most two per workgroup in practice); `bf` holds the first unit's window
__device__ void gemm_phase(const PL& p, bf16x8 (&bf)[GU], GemmUnit g, const bf16_t* W, const bf16_t* A, int N, int K, int ks, float* P, char* smem) {
    const int units = ((N + COLS - 1) / COLS) * ks;
    gemm_run(g, bf, A, p.B, N, K, ks, P, smem);
    for (int u = blockIdx.x + gridDim.x; u < units; u += gridDim.x) {
        const GemmUnit g2 = gemm_issue(bf, W, N, K, ks, u, units);
        gemm_run(g2, bf, A, p.B, N, K, ks, P, smem);
    }
}

// ---- f = act(sum_s P[s] + bias) as bf16 (splitk_finish_kernel of decode.hip)
__device__ __noinline__ void finish_phase(const PL& p, const float* P, int ks, const float* bias, int act, int M, int N, bf16_t* out) {
    const int nq = N >> 2;
    for (int idx = blockIdx.x * NT + threadIdx.x; idx < M * nq; idx += gridDim.x * NT) {
        const int m = idx / nq, n = (idx - m * nq) * 4;
        float4 v = cld4f(P + (int64_t)m * N + n);
        for (int s = 1; s < ks; ++s) {
            const float4 t = cld4f(P + ((int64_t)s * M + m) * N + n);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        v.x = act_fwd(act, v.x); v.y = act_fwd(act, v.y); v.z = act_fwd(act, v.z); v.w = act_fwd(act, v.w);
        cst64(out + (int64_t)m * N + n, pack4bf(v));
    }
}

// ---- attention of one (sample, 4 heads): attn_decode_kernel of attention.hip with 2 waves per head, 10 loads per lane, the V slice
//      by LDS-DMA (K in batches of 80 keys); q and the new K / V row summed from the QKV partial sums, the new row appended to the cache
template <int LPK>
__device__ __noinline__ void attention_unit(const PL& p, const eavqa_lm_layer_t& L, int unit, const float* qkv_part, int ks, char* smem) {
    constexpr int WPH = 2, U = 10, KPI = 64 / LPK, STEP = KPI * WPH * U;
    const int H = p.H, hd = p.hd, Sk = p.Sk, E = H * hd, E3 = 3 * E;
    const int groups = H / 4;
    const int b = unit / groups, hg = unit - b * groups;
    float* dec_sc = reinterpret_cast<float*>(smem);                     // [4][Sk] scores, [4][WPH][128] partial outputs, V image
    char* vimg = reinterpret_cast<char*>(dec_sc + 4 * Sk + 4 * WPH * 128);
    const int cpk = hd >> 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hh = wave / WPH, part = wave % WPH;
    const int h = hg * 4 + hh;
    const int sub = lane / LPK, dl = lane % LPK;
    const bool active = 8 * dl < hd;
    float* sc = dec_sc + hh * Sk;
    float* opart = dec_sc + 4 * Sk + (hh * WPH + part) * 128;
    const int64_t ldk = E;
    const bf16_t* kb = reinterpret_cast<const bf16_t*>(L.k_cache) + (int64_t)b * p.S_max * ldk + h * hd + 8 * dl;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(L.v_cache) + (int64_t)b * p.S_max * ldk + h * hd + 8 * dl;
    {   // the V slice of keys 0 .. Sk-2 (the new row is not in the cache yet)
        const int total = (Sk - 1) * cpk;
        const bf16_t* vsrc = reinterpret_cast<const bf16_t*>(L.v_cache) + (int64_t)b * p.S_max * ldk + hg * 4 * hd;
        for (int base = __builtin_amdgcn_readfirstlane(wave) * 64; base < total; base += NT) {
            const int c = base + lane;
            const int key = c / cpk, piece = c - key * cpk;
            if (c < total)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + (int64_t)key * ldk + piece * 8),
                                                 (__attribute__((address_space(3))) void*)(vimg + base * 16), 16, 0, 0);
        }
    }
    auto key_of = [&](int j0, int u) { return j0 + (u * WPH + part) * KPI + sub; };
    bf16x8 kv[U];                                                       // first batch of K: issued before anything that waits for P2's results
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = key_of(0, u);
        kv[u] = (bf16x8){};
        if (active && j < Sk - 1) kv[u] = *reinterpret_cast<const bf16x8*>(kb + (int64_t)j * ldk);
    }
    // bf16(sum_s P[s][b][col ..] + bias): what the finish pass would have stored
    auto from_part = [&](int col) -> bf16x8 {
        const float* p0 = qkv_part + (int64_t)b * E3 + col;
        const int64_t slice = (int64_t)p.B * E3;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
        for (int s0 = 0; s0 < ks; s0 += 4) {
            float4 ta[4], tc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* ps = p0 + min(s0 + i, ks - 1) * slice;
                ta[i] = cld4f(ps);
                tc[i] = cld4f(ps + 4);
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
        if (L.b_qkv) {
            const float4 a2 = *reinterpret_cast<const float4*>(L.b_qkv + col), c2 = *reinterpret_cast<const float4*>(L.b_qkv + col + 4);
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w; c.x += c2.x; c.y += c2.y; c.z += c2.z; c.w += c2.w;
        }
        bf16x8 r;
        r[0] = (bf16_t)a.x; r[1] = (bf16_t)a.y; r[2] = (bf16_t)a.z; r[3] = (bf16_t)a.w;
        r[4] = (bf16_t)c.x; r[5] = (bf16_t)c.y; r[6] = (bf16_t)c.z; r[7] = (bf16_t)c.w;
        return r;
    };
    bf16x8 knew = {}, vnew = {};
    const int rem = (Sk - 1) % STEP, grp = rem / KPI;
    const bool own_new = active && (rem % KPI) == sub && (grp % WPH) == part;
    if (own_new) {
        knew = from_part(E + h * hd + 8 * dl);
        vnew = from_part(2 * E + h * hd + 8 * dl);
        *reinterpret_cast<bf16x8*>(const_cast<bf16_t*>(kb) + (int64_t)(Sk - 1) * ldk) = knew;
        *reinterpret_cast<bf16x8*>(const_cast<bf16_t*>(vb) + (int64_t)(Sk - 1) * ldk) = vnew;
        *reinterpret_cast<bf16x8*>(vimg + ((Sk - 1) * cpk + hh * (hd >> 3) + dl) * 16) = vnew;
    }
    float qf[8];
    {
        bf16x8 t = {};
        if (active) t = from_part(h * hd + 8 * dl);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[e] = (float)t[e];
    }
    const int32_t* mrow = p.key_mask ? p.key_mask + (int64_t)b * p.ld_mask : nullptr;
    for (int j0 = 0; j0 < Sk; j0 += STEP) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = key_of(j0, u);
            if (j0 > 0) {
                kv[u] = (bf16x8){};
                if (active && j < Sk - 1) kv[u] = *reinterpret_cast<const bf16x8*>(kb + (int64_t)j * ldk);
            }
            if (own_new && j == Sk - 1) kv[u] = knew;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = key_of(j0, u);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d += qf[e] * (float)kv[u][e];
#pragma unroll
            for (int o = LPK >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
            if (dl == 0 && j < Sk) sc[j] = (mrow && mrow[j] == 0) ? -FLT_MAX : d * p.scale;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00);          // vmcnt(0): this wave's share of the V image has landed
    __syncthreads();
    float mx = -FLT_MAX;
    for (int j = lane; j < Sk; j += 64) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < Sk; j += 64) sum += __expf(sc[j] - mx);
    sum = wave_sum(sum);
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < Sk; j0 += STEP) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = key_of(j0, u);
            if (active && j < Sk) {
                const bf16x8 vv = *reinterpret_cast<const bf16x8*>(vimg + (j * cpk + hh * (hd >> 3) + dl) * 16);
                const float pj = __expf(sc[j] - mx);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += pj * (float)vv[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int off = LPK; off < 64; off <<= 1) o[e] += __shfl_xor(o[e], off, 64);
    __syncthreads();
    if (sub == 0 && active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) opart[8 * dl + e] = o[e];
    }
    __syncthreads();
    if (part == 0 && sub == 0 && active) {
        const float* p0 = dec_sc + 4 * Sk + hh * WPH * 128 + 8 * dl;
        const float inv = 1.f / sum;
        float4 lo, hi;
        lo.x = (p0[0] + p0[128 + 0]) * inv; lo.y = (p0[1] + p0[128 + 1]) * inv; lo.z = (p0[2] + p0[128 + 2]) * inv; lo.w = (p0[3] + p0[128 + 3]) * inv;
        hi.x = (p0[4] + p0[128 + 4]) * inv; hi.y = (p0[5] + p0[128 + 5]) * inv; hi.z = (p0[6] + p0[128 + 6]) * inv; hi.w = (p0[7] + p0[128 + 7]) * inv;
        bf16_t* dst = p.ctx + (int64_t)b * E + h * hd + 8 * dl;
        cst64(dst, pack4bf(lo));
        cst64(dst + 4, pack4bf(hi));
    }
    __syncthreads();                                       // LDS free for the next phase
}

__global__ __launch_bounds__(NT) void lm_decode_persistent_kernel(PL p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[NWAVE];
    __shared__ int flag;
    unsigned epoch = 0;
    const int bid = blockIdx.x;
    const int E = p.E, F = p.F, B = p.B;
    bf16x8 bf[GU];
    const int u_qkv = ((3 * E + COLS - 1) / COLS) * p.ks_qkv, u_o = ((E + COLS - 1) / COLS) * p.ks_o;
    const int u_fc1 = ((F + COLS - 1) / COLS) * p.ks_fc1, u_fc2 = ((E + COLS - 1) / COLS) * p.ks_fc2;
    for (int l = 0; l < p.n_layer; ++l) {
        const eavqa_lm_layer_t& L = p.layers[l];
        // P1: x = x1 + b_fc2(prev) + sum FFN-down partials(prev); a = LN1(x)      (layer 0: x is the input)
        if (bid < B) {
            if (l == 0) ln_row(bid, B, E, p.x, nullptr, 0, nullptr, nullptr, L.ln1_g, L.ln1_b, p.eps, p.a, red);
            else ln_row(bid, B, E, p.x1, p.part2, p.ks_fc2, p.layers[l - 1].b_fc2, p.x, L.ln1_g, L.ln1_b, p.eps, p.a, red);
        }
        GemmUnit g = gemm_issue(bf, reinterpret_cast<const bf16_t*>(L.w_qkv), 3 * E, E, p.ks_qkv, bid, u_qkv);
        if (!grid_barrier(p, epoch, &flag)) return;
        // P2: QKV projection -> partial sums
        gemm_phase(p, bf, g, reinterpret_cast<const bf16_t*>(L.w_qkv), p.a, 3 * E, E, p.ks_qkv, p.part, smem);
        if (!grid_barrier(p, epoch, &flag)) return;
        // P3: attention (+ K / V append)
        {
            const int units = B * (p.H / 4);
            for (int u = bid; u < units; u += gridDim.x) {
                if (p.hd <= 64) attention_unit<8>(p, L, u, p.part, p.ks_qkv, smem);
                else attention_unit<16>(p, L, u, p.part, p.ks_qkv, smem);
            }
        }
        g = gemm_issue(bf, reinterpret_cast<const bf16_t*>(L.w_o), E, E, p.ks_o, bid, u_o);
        if (!grid_barrier(p, epoch, &flag)) return;
        // P4: out-projection -> partial sums
        gemm_phase(p, bf, g, reinterpret_cast<const bf16_t*>(L.w_o), p.ctx, E, E, p.ks_o, p.part, smem);
        if (!grid_barrier(p, epoch, &flag)) return;
        // P5: x1 = x + b_o + sum partials; a = LN2(x1)
        if (bid < B) ln_row(bid, B, E, p.x, p.part, p.ks_o, L.b_o, p.x1, L.ln2_g, L.ln2_b, p.eps, p.a, red);
        g = gemm_issue(bf, reinterpret_cast<const bf16_t*>(L.w_fc1), F, E, p.ks_fc1, bid, u_fc1);
        if (!grid_barrier(p, epoch, &flag)) return;
        // P6: FFN-up -> partial sums
        gemm_phase(p, bf, g, reinterpret_cast<const bf16_t*>(L.w_fc1), p.a, F, E, p.ks_fc1, p.part, smem);
        if (!grid_barrier(p, epoch, &flag)) return;
        // P7: f = act(sum partials + b_fc1)
        finish_phase(p, p.part, p.ks_fc1, L.b_fc1, p.act, B, F, p.f);
        g = gemm_issue(bf, reinterpret_cast<const bf16_t*>(L.w_fc2), E, F, p.ks_fc2, bid, u_fc2);
        if (!grid_barrier(p, epoch, &flag)) return;
        // P8: FFN-down -> partial sums (summed by the next layer's LayerNorm pass / the final pass)
        gemm_phase(p, bf, g, reinterpret_cast<const bf16_t*>(L.w_fc2), p.f, E, F, p.ks_fc2, p.part2, smem);
        if (!grid_barrier(p, epoch, &flag)) return;
    }
    // x = x1 + b_fc2 + sum(last FFN-down partials)
    {
        const float* bias = p.layers[p.n_layer - 1].b_fc2;
        const int nq = E >> 2;
        for (int idx = bid * NT + threadIdx.x; idx < B * nq; idx += gridDim.x * NT) {
            const int m = idx / nq, n = (idx - m * nq) * 4;
            float4 v = cld4f(p.part2 + (int64_t)m * E + n);
            for (int s = 1; s < p.ks_fc2; ++s) {
                const float4 t = cld4f(p.part2 + ((int64_t)s * B + m) * E + n);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            const float4 r = cld4f(p.x1 + (int64_t)m * E + n);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            *reinterpret_cast<float4*>(p.x + (int64_t)m * E + n) = v;
        }
    }
}

struct Persist {
    void* d_layers = nullptr; size_t cap = 0; unsigned* d_sync = nullptr; int max_lds = 0;
};
Persist g_persist;
}  // namespace

// Returns EAVQA_OK when the step was enqueued, EAVQA_E_SHAPE when this shape is not covered (the caller takes the multi-kernel route).
int eavqa_detail_lm_decode_persistent(int dtype, int n_layer, const eavqa_lm_layer_t* layers, int E, int H, int F, int act, float eps, int B,
                                          int row0, int S_max, float* x, const int32_t* key_mask, int64_t ld_mask, void* a, void* ctx, float* x1,
                                          void* f, float* part, float* part2, int ks_qkv, int ks_o, int ks_fc1, int ks_fc2, void* stream) {
    if (dtype != EAVQA_BF16 || !layers || !x || n_layer <= 0) return EAVQA_E_SHAPE;
    const int hd = E / H, Sk = row0 + 1;
    if (B <= 0 || B > 32 || E % H || H % 4 || hd % 8 || hd > 128 || E % 128 || F % 128 || E > 4096 || Sk < 1 || Sk > S_max) return EAVQA_E_SHAPE;
    if (ks_qkv <= 0 || ks_o <= 0 || ks_fc1 <= 0 || ks_fc2 <= 0) return EAVQA_E_SHAPE;
    auto slice_lds = [](int K, int ks) { return (size_t)(K / ks) * 2 * 32; };
    size_t lds = slice_lds(E, ks_qkv);
    lds = lds > slice_lds(E, ks_o) ? lds : slice_lds(E, ks_o);
    lds = lds > slice_lds(E, ks_fc1) ? lds : slice_lds(E, ks_fc1);
    lds = lds > slice_lds(F, ks_fc2) ? lds : slice_lds(F, ks_fc2);
    const size_t attn_lds = ((size_t)4 * Sk + 4 * 2 * 128) * 4 + (size_t)Sk * 4 * hd * 2;
    lds = lds > attn_lds ? lds : attn_lds;
    if (lds > 150 * 1024) return EAVQA_E_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Persist& ps = g_persist;
    const size_t need = (size_t)n_layer * sizeof(eavqa_lm_layer_t);
    if (ps.cap < need) {
        if (ps.d_layers) (void)hipFree(ps.d_layers);
        if (hipMalloc(&ps.d_layers, need) != hipSuccess) { ps.d_layers = nullptr; ps.cap = 0; return EAVQA_E_LAUNCH; }
        ps.cap = need;
    }
    if (!ps.d_sync && hipMalloc(reinterpret_cast<void**>(&ps.d_sync), 8) != hipSuccess) return EAVQA_E_LAUNCH;
    if (ps.max_lds == 0) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lm_decode_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return EAVQA_E_LAUNCH;
        ps.max_lds = 150 * 1024;
    }
    if (hipMemcpyAsync(ps.d_layers, layers, need, hipMemcpyHostToDevice, s) != hipSuccess) return EAVQA_E_LAUNCH;
    if (hipMemsetAsync(ps.d_sync, 0, 8, s) != hipSuccess) return EAVQA_E_LAUNCH;
    PL p = {};
    p.layers = static_cast<const eavqa_lm_layer_t*>(ps.d_layers);
    p.n_layer = n_layer; p.E = E; p.H = H; p.F = F; p.act = act; p.B = B; p.row0 = row0; p.S_max = S_max; p.Sk = Sk; p.hd = hd;
    p.eps = eps; p.scale = 1.0f / sqrtf((float)hd);
    p.x = x; p.key_mask = key_mask; p.ld_mask = ld_mask;
    p.x1 = x1; p.a = static_cast<bf16_t*>(a); p.ctx = static_cast<bf16_t*>(ctx); p.f = static_cast<bf16_t*>(f); p.part = part; p.part2 = part2;
    p.sync = ps.d_sync;
    p.ks_qkv = ks_qkv; p.ks_o = ks_o; p.ks_fc1 = ks_fc1; p.ks_fc2 = ks_fc2;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return EAVQA_E_LAUNCH;
    void* args[] = {&p};
    if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(lm_decode_persistent_kernel), dim3(cus), dim3(NT), args, (unsigned)lds, s) != hipSuccess) {
        (void)hipGetLastError();
        return EAVQA_E_LAUNCH;
    }
    return EAVQA_OK;
}
