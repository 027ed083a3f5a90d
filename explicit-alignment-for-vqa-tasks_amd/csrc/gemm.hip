// MFMA GEMM with fused epilogue for gfx950 (eavqa_gemm in include/eavqa.h).
//
//   C[M,N] = epilogue(alpha * sum_k A(m,k) * B(n,k))
//
// Two kernels share one tile shape (128x128 output per 256-thread workgroup, 2x2 waves,
// 64x64 per wave) and one LDS-staged epilogue:
//   * bf16: v_mfma_f32_16x16x32_bf16, BK = 64, operands staged k-contiguous in LDS as
//     [128 rows][64 k] with a 16-byte-chunk XOR swizzle (chunk ^= row & 7) so that the
//     ds_read_b128 fragment reads are bank-conflict free;
//   * f32 : v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain - the parity path), BK = 16,
//     operands staged as [128 rows][16 k] with a 17-float row pitch.
// Global->LDS staging goes through registers: the loads of K-tile t+1 are issued before
// the MFMAs of tile t and written to the other LDS buffer after them (one barrier per tile).
// An operand whose contiguous memory dimension is NOT k (Conv1D-style [K,N] weights, the
// transposed operands of the mapper's dgrad/wgrad) is transposed while it is written to LDS.
// The accumulators leave through LDS so that bias / residual / aux traffic and the C stores
// are 16-byte row-contiguous accesses.
#include "common.h"

namespace {

struct GemmParams {
    const void* A; const void* B; void* C;
    const float* bias; const void* aux_in; void* aux_out; const float* residual;
    int M, N, K;
    int64_t lda, ldb, ldc, ld_aux, ldr;
    int act, out_f32;
    float alpha;
    int tiles_m, tiles_n;
    int vec_c, vec_aux, vec_res;   // 16-byte (8-byte for bf16) vector access allowed on C / aux / residual
};

constexpr int BM = 128, BN = 128;
constexpr int CS_PITCH = 132;                       // floats per row of the staged C tile
constexpr int CS_BYTES = BM * CS_PITCH * 4;         // 67,584 B

// XCD-aware, bijective block -> tile map: the dispatcher deals blocks round-robin over the
// 8 XCDs, so give each XCD a contiguous run of tiles (M fastest) to share operand panels in L2.
__device__ __forceinline__ void tile_coords(const GemmParams& p, int& tm, int& tn) {
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tm = wgid % p.tiles_m;
    tn = wgid / p.tiles_m;
}

// ---- epilogue shared by both kernels: Cs holds the 128x128 fp32 tile (pitch CS_PITCH) ----
template <typename T>
__device__ __forceinline__ void epilogue(const GemmParams& p, const float* Cs, int m0, int n0) {
    const int tid = threadIdx.x;
    const int c4 = (tid & 31) * 4;
    const int n = n0 + c4;
    const bool vec_c = p.vec_c, vec_aux = p.vec_aux, vec_res = p.vec_res;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n + j < p.N) bias4[j] = p.bias[n + j];
    }
    const T* aux_in = reinterpret_cast<const T*>(p.aux_in);
    T* aux_out = reinterpret_cast<T*>(p.aux_out);
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
        const int row = (tid >> 5) + pass * 8;
        const int m = m0 + row;
        if (m >= p.M || n >= p.N) continue;
        const float4 a = *reinterpret_cast<const float4*>(&Cs[row * CS_PITCH + c4]);
        float v[4] = {a.x, a.y, a.z, a.w};
        const bool full = (n + 3 < p.N);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = p.alpha * v[j] + bias4[j];
        if (aux_out) {
            T* q = aux_out + (int64_t)m * p.ld_aux + n;
            if (full && vec_aux) elem<T>::st4(q, make_float4(v[0], v[1], v[2], v[3]));
            else
                for (int j = 0; j < 4; ++j)
                    if (n + j < p.N) elem<T>::st(q + j, v[j]);
        }
        if (aux_in) {
            const T* q = aux_in + (int64_t)m * p.ld_aux + n;
            float u[4] = {0.f, 0.f, 0.f, 0.f};
            if (full && vec_aux) { float4 t = elem<T>::ld4(q); u[0] = t.x; u[1] = t.y; u[2] = t.z; u[3] = t.w; }
            else
                for (int j = 0; j < 4; ++j)
                    if (n + j < p.N) u[j] = elem<T>::ld(q + j);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= act_bwd(p.act, u[j]);
        } else if (p.act != EAVQA_ACT_NONE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act_fwd(p.act, v[j]);
        }
        if (p.residual) {
            const float* q = p.residual + (int64_t)m * p.ldr + n;
            if (full && vec_res) { float4 t = *reinterpret_cast<const float4*>(q); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
            else
                for (int j = 0; j < 4; ++j)
                    if (n + j < p.N) v[j] += q[j];
        }
        if (p.out_f32) {
            float* q = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
            if (full && vec_c) *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
            else
                for (int j = 0; j < 4; ++j)
                    if (n + j < p.N) q[j] = v[j];
        } else {
            T* q = reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + n;
            if (full && vec_c) elem<T>::st4(q, make_float4(v[0], v[1], v[2], v[3]));
            else
                for (int j = 0; j < 4; ++j)
                    if (n + j < p.N) elem<T>::st(q + j, v[j]);
        }
    }
}

// =============================================================== bf16 ===
constexpr int BK16 = 64;                          // k per LDS tile (bf16)
constexpr int OPER16_BYTES = 128 * BK16 * 2;      // 16 KiB per operand per buffer

// byte offset of the 16-byte chunk (row, kc) inside a swizzled [128][64] bf16 tile
__device__ __forceinline__ int swz16(int row, int kc) { return row * 128 + ((kc ^ (row & 7)) << 4); }

// Stage one operand tile (128 rows x 64 k) from global memory into registers.
// KC: memory is [rows][K] (k contiguous); else memory is [K][rows] (row contiguous).
template <bool KC>
__device__ __forceinline__ void g2r_16(uint4 (&r)[4], const bf16_t* X, int64_t ld, int row0, int rows_max,
                                       int k0, int K) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (KC) {
            const int row = c >> 3, kc = c & 7;
            const int gr = row0 + row, gk = k0 + kc * 8;
            if (gr < rows_max && gk < K) v = *reinterpret_cast<const uint4*>(X + (int64_t)gr * ld + gk);
        } else {
            const int k = c >> 4, rc = c & 15;
            const int gk = k0 + k, gr = row0 + rc * 8;
            if (gk < K && gr < rows_max) v = *reinterpret_cast<const uint4*>(X + (int64_t)gk * ld + gr);
        }
        r[i] = v;
    }
}
template <bool KC>
__device__ __forceinline__ void r2s_16(const uint4 (&r)[4], char* S) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        if (KC) {
            const int row = c >> 3, kc = c & 7;
            *reinterpret_cast<uint4*>(S + swz16(row, kc)) = r[i];
        } else {
            const int k = c >> 4, rc = c & 15;
            const unsigned w[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = rc * 8 + j;
                const unsigned short e = (unsigned short)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu));
                *reinterpret_cast<unsigned short*>(S + swz16(row, k >> 3) + (k & 7) * 2) = e;
            }
        }
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS carve: A buffers at 0 / 16 KiB, B buffers at 32 / 48 KiB (pointer arrays of LDS addresses
    // would become static initialisers, which the backend rejects - use offsets)
    char* const As0 = smem;
    char* const Bs0 = smem + 2 * OPER16_BYTES;

    int tm, tn;
    tile_coords(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK16 - 1) / BK16;
    uint4 ra[4], rb[4];
    g2r_16<A_KC>(ra, A, p.lda, m0, p.M, 0, p.K);
    g2r_16<B_KC>(rb, B, p.ldb, n0, p.N, 0, p.K);
    r2s_16<A_KC>(ra, As0);
    r2s_16<B_KC>(rb, Bs0);
    __syncthreads();

    const int frow = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            g2r_16<A_KC>(ra, A, p.lda, m0, p.M, (kt + 1) * BK16, p.K);
            g2r_16<B_KC>(rb, B, p.ldb, n0, p.N, (kt + 1) * BK16, p.K);
        }
        const char* Ac = As0 + cur * OPER16_BYTES;
        const char* Bc = Bs0 + cur * OPER16_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + frow;
                af[i] = *reinterpret_cast<const bf16x8*>(Ac + swz16(row, s * 4 + fk));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + j * 16 + frow;
                bfr[j] = *reinterpret_cast<const bf16x8*>(Bc + swz16(row, s * 4 + fk));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            r2s_16<A_KC>(ra, As0 + (cur ^ 1) * OPER16_BYTES);
            r2s_16<B_KC>(rb, Bs0 + (cur ^ 1) * OPER16_BYTES);
        }
        __syncthreads();
    }

    // accumulators -> LDS (C/D map of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg)
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                const int col = wn * 64 + j * 16 + (lane & 15);
                Cs[row * CS_PITCH + col] = acc[i][j][r];
            }
    __syncthreads();
    epilogue<bf16_t>(p, Cs, m0, n0);
}

// ================================================================ f32 ===
constexpr int BK32 = 16;
constexpr int PITCH32 = 17;                                  // floats per staged row
constexpr int OPER32_FLOATS = 128 * PITCH32;                 // 2176 floats = 8704 B

template <bool KC>
__device__ __forceinline__ void g2r_32(float4 (&r)[2], const float* X, int64_t ld, int row0, int rows_max,
                                       int k0, int K) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KC) {
            const int row = c >> 2, kc = c & 3;
            const int gr = row0 + row, gk = k0 + kc * 4;
            if (gr < rows_max && gk < K) v = *reinterpret_cast<const float4*>(X + (int64_t)gr * ld + gk);
        } else {
            const int k = c >> 5, rc = c & 31;
            const int gk = k0 + k, gr = row0 + rc * 4;
            if (gk < K && gr < rows_max) v = *reinterpret_cast<const float4*>(X + (int64_t)gk * ld + gr);
        }
        r[i] = v;
    }
}
template <bool KC>
__device__ __forceinline__ void r2s_32(const float4 (&r)[2], float* S) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;
        const float w[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
        if (KC) {
            const int row = c >> 2, kc = c & 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) S[row * PITCH32 + kc * 4 + j] = w[j];
        } else {
            const int k = c >> 5, rc = c & 31;
#pragma unroll
            for (int j = 0; j < 4; ++j) S[(rc * 4 + j) * PITCH32 + k] = w[j];
        }
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sm = reinterpret_cast<float*>(smem);
    float* const As0 = sm;
    float* const Bs0 = sm + 2 * OPER32_FLOATS;

    int tm, tn;
    tile_coords(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const float* A = reinterpret_cast<const float*>(p.A);
    const float* B = reinterpret_cast<const float*>(p.B);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK32 - 1) / BK32;
    float4 ra[2], rb[2];
    g2r_32<A_KC>(ra, A, p.lda, m0, p.M, 0, p.K);
    g2r_32<B_KC>(rb, B, p.ldb, n0, p.N, 0, p.K);
    r2s_32<A_KC>(ra, As0);
    r2s_32<B_KC>(rb, Bs0);
    __syncthreads();

    const int frow = lane & 31, fk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            g2r_32<A_KC>(ra, A, p.lda, m0, p.M, (kt + 1) * BK32, p.K);
            g2r_32<B_KC>(rb, B, p.ldb, n0, p.N, (kt + 1) * BK32, p.K);
        }
        const float* Ac = As0 + cur * OPER32_FLOATS;
        const float* Bc = Bs0 + cur * OPER32_FLOATS;
#pragma unroll
        for (int s = 0; s < BK32 / 2; ++s) {
            float af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = Ac[(wm * 64 + i * 32 + frow) * PITCH32 + s * 2 + fk];
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = Bc[(wn * 64 + j * 32 + frow) * PITCH32 + s * 2 + fk];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            r2s_32<A_KC>(ra, As0 + (cur ^ 1) * OPER32_FLOATS);
            r2s_32<B_KC>(rb, Bs0 + (cur ^ 1) * OPER32_FLOATS);
        }
        __syncthreads();
    }

    // C/D map of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = wn * 64 + j * 32 + (lane & 31);
                Cs[row * CS_PITCH + col] = acc[i][j][r];
            }
    __syncthreads();
    epilogue<float>(p, Cs, m0, n0);
}

typedef void (*gemm_kernel_t)(GemmParams);

int launch(gemm_kernel_t kernel, const GemmParams& p, hipStream_t stream) {
    // dynamic LDS above 64 KiB must be opted into once per kernel; remember which ones were
    // (idempotent, so a race between host threads only repeats the call)
    static gemm_kernel_t configured[8] = {nullptr};
    bool done = false;
    for (int i = 0; i < 8; ++i) done |= (configured[i] == kernel);
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                CS_BYTES) != hipSuccess)
            return EAVQA_E_LAUNCH;
        for (int i = 0; i < 8; ++i)
            if (configured[i] == nullptr) { configured[i] = kernel; break; }
    }
    const int nwg = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL(kernel, dim3(nwg), dim3(256), CS_BYTES, stream, p);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

}  // namespace

extern "C" int eavqa_gemm(int dtype, int a_kc, int b_kc, int M, int N, int K,
                          const void* A, int64_t lda, const void* B, int64_t ldb,
                          void* C, int64_t ldc, int out_f32, float alpha,
                          const float* bias, int act,
                          const void* aux_in, void* aux_out, int64_t ld_aux,
                          const float* residual, int64_t ldr, void* stream) {
    if (!A || !B || !C) return EAVQA_E_ARG;
    if (M <= 0 || N <= 0 || K <= 0) return EAVQA_E_ARG;
    if (dtype != EAVQA_F32 && dtype != EAVQA_BF16) return EAVQA_E_DTYPE;
    if (act < EAVQA_ACT_NONE || act > EAVQA_ACT_QUICK_GELU) return EAVQA_E_DTYPE;
    const int vec = dtype == EAVQA_BF16 ? 8 : 4;
    // the contiguous memory dimension of each operand is read in 16-byte chunks
    const int a_contig = a_kc ? K : M, b_contig = b_kc ? K : N;
    if (a_contig % vec || b_contig % vec) return EAVQA_E_SHAPE;
    if (lda % vec || ldb % vec) return EAVQA_E_ALIGN;
    if (!eavqa_aligned16(A) || !eavqa_aligned16(B)) return EAVQA_E_ALIGN;
    if (lda < a_contig || ldb < b_contig || ldc < N) return EAVQA_E_ARG;
    if ((aux_in || aux_out) && ld_aux < N) return EAVQA_E_ARG;
    if (residual && ldr < N) return EAVQA_E_ARG;

    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.aux_in = aux_in; p.aux_out = aux_out; p.residual = residual;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ld_aux = ld_aux; p.ldr = ldr;
    p.act = act; p.out_f32 = out_f32; p.alpha = alpha;
    p.tiles_m = (M + BM - 1) / BM;
    p.tiles_n = (N + BN - 1) / BN;
    const int esz = dtype == EAVQA_BF16 ? 2 : 4;
    auto vec_ok = [](const void* ptr, int64_t ld, int bytes_per_elem) {
        return ((reinterpret_cast<uintptr_t>(ptr) % (4 * bytes_per_elem)) == 0) && (ld % 4 == 0);
    };
    p.vec_c = vec_ok(C, ldc, out_f32 ? 4 : esz);
    p.vec_aux = vec_ok(aux_in ? aux_in : aux_out, ld_aux, esz);
    p.vec_res = vec_ok(residual, ldr, 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_BF16) {
        if (a_kc && b_kc) return launch(gemm_bf16_kernel<true, true>, p, s);
        if (a_kc && !b_kc) return launch(gemm_bf16_kernel<true, false>, p, s);
        if (!a_kc && b_kc) return launch(gemm_bf16_kernel<false, true>, p, s);
        return launch(gemm_bf16_kernel<false, false>, p, s);
    }
    if (a_kc && b_kc) return launch(gemm_f32_kernel<true, true>, p, s);
    if (a_kc && !b_kc) return launch(gemm_f32_kernel<true, false>, p, s);
    if (!a_kc && b_kc) return launch(gemm_f32_kernel<false, true>, p, s);
    return launch(gemm_f32_kernel<false, false>, p, s);
}
