// RICES retrieval on the GPU: brute-force inner-product k-nearest-neighbours over
// L2-normalised CLIP text embeddings, the arithmetic of src/in_context_example_selection/get_question_knn.py:64-76
// (faiss.normalize_L2 + IndexFlatIP.search, k = 2048 over ~443 k x 768).  The scores are one exact-fp32 eavqa_gemm
// per query tile; this file holds the two HBM-bound pieces around it:
//
//   eavqa_l2_normalize_rows   x[r, :] /= ||x[r, :]||_2   (rows of norm 0 are left alone, as faiss does)
//   eavqa_topk_rows           per row: the k largest scores, sorted descending, ties broken by the smaller column
//
// Top-k: one 1024-thread workgroup per row.  (1) radix select on the order-preserving integer image of the floats,
// four 8-bit passes with an LDS histogram each, finds the k-th largest key T; (2) every wave owns a contiguous segment
// of the row, counts its elements > T and == T, and after an exclusive scan over the 16 waves writes them at positions
// that depend only on the data (ballot + popcount inside the wave): elements equal to T are kept in column order until
// k is reached; (3) the k (key, column) pairs are bitonic-sorted in LDS.  The row is read six times (it stays in L2
// for rows up to a few MB); nothing depends on scheduling, so the result is bitwise reproducible.
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t key_of(float v) {
    uint32_t u = __float_as_uint(v);
    if (u == 0x80000000u) u = 0u;                           // -0 compares equal to +0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // ascending in the float order (NaNs at the ends)
}
__device__ __forceinline__ float value_of(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void l2_normalize_kernel(int rows, int cols, float* x, int64_t ld) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    float* p = x + (int64_t)row * ld;
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) { const float v = p[c]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float nr = (red[0] + red[1]) + (red[2] + red[3]);
    if (nr > 0.f) {
        const float inv = 1.f / sqrtf(nr);
        for (int c = threadIdx.x; c < cols; c += 256) p[c] *= inv;
    }
}

constexpr int TK_THREADS = 1024, TK_WAVES = 16, TK_MAX = 2048;

__global__ __launch_bounds__(TK_THREADS) void topk_rows_kernel(int cols, const float* __restrict__ scores, int64_t ld, int k,
                                                               float* __restrict__ out_val, int64_t* __restrict__ out_idx) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel[2];                       // [0] prefix found so far, [1] how many of the k remain below it
    __shared__ uint32_t wave_gt[TK_WAVES], wave_eq[TK_WAVES];
    __shared__ unsigned long long cand[TK_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = scores + (int64_t)blockIdx.x * ld;

    // ---- (1) k-th largest key by radix select, most significant byte first
    uint32_t prefix = 0, mask = 0, want = (uint32_t)k;   // `want`: rank (1-based, from the top) inside the current bucket
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int c = tid; c < cols; c += TK_THREADS) {
            const uint32_t key = key_of(x[c]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t acc = 0;
            int b = 255;
            for (; b > 0; --b) {
                if (acc + hist[b] >= want) break;
                acc += hist[b];
            }
            sel[0] = prefix | ((uint32_t)b << shift);
            sel[1] = want - acc;
        }
        __syncthreads();
        prefix = sel[0];
        want = sel[1];
        mask |= 255u << shift;
        __syncthreads();
    }
    const uint32_t T = prefix;                        // exactly `want` of the elements equal to T belong to the top k
    const uint32_t need_eq = want;

    // ---- (2) deterministic compaction: wave w owns columns [w * seg, (w + 1) * seg)
    const int seg = ((cols + TK_WAVES - 1) / TK_WAVES + 63) & ~63;
    const int c_begin = wave * seg, c_end = min(cols, c_begin + seg);
    uint32_t n_gt = 0, n_eq = 0;
    for (int c = c_begin + lane; c < c_begin + seg; c += 64) {
        const uint32_t key = c < c_end ? key_of(x[c]) : 0u;
        n_gt += __popcll(__ballot(c < c_end && key > T));
        n_eq += __popcll(__ballot(c < c_end && key == T));
    }
    if (lane == 0) { wave_gt[wave] = n_gt; wave_eq[wave] = n_eq; }
    __syncthreads();
    uint32_t base_gt = 0, base_eq = 0, total_gt = 0;
    for (int w = 0; w < TK_WAVES; ++w) {
        if (w < wave) { base_gt += wave_gt[w]; base_eq += wave_eq[w]; }
        total_gt += wave_gt[w];
    }
    // layout of cand: [0, total_gt) the elements above T, then the first need_eq elements equal to T in column order
    for (int c = c_begin + lane; c < c_begin + seg; c += 64) {
        const bool in = c < c_end;
        const uint32_t key = in ? key_of(x[c]) : 0u;
        const bool gt = in && key > T, eq = in && key == T;
        const unsigned long long bg = __ballot(gt), be = __ballot(eq);
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned long long entry = ((unsigned long long)key << 32) | (uint32_t)(0xffffffffu - (uint32_t)c);
        if (gt) cand[base_gt + __popcll(bg & below)] = entry;
        if (eq) {
            const uint32_t pos = base_eq + __popcll(be & below);
            if (pos < need_eq) cand[total_gt + pos] = entry;
        }
        base_gt += __popcll(bg);
        base_eq += __popcll(be);
    }
    // pad to the next power of two with entries that sort last
    int n2 = 1;
    while (n2 < k) n2 <<= 1;
    for (int i = k + tid; i < n2; i += TK_THREADS) cand[i] = 0ull;
    __syncthreads();

    // ---- (3) bitonic sort, descending: larger key first, then smaller column (stored as 0xffffffff - column)
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (n2 >> 1); i += TK_THREADS) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = cand[lo], b = cand[hi];
                if ((a < b) == desc) { cand[lo] = b; cand[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += TK_THREADS) {
        const unsigned long long e = cand[i];
        out_val[(int64_t)blockIdx.x * k + i] = value_of((uint32_t)(e >> 32));
        out_idx[(int64_t)blockIdx.x * k + i] = (int64_t)(0xffffffffu - (uint32_t)e);
    }
}

}  // namespace

extern "C" int eavqa_l2_normalize_rows(int rows, int cols, float* x, int64_t ld, void* stream) {
    if (!x || rows <= 0 || cols <= 0 || ld < cols) return EAVQA_E_ARG;
    hipLaunchKernelGGL(l2_normalize_kernel, dim3(rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rows, cols, x, ld);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_topk_rows(int rows, int cols, const float* scores, int64_t ld, int k, float* out_val, int64_t* out_idx,
                               void* stream) {
    if (!scores || !out_val || !out_idx || rows <= 0 || cols <= 0 || ld < cols) return EAVQA_E_ARG;
    if (k <= 0 || k > TK_MAX || k > cols) return EAVQA_E_SHAPE;
    hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(TK_THREADS), 0, reinterpret_cast<hipStream_t>(stream), cols, scores, ld, k,
                       out_val, out_idx);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}
