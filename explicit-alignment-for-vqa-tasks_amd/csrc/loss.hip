// Shifted cross-entropy over float32 logits and the greedy pick (eavqa_ce_fwd / eavqa_ce_bwd /
// eavqa_greedy_pick in include/eavqa.h).  HBM-bound: a kept row is read once in the forward
// (online max / sum-exp, one 256-thread block per row) and once in the backward; ignored rows
// (label -100 after the shift) are never read.  Reductions are fixed-order trees: results are
// bitwise reproducible run to run.
#include "common.h"

namespace {

// S > 0: `labels` is the unshifted [B,S] matrix and row r = b*S+s is scored against labels[b, s+1];
// S == 0: `labels` already holds one (shifted) label per row (packed rows, eavqa_build_row_plan).
__device__ __forceinline__ int64_t shifted_label(const int64_t* labels, int S, int row) {
    if (S == 0) return labels[row];
    const int b = row / S, s = row - b * S;
    return (s + 1 < S) ? labels[(int64_t)b * S + s + 1] : -100;
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(int S, int V, const float* logits, int64_t ld, const int64_t* labels,
                                                     float* row_loss, float* row_lse) {
    __shared__ float sm[8], ss[8];
    const int row = blockIdx.x;
    const int64_t lab = shifted_label(labels, S, row);
    if (lab < 0 || lab >= V) {  // ignore_index (-100); out-of-range labels are treated as ignored
        if (threadIdx.x == 0) { row_loss[row] = 0.f; row_lse[row] = 0.f; }
        return;
    }
    const float* x = logits + (int64_t)row * ld;
    float m = -INFINITY, s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) {
        const float v = x[c];
        if (v > m) { s = s * expf(m - v) + 1.f; m = v; }
        else s += expf(v - m);
    }
    // wave then block combine of (m, s) pairs
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        const float mn = fmaxf(m, m2);
        s = (m == -INFINITY ? 0.f : s * expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * expf(m2 - mn));
        m = mn;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sm[wave] = m; ss[wave] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = sm[0], Ssum = ss[0];
        for (int w = 1; w < 4; ++w) {
            const float mn = fmaxf(M, sm[w]);
            Ssum = (M == -INFINITY ? 0.f : Ssum * expf(M - mn)) + (sm[w] == -INFINITY ? 0.f : ss[w] * expf(sm[w] - mn));
            M = mn;
        }
        const float lse = M + logf(Ssum);
        row_lse[row] = lse;
        row_loss[row] = lse - x[lab];
    }
}

// V <= 64 Ki: one 1024-thread workgroup per row keeps the whole row in registers (<= 64 values per thread): the row is
// read from memory once, every load is in flight before the first use, one exp per element; max and sum are fixed-order
// trees (wave shuffles, then the 16 wave results in order).
constexpr int CE_T = 1024, CE_NPT = 64;
__global__ __launch_bounds__(CE_T) void ce_fwd_row_kernel(int S, int V, const float* __restrict__ logits, int64_t ld,
                                                          const int64_t* __restrict__ labels, float* __restrict__ row_loss,
                                                          float* __restrict__ row_lse) {
    __shared__ float red[16];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t lab = shifted_label(labels, S, row);
    if (lab < 0 || lab >= V) {
        if (tid == 0) { row_loss[row] = 0.f; row_lse[row] = 0.f; }
        return;
    }
    const float* x = logits + (int64_t)row * ld;
    float v[CE_NPT];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < CE_NPT; ++i) {
        const int c = tid + CE_T * i;
        v[i] = c < V ? x[c] : -INFINITY;
    }
#pragma unroll
    for (int i = 0; i < CE_NPT; ++i) m = fmaxf(m, v[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    float M = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) M = fmaxf(M, red[w]);
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CE_NPT; ++i) s += __expf(v[i] - M);          // exp(-inf) = 0 for the padding
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int w = 0; w < 16; ++w) tot += red[w];
        const float lse = M + logf(tot);
        row_lse[row] = lse;
        row_loss[row] = lse - x[lab];
    }
}

__global__ __launch_bounds__(1024) void ce_reduce_kernel(int rows, int S, const float* row_loss, const int64_t* labels,
                                                         int V, float* loss, float* count) {
    __shared__ float sl[1024], sc[1024];
    float a = 0.f, c = 0.f;
    int bad = 0;                       // a label that is neither ignore_index nor a class: torch.cross_entropy asserts on the device
    for (int r = threadIdx.x; r < rows; r += 1024) {
        const int64_t lab = shifted_label(labels, S, r);
        if (lab >= 0 && lab < V) { a += row_loss[r]; c += 1.f; }
        else if (lab != -100) bad = 1;
    }
    bad = __syncthreads_or(bad);
    sl[threadIdx.x] = a; sc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
        __syncthreads();
    }
    // an out-of-range label (e.g. a tokenizer longer than the embedding matrix) poisons the loss instead of being skipped
    if (threadIdx.x == 0) { count[0] = sc[0]; loss[0] = bad ? __builtin_nanf("") : sl[0] / sc[0]; }
}

__global__ void guard_count_kernel(const int32_t* count, int capacity, float* loss) {
    if (count[0] > capacity) loss[0] = __builtin_nanf("");
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(int S, int V, const float* logits, int64_t ld, const int64_t* labels,
                                                     const float* row_lse, const float* count, const float* gscale,
                                                     T* dlogits, int64_t ldd) {
    const int row = blockIdx.x;
    const int64_t lab = shifted_label(labels, S, row);
    const bool keep = lab >= 0 && lab < V;
    const float g = keep ? gscale[0] / count[0] : 0.f;
    const float lse = keep ? row_lse[row] : 0.f;
    const float* x = logits + (int64_t)row * ld;
    T* d = dlogits + (int64_t)row * ldd;
    const int chunk0 = blockIdx.y * 1024 * 4;   // 4096 columns per block
    for (int c = chunk0 + threadIdx.x * 4; c < min((int)ldd, chunk0 + 4096); c += 1024) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = c + j;
            v[j] = (keep && col < V) ? g * (expf(x[col] - lse) - (col == lab ? 1.f : 0.f)) : 0.f;
        }
        if (c + 3 < ldd) elem<T>::st4(d + c, make_float4(v[0], v[1], v[2], v[3]));
        else
            for (int j = 0; j < 4; ++j)
                if (c + j < ldd) elem<T>::st(d + c + j, v[j]);
    }
}

// one 1024-thread workgroup per row: a decode step has at most 64 rows, the parallelism has to come from inside the row
constexpr int GP_THREADS = 1024;

__global__ __launch_bounds__(GP_THREADS) void greedy_pick_kernel(int V, const float* logits, int64_t ld, int64_t pad, int64_t eos,
                                                          int32_t* raw, int64_t* emitted, int64_t ld_emitted,
                                                          int32_t* unfinished, float* logprob, int32_t* any_unfinished) {
    __shared__ float sv[GP_THREADS];
    __shared__ int si[GP_THREADS];
    const int b = blockIdx.x;
    const float* x = logits + (int64_t)b * ld;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int c = threadIdx.x; c < V; c += GP_THREADS) {
        const float v = x[c];
        if (v > best || idx == 0x7fffffff) { best = v; idx = c; }   // strict >: first maximal index wins
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = idx;
    __syncthreads();
    for (int o = GP_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float v2 = sv[threadIdx.x + o];
            const int i2 = si[threadIdx.x + o];
            if (v2 > sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x])) { sv[threadIdx.x] = v2; si[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int r = si[0];
        raw[b] = r;
        int64_t e = r;
        if (eos >= 0) {                          // clipcap.py:426-434 (finished rows emit pad)
            const int u = unfinished[b];
            e = u ? (int64_t)r : pad;
            const int u2 = u * (e != eos ? 1 : 0);    // :458-461
            unfinished[b] = u2;
            // "some row is still generating" for the host's early-stop check (:463), without a reduction kernel: the caller zeroes the
            // word, every row that goes on ORs a one into it (idempotent: no order dependence)
            if (any_unfinished && u2) atomicOr(any_unfinished, 1);
        }
        emitted[(int64_t)b * ld_emitted] = e;
    }
    if (logprob) {     // log_softmax(logits[b])[argmax] = -log(sum exp(x - max)); fixed-order tree
        const float mx = sv[0];
        __syncthreads();
        float sum = 0.f;
        for (int c = threadIdx.x; c < V; c += GP_THREADS) sum += expf(x[c] - mx);
        sv[threadIdx.x] = sum;
        __syncthreads();
        for (int o = GP_THREADS / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sv[threadIdx.x] += sv[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) logprob[b] = -logf(sv[0]);
    }
}

}  // namespace

extern "C" int eavqa_ce_fwd(int B, int S, int V, const float* logits, int64_t ld, const int64_t* labels, float* row_loss,
                            float* row_lse, float* loss, float* count, void* stream) {
    if (B <= 0 || S < 0 || V <= 0 || !logits || !labels || !row_loss || !row_lse || !loss || !count) return EAVQA_E_ARG;
    if (ld < V) return EAVQA_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rows = S > 0 ? B * S : B;
    if (V <= CE_T * CE_NPT) hipLaunchKernelGGL(ce_fwd_row_kernel, dim3(rows), dim3(CE_T), 0, s, S, V, logits, ld, labels, row_loss, row_lse);
    else hipLaunchKernelGGL(ce_fwd_kernel, dim3(rows), dim3(256), 0, s, S, V, logits, ld, labels, row_loss, row_lse);
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, s, rows, S, row_loss, labels, V, loss, count);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_guard_count(const int32_t* count, int capacity, float* loss, void* stream) {
    if (!count || !loss || capacity < 0) return EAVQA_E_ARG;
    hipLaunchKernelGGL(guard_count_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), count, capacity, loss);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_ce_bwd(int dtype, int B, int S, int V, const float* logits, int64_t ld, const int64_t* labels,
                            const float* row_lse, const float* count, const float* gscale, void* dlogits, int64_t ldd,
                            void* stream) {
    if (B <= 0 || S < 0 || V <= 0 || !logits || !labels || !row_lse || !count || !gscale || !dlogits) return EAVQA_E_ARG;
    if (ld < V || ldd < V) return EAVQA_E_ARG;
    if (ldd % 4) return EAVQA_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(S > 0 ? B * S : B, (unsigned)((ldd + 4095) / 4096));
    if (dtype == EAVQA_F32)
        hipLaunchKernelGGL(ce_bwd_kernel<float>, grid, dim3(256), 0, s, S, V, logits, ld, labels, row_lse, count, gscale, (float*)dlogits, ldd);
    else if (dtype == EAVQA_BF16)
        hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, S, V, logits, ld, labels, row_lse, count, gscale, (bf16_t*)dlogits, ldd);
    else return EAVQA_E_DTYPE;
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_greedy_pick(int B, int V, const float* logits, int64_t ld, int64_t pad_token_id, int64_t eos_token_id,
                                 int32_t* raw, int64_t* emitted, int64_t ld_emitted, int32_t* unfinished, float* logprob,
                                 int32_t* any_unfinished, void* stream) {
    if (B <= 0 || V <= 0 || !logits || !raw || !emitted) return EAVQA_E_ARG;
    if (eos_token_id >= 0 && !unfinished) return EAVQA_E_ARG;
    if (ld < V) return EAVQA_E_ARG;
    hipLaunchKernelGGL(greedy_pick_kernel, dim3(B), dim3(GP_THREADS), 0, reinterpret_cast<hipStream_t>(stream), V, logits, ld,
                       pad_token_id, eos_token_id, raw, emitted, ld_emitted, unfinished, logprob, any_unfinished);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}
