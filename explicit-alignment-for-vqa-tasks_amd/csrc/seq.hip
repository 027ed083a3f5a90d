// Sequence assembly: the integer / index work around the LM (bit-exact against the oracle) and
// the HBM-bound gathers that build the residual stream.  One wavefront per batch row does the
// prefix scans (sentinel counts, OPT position ids, first-pad / seen-<BOS> flags).
#include "common.h"
#include <algorithm>

namespace {

// inclusive scan over the 64 lanes of a wave
__device__ __forceinline__ int wave_iscan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// positions from a mask row held in LDS: GPT-2 arange, OPT cumsum(mask)*mask - 1 + 2
__device__ __forceinline__ void write_positions(const int* mrow, int S, int pos_mode, int32_t* pos_row) {
    const int lane = threadIdx.x & 63;
    int carry = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const int mv = (s < S) ? mrow[s] : 0;
        const int inc = wave_iscan(mv) + carry;
        if (s < S) pos_row[s] = pos_mode == 0 ? s : (inc * mv - 1 + 2);
        carry = __shfl(inc, 63, 64);
    }
}

__global__ __launch_bounds__(64) void prefix_rows_kernel(int L, int T, const int64_t* tokens, const int64_t* qmask,
                                                         int pos_mode, int pstride, int poff, int32_t* src,
                                                         int32_t* mask_out, int32_t* pos) {
    extern __shared__ int lds_mask[];
    const int b = blockIdx.x, S = L + T, lane = threadIdx.x;
    for (int s = lane; s < S; s += 64) {
        int sv, mv;
        if (s < L) { sv = -(1 + b * pstride + poff + s); mv = 1; }
        else { sv = (int)tokens[(int64_t)b * T + (s - L)]; mv = qmask[(int64_t)b * T + (s - L)] != 0 ? 1 : 0; }
        src[(int64_t)b * S + s] = sv;
        mask_out[(int64_t)b * S + s] = mv;
        lds_mask[s] = mv;
    }
    __syncthreads();
    write_positions(lds_mask, S, pos_mode, pos + (int64_t)b * S);
}

__global__ __launch_bounds__(64) void fewshot_rows_kernel(int T, int L, int n_img, int64_t special, const int64_t* tokens,
                                                          const int64_t* qmask, int pos_mode, int32_t* src,
                                                          int32_t* mask_out, int32_t* pos, int32_t* status) {
    extern __shared__ int lds_mask[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int T_out = T + (L - 1) * n_img;
    // pad everything first: rows with too few sentinels leave a defined tail
    for (int s = lane; s < T_out; s += 64) { src[(int64_t)b * T_out + s] = 0; lds_mask[s] = 0; }
    __syncthreads();
    int carry = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        int64_t tok = 0;
        int is_s = 0;
        if (t < T) {
            tok = tokens[(int64_t)b * T + t];
            is_s = (tok <= special && tok > special - n_img) ? 1 : 0;
        }
        const int inc = wave_iscan(is_s) + carry;
        const int before = inc - is_s;                  // sentinels strictly before t
        if (t < T) {
            const int o = t + before * (L - 1);
            if (is_s) {
                if (before < n_img) {
                    for (int l = 0; l < L; ++l) {
                        src[(int64_t)b * T_out + o + l] = -(1 + (b * n_img + before) * L + l);
                        lds_mask[o + l] = 1;
                    }
                }
            } else if (o < T_out) {
                src[(int64_t)b * T_out + o] = (int)tok;
                lds_mask[o] = qmask[(int64_t)b * T + t] != 0 ? 1 : 0;
            }
        }
        carry = __shfl(inc, 63, 64);
    }
    __syncthreads();
    if (lane == 0) status[b] = carry;
    for (int s = lane; s < T_out; s += 64) mask_out[(int64_t)b * T_out + s] = lds_mask[s];
    write_positions(lds_mask, T_out, pos_mode, pos + (int64_t)b * T_out);
}

template <typename T>
__global__ __launch_bounds__(256) void embed_assemble_kernel(int E, const int32_t* src, const int32_t* pos,
                                                             const T* wte, int64_t ld_wte, const T* prefix, int64_t ld_prefix,
                                                             const T* wpe, int64_t ld_wpe, float* x, int64_t ldx) {
    const int row = blockIdx.x;
    const int sv = src[row];
    const T* e = sv >= 0 ? wte + (int64_t)sv * ld_wte : prefix + (int64_t)(-sv - 1) * ld_prefix;
    const T* pe = wpe ? wpe + (int64_t)pos[row] * ld_wpe : nullptr;
    for (int c = threadIdx.x * 4; c < E; c += 256 * 4) {
        float4 v = elem<T>::ld4(e + c);
        if (pe) { const float4 w = elem<T>::ld4(pe + c); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        *reinterpret_cast<float4*>(x + (int64_t)row * ldx + c) = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_assemble_bwd_kernel(int E, const int32_t* src, const float* dx, int64_t lddx,
                                                                 T* dprefix, int64_t ld_dprefix) {
    const int row = blockIdx.x;
    const int sv = src[row];
    if (sv >= 0) return;
    T* d = dprefix + (int64_t)(-sv - 1) * ld_dprefix;
    for (int c = threadIdx.x * 4; c < E; c += 256 * 4)
        elem<T>::st4(d + c, *reinterpret_cast<const float4*>(dx + (int64_t)row * lddx + c));
}

// labels: one wave per row; flags via prefix scans
__global__ __launch_bounds__(64) void labels_kernel(int mode, int T, int L, const int64_t* ids, int64_t pad, int64_t bos,
                                                    int64_t* out) {
    const int b = blockIdx.x, lane = threadIdx.x, S = L + T;
    for (int s = lane; s < L; s += 64) out[(int64_t)b * S + s] = -100;
    int pad_seen = 0, bos_seen = 0;  // inclusive counts carried across chunks
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const int64_t tok = t < T ? ids[(int64_t)b * T + t] : pad - 1;
        const int is_pad = (t < T && tok == pad) ? 1 : 0;
        const int is_bos = (t < T && tok == bos && !is_pad) ? 1 : 0;
        const int pad_inc = wave_iscan(is_pad) + pad_seen;
        const int bos_inc = wave_iscan(is_bos) + bos_seen;
        if (t < T) {
            int64_t lab;
            if (mode == 1) {
                lab = is_pad ? -100 : tok;
            } else {
                const int pads_before = pad_inc - is_pad;
                const int bos_before = bos_inc - is_bos;
                if (pads_before == 0) {                       // at or before the first pad
                    if (is_pad) lab = pad;                    // first -100 restored to the pad (= eos) id
                    else if (is_bos) lab = -100;
                    else lab = bos_before > 0 ? tok : -100;   // answer tokens kept, question masked
                } else {
                    lab = is_pad ? -100 : tok;                // loop already broke: left as initialised
                }
            }
            out[(int64_t)b * S + L + t] = lab;
        }
        pad_seen = __shfl(pad_inc, 63, 64);
        bos_seen = __shfl(bos_inc, 63, 64);
    }
}

// One thread per PIXEL (reads fully coalesced along x; a patch row of ps pixels lands contiguously in its patch): the round-1 kernel
// walked the patch's columns and gathered single floats, 2.3 ms for 32 images at patch 14 (ViT-L/14).
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(int64_t total, int img, int ps, int g, const float* __restrict__ px, T* __restrict__ out,
                                                       int64_t ldp) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % img);
    const int64_t t = idx / img;
    const int y = (int)(t % img);
    const int64_t t2 = t / img;
    const int ch = (int)(t2 % 3), b = (int)(t2 / 3);
    const int gy = y / ps, gx = x / ps;
    if (gy >= g || gx >= g) return;                              // pixels beyond the last whole patch (img % ps != 0)
    const int64_t row = ((int64_t)b * g + gy) * g + gx;
    const int col = (ch * ps + (y - gy * ps)) * ps + (x - gx * ps);
    elem<T>::st(out + row * ldp + col, px[idx]);
}

template <typename T>
__global__ __launch_bounds__(256) void zero_cols_kernel(int64_t rows, int c0, int c1, T* out, int64_t ldp) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int w = c1 - c0;
    if (idx >= rows * w) return;
    elem<T>::st(out + (idx / w) * ldp + c0 + (int)(idx % w), 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void vit_assemble_kernel(int n_patch, int W, const T* pe, int64_t ldpe, const float* cls,
                                                           const float* pos, float* x, int64_t ldx) {
    const int row = blockIdx.x;                // b*(n_patch+1) + t
    const int N = n_patch + 1;
    const int b = row / N, t = row - b * N;
    for (int c = threadIdx.x * 4; c < W; c += 256 * 4) {
        float4 v = t == 0 ? *reinterpret_cast<const float4*>(cls + c)
                          : elem<T>::ld4(pe + ((int64_t)b * n_patch + (t - 1)) * ldpe + c);
        const float4 w = *reinterpret_cast<const float4*>(pos + (int64_t)t * W + c);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        *reinterpret_cast<float4*>(x + (int64_t)row * ldx + c) = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_rows_kernel(int64_t cols, const float* x, int64_t ldx, T* y, int64_t ldy) {
    const int row = blockIdx.x;   // blockIdx.y walks 1024-column chunks so that one very long row still fills the chip
    for (int64_t c = (int64_t)blockIdx.y * 1024 + threadIdx.x * 4; c < cols; c += (int64_t)gridDim.y * 1024)
        elem<T>::st4(y + (int64_t)row * ldy + c, *reinterpret_cast<const float4*>(x + (int64_t)row * ldx + c));
}

template <typename T>
__global__ __launch_bounds__(256) void copy_rows_kernel(int S, int cols, const T* src, int64_t lds, int64_t sbr, T* dst,
                                                        int64_t ldd, int64_t dbr, int64_t drow0) {
    const int row = blockIdx.x;               // b*S + s
    const int b = row / S, s = row - b * S;
    const T* x = src + ((int64_t)b * sbr + s) * lds;
    T* y = dst + ((int64_t)b * dbr + drow0 + s) * ldd;
    constexpr int V = 16 / sizeof(T);         // 16-byte vectors when cols allows, else 4 elements
    if ((cols % V) == 0 && (lds % V) == 0 && (ldd % V) == 0) {
        for (int c = threadIdx.x * V; c < cols; c += 256 * V)
            *reinterpret_cast<uint4*>(y + c) = *reinterpret_cast<const uint4*>(x + c);
    } else {
        for (int c = threadIdx.x * 4; c < cols; c += 256 * 4) elem<T>::st4(y + c, elem<T>::ld4(x + c));
    }
}

// column sums: block = 64 columns x 4 row-groups, rows strided over blockIdx.y, fp32 atomics between blocks
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(int rows, int cols, const T* x, int64_t ldx, float* out) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    float a = 0.f;
    if (c < cols)
        for (int r = blockIdx.y * 4 + rg; r < rows; r += gridDim.y * 4) a += elem<T>::ld(x + (int64_t)r * ldx + c);
    part[rg][threadIdx.x & 63] = a;
    __syncthreads();
    if (rg == 0 && c < cols) {
        const float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (gridDim.y == 1) out[c] += t; else atomicAdd(out + c, t);
    }
}

// Row plan for the training forward: keep only attended positions (mask != 0) of each sample, in order.
// One 256-thread block: per-sample counts -> exclusive scan (cu_seqlens) -> per-sample compaction.
__global__ __launch_bounds__(256) void row_plan_kernel(int B, int S, int pack, const int32_t* mask, const int64_t* labels,
                                                       const int32_t* src, const int32_t* pos, int32_t* cu, int32_t* src_p,
                                                       int32_t* pos_p, int64_t* row_labels, int32_t* flat_index) {
    extern __shared__ int counts[];          // [B + 1]
    for (int b = threadIdx.x; b < B; b += 256) {
        int c = S;
        if (pack) {
            c = 0;
            for (int s = 0; s < S; ++s) c += mask[(int64_t)b * S + s] != 0 ? 1 : 0;
        }
        counts[b + 1] = c;
    }
    if (threadIdx.x == 0) counts[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int b = 1; b <= B; ++b) counts[b] += counts[b - 1];   // B <= a few hundred: serial scan is fine
    __syncthreads();
    for (int b = threadIdx.x; b <= B; b += 256) cu[b] = counts[b];
    // one wave per sample walks its row with a running offset
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int b = wave; b < B; b += 4) {
        int carry = counts[b];
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + lane;
            const int keep = (s < S) && (!pack || mask[(int64_t)b * S + s] != 0) ? 1 : 0;
            const int inc = wave_iscan(keep) + carry;
            if (keep) {
                const int r = inc - 1;
                const int64_t f = (int64_t)b * S + s;
                src_p[r] = src[f];
                pos_p[r] = pos[f];
                flat_index[r] = (int32_t)f;
                if (row_labels) row_labels[r] = (labels && s + 1 < S) ? labels[f + 1] : -100;   // shift (loss_utils.py:61-63)
            }
            carry = __shfl(inc, 63, 64);
        }
    }
}

// dst[c][r] = src[r][c] through a 64 x 64 LDS tile (pitch 65 elements): both sides move whole 128-byte rows
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(int rows, int cols, const T* src, int64_t lds_, T* dst, int64_t ldd) {
    __shared__ T tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[(int64_t)(r0 + i) * lds_ + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 64; i += 4)
        if (c0 + i < cols && r0 + tx < rows) dst[(int64_t)(c0 + i) * ldd + r0 + tx] = tile[tx][i];
}

// 16-bit elements, everything a multiple of 4: 8-byte global loads and stores (128 contiguous bytes per 16 lanes on both sides) through
// an LDS tile of 32-bit words that pair two neighbouring SOURCE rows, i.e. two neighbouring elements of a destination row:
// word[c][r / 2] = (src[r][c], src[r + 1][c]), pitch 33 words (2-way bank conflicts at worst).  The scalar kernel above moves 2 bytes
// per lane and instruction: 1.2 TB/s on the [2 048, 4 096] activations of the transformer mapper's weight gradients, this one ~3x.
__global__ __launch_bounds__(256) void transpose16_kernel(int rows, int cols, const uint16_t* __restrict__ src, int64_t lds_,
                                                          uint16_t* __restrict__ dst, int64_t ldd) {
    __shared__ uint32_t tile[64 * 33];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const int cq = lane & 15;                                   // column quad: columns c0 + 4 cq .. +3
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int rp = pass * 16 + wave * 4 + (lane >> 4);      // row pair: rows r0 + 2 rp, + 1
            const int r = r0 + 2 * rp, c = c0 + 4 * cq;
            uint2 lo = make_uint2(0u, 0u), hi = lo;
            if (r < rows && c < cols) {
                lo = *reinterpret_cast<const uint2*>(src + (int64_t)r * lds_ + c);
                if (r + 1 < rows) hi = *reinterpret_cast<const uint2*>(src + (int64_t)(r + 1) * lds_ + c);
            }
            uint32_t* t = tile + (4 * cq) * 33 + rp;
            t[0] = (lo.x & 0xFFFFu) | (hi.x << 16);
            t[33] = (lo.x >> 16) | (hi.x & 0xFFFF0000u);
            t[66] = (lo.y & 0xFFFFu) | (hi.y << 16);
            t[99] = (lo.y >> 16) | (hi.y & 0xFFFF0000u);
        }
    }
    __syncthreads();
    {
        const int rq = lane & 15;                                   // destination columns r0 + 4 rq .. +3
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int c = pass * 16 + wave * 4 + (lane >> 4);
            if (c0 + c < cols && r0 + 4 * rq < rows) {
                const uint32_t* t = tile + c * 33 + 2 * rq;
                *reinterpret_cast<uint2*>(dst + (int64_t)(c0 + c) * ldd + r0 + 4 * rq) = make_uint2(t[0], t[1]);
            }
        }
    }
}

// ---- scored rows: only rows that carry a label reach the lm_head (2 E V FLOP per row forward, the same again for its
// dgrad; on Conceptual-Captions batches a third of the packed rows - the prefix and the last token of every caption -
// carry none).  One 1024-thread workgroup compacts the row numbers in order (ballot + popcount per wave, scan over the
// 16 waves per chunk), so the result does not depend on scheduling.
__global__ __launch_bounds__(1024) void select_rows_kernel(int M, const int64_t* __restrict__ row_labels, int cap,
                                                           int32_t* __restrict__ sel_idx, int64_t* __restrict__ sel_labels,
                                                           int32_t* __restrict__ count) {
    __shared__ int wave_n[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int r0 = 0; r0 < M; r0 += 1024) {
        const int r = r0 + tid;
        const int64_t lab = r < M ? row_labels[r] : -100;
        const bool keep = lab >= 0;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_n[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_n[w];
        const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
        if (keep && pos < cap) { sel_idx[pos] = r; sel_labels[pos] = lab; }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_n[w]; base += t; }
        __syncthreads();
    }
    if (tid == 0 && count) count[0] = base;
}

// dst[i, :] = src[idx[i], :] (gather) or dst[idx[i], :] = src[i, :] (scatter), 16-byte vectors
template <typename T, bool SCATTER>
__global__ __launch_bounds__(256) void move_rows_kernel(int cols, const T* __restrict__ src, int64_t lds, const int32_t* __restrict__ idx,
                                                        T* __restrict__ dst, int64_t ldd) {
    const int i = blockIdx.x, j = idx[i];
    const T* x = src + (int64_t)(SCATTER ? i : j) * lds;
    T* y = dst + (int64_t)(SCATTER ? j : i) * ldd;
    constexpr int V = 16 / sizeof(T);
    for (int c = threadIdx.x * V; c < cols; c += 256 * V) *reinterpret_cast<uint4*>(y + c) = *reinterpret_cast<const uint4*>(x + c);
}

__global__ void zero_kernel(int n, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 0.f;
}

}  // namespace

extern "C" int eavqa_build_row_plan(int B, int S, int pack, const int32_t* mask, const int64_t* labels, const int32_t* src,
                                    const int32_t* pos, int32_t* cu_seqlens, int32_t* src_rows, int32_t* pos_rows,
                                    int64_t* row_labels, int32_t* flat_index, void* stream) {
    if (B <= 0 || S <= 0 || !mask || !src || !pos || !cu_seqlens || !src_rows || !pos_rows || !flat_index) return EAVQA_E_ARG;
    if ((size_t)(B + 1) * 4 > 60 * 1024) return EAVQA_E_SHAPE;
    hipLaunchKernelGGL(row_plan_kernel, dim3(1), dim3(256), (size_t)(B + 1) * 4, reinterpret_cast<hipStream_t>(stream), B, S, pack,
                       mask, labels, src, pos, cu_seqlens, src_rows, pos_rows, row_labels, flat_index);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_copy_rows(int dtype, int B, int S, int cols, const void* src, int64_t lds, int64_t src_batch_rows,
                               void* dst, int64_t ldd, int64_t dst_batch_rows, int64_t dst_row0, void* stream) {
    if (B <= 0 || S <= 0 || cols <= 0 || !src || !dst) return EAVQA_E_ARG;
    if (cols % 4 || lds % 4 || ldd % 4) return EAVQA_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_F32)
        hipLaunchKernelGGL(copy_rows_kernel<float>, dim3(B * S), dim3(256), 0, s, S, cols, (const float*)src, lds, src_batch_rows,
                           (float*)dst, ldd, dst_batch_rows, dst_row0);
    else if (dtype == EAVQA_BF16)
        hipLaunchKernelGGL(copy_rows_kernel<bf16_t>, dim3(B * S), dim3(256), 0, s, S, cols, (const bf16_t*)src, lds, src_batch_rows,
                           (bf16_t*)dst, ldd, dst_batch_rows, dst_row0);
    else return EAVQA_E_DTYPE;
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_select_rows(int M, const int64_t* row_labels, int capacity, int32_t* sel_idx, int64_t* sel_labels, int32_t* count,
                                 void* stream) {
    if (M <= 0 || capacity < 0 || !row_labels || !sel_idx || !sel_labels) return EAVQA_E_ARG;
    hipLaunchKernelGGL(select_rows_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), M, row_labels, capacity, sel_idx,
                       sel_labels, count);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_move_rows(int dtype, int scatter, int n, int cols, const void* src, int64_t ld_src, const int32_t* idx, void* dst,
                               int64_t ld_dst, void* stream) {
    if (n < 0 || cols <= 0 || !src || !idx || !dst) return EAVQA_E_ARG;
    if (n == 0) return EAVQA_OK;
    const int vec = dtype == EAVQA_BF16 ? 8 : 4;
    if (dtype != EAVQA_BF16 && dtype != EAVQA_F32) return EAVQA_E_DTYPE;
    if (cols % vec || ld_src % vec || ld_dst % vec || !eavqa_aligned16(src) || !eavqa_aligned16(dst)) return EAVQA_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_BF16) {
        if (scatter) hipLaunchKernelGGL((move_rows_kernel<bf16_t, true>), dim3(n), dim3(256), 0, s, cols, (const bf16_t*)src, ld_src, idx, (bf16_t*)dst, ld_dst);
        else hipLaunchKernelGGL((move_rows_kernel<bf16_t, false>), dim3(n), dim3(256), 0, s, cols, (const bf16_t*)src, ld_src, idx, (bf16_t*)dst, ld_dst);
    } else {
        if (scatter) hipLaunchKernelGGL((move_rows_kernel<float, true>), dim3(n), dim3(256), 0, s, cols, (const float*)src, ld_src, idx, (float*)dst, ld_dst);
        else hipLaunchKernelGGL((move_rows_kernel<float, false>), dim3(n), dim3(256), 0, s, cols, (const float*)src, ld_src, idx, (float*)dst, ld_dst);
    }
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_transpose(int dtype, int rows, int cols, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, void* stream) {
    if (rows <= 0 || cols <= 0 || !src || !dst || ld_src < cols || ld_dst < rows) return EAVQA_E_ARG;
    dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_F32) hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, s, rows, cols, (const float*)src, ld_src, (float*)dst, ld_dst);
    else if (dtype == EAVQA_BF16 && !((rows | cols | ld_src | ld_dst) & 3) && !((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 7))
        hipLaunchKernelGGL(transpose16_kernel, grid, dim3(256), 0, s, rows, cols, (const uint16_t*)src, ld_src, (uint16_t*)dst, ld_dst);
    else if (dtype == EAVQA_BF16) hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, s, rows, cols, (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst);
    else return EAVQA_E_DTYPE;
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_colsum(int dtype, int rows, int cols, const void* x, int64_t ldx, float* out, int accumulate, void* stream) {
    if (rows <= 0 || cols <= 0 || !x || !out) return EAVQA_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!accumulate) hipLaunchKernelGGL(zero_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, cols, out);
    int gy = (rows + 255) / 256;          // ~64 rows per row-group per block
    if (gy > 64) gy = 64;
    dim3 grid((cols + 63) / 64, gy);
    if (dtype == EAVQA_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, rows, cols, (const float*)x, ldx, out);
    else if (dtype == EAVQA_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, rows, cols, (const bf16_t*)x, ldx, out);
    else return EAVQA_E_DTYPE;
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_build_prefix_rows(int B, int L, int T, const int64_t* tokens, const int64_t* question_mask,
                                       int pos_mode, int prefix_row_stride, int prefix_row_offset,
                                       int32_t* src, int32_t* mask_out, int32_t* pos, void* stream) {
    if (B <= 0 || L < 0 || T < 0 || L + T <= 0 || !src || !mask_out || !pos) return EAVQA_E_ARG;
    if (prefix_row_stride < L || prefix_row_offset < 0) return EAVQA_E_ARG;
    if (T > 0 && (!tokens || !question_mask)) return EAVQA_E_ARG;
    if ((size_t)(L + T) * 4 > 60 * 1024) return EAVQA_E_SHAPE;
    hipLaunchKernelGGL(prefix_rows_kernel, dim3(B), dim3(64), (size_t)(L + T) * 4, reinterpret_cast<hipStream_t>(stream),
                       L, T, tokens, question_mask, pos_mode, prefix_row_stride, prefix_row_offset, src, mask_out, pos);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_build_fewshot_rows(int B, int T, int L, int n_img, int64_t special_token_id, const int64_t* tokens,
                                        const int64_t* question_mask, int pos_mode, int32_t* src, int32_t* mask_out,
                                        int32_t* pos, int32_t* status, void* stream) {
    if (B <= 0 || T <= 0 || L <= 0 || n_img <= 0 || !tokens || !question_mask || !src || !mask_out || !pos || !status)
        return EAVQA_E_ARG;
    const size_t T_out = (size_t)T + (size_t)(L - 1) * n_img;
    if (T_out * 4 > 60 * 1024) return EAVQA_E_SHAPE;
    hipLaunchKernelGGL(fewshot_rows_kernel, dim3(B), dim3(64), T_out * 4, reinterpret_cast<hipStream_t>(stream), T, L, n_img,
                       special_token_id, tokens, question_mask, pos_mode, src, mask_out, pos, status);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_embed_assemble(int dtype, int rows, int E, const int32_t* src, const int32_t* pos, const void* wte,
                                    int64_t ld_wte, const void* prefix_rows, int64_t ld_prefix, const void* wpe,
                                    int64_t ld_wpe, float* x, int64_t ldx, void* stream) {
    if (rows <= 0 || E <= 0 || !src || !wte || !x) return EAVQA_E_ARG;
    if (wpe && !pos) return EAVQA_E_ARG;
    if (E % 4 || ld_wte % 4 || ldx % 4 || (prefix_rows && ld_prefix % 4) || (wpe && ld_wpe % 4)) return EAVQA_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_F32)
        hipLaunchKernelGGL(embed_assemble_kernel<float>, dim3(rows), dim3(256), 0, s, E, src, pos, (const float*)wte, ld_wte,
                           (const float*)prefix_rows, ld_prefix, (const float*)wpe, ld_wpe, x, ldx);
    else if (dtype == EAVQA_BF16)
        hipLaunchKernelGGL(embed_assemble_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, E, src, pos, (const bf16_t*)wte, ld_wte,
                           (const bf16_t*)prefix_rows, ld_prefix, (const bf16_t*)wpe, ld_wpe, x, ldx);
    else return EAVQA_E_DTYPE;
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_embed_assemble_bwd(int dtype, int rows, int E, const int32_t* src, const float* dx, int64_t lddx,
                                        void* dprefix, int64_t ld_dprefix, void* stream) {
    if (rows <= 0 || E <= 0 || !src || !dx || !dprefix) return EAVQA_E_ARG;
    if (E % 4 || lddx % 4 || ld_dprefix % 4) return EAVQA_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_F32)
        hipLaunchKernelGGL(embed_assemble_bwd_kernel<float>, dim3(rows), dim3(256), 0, s, E, src, dx, lddx, (float*)dprefix, ld_dprefix);
    else if (dtype == EAVQA_BF16)
        hipLaunchKernelGGL(embed_assemble_bwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, E, src, dx, lddx, (bf16_t*)dprefix, ld_dprefix);
    else return EAVQA_E_DTYPE;
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_build_labels(int mode, int B, int T, int L, const int64_t* input_ids, int64_t pad_token_id,
                                  int64_t bos_token_id, int64_t* labels_out, void* stream) {
    if (B <= 0 || T <= 0 || L < 0 || !input_ids || !labels_out) return EAVQA_E_ARG;
    if (mode != 0 && mode != 1) return EAVQA_E_ARG;
    hipLaunchKernelGGL(labels_kernel, dim3(B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), mode, T, L, input_ids,
                       pad_token_id, bos_token_id, labels_out);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_patchify(int dtype, int B, int img, int ps, const float* pixels, void* patches, int64_t ldp, void* stream) {
    if (B <= 0 || img <= 0 || ps <= 0 || !pixels || !patches) return EAVQA_E_ARG;
    if (img % ps) return EAVQA_E_SHAPE;
    if (ldp < 3 * ps * ps) return EAVQA_E_ARG;
    const int g = img / ps;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * 3 * img * img, rows = (int64_t)B * g * g;
    const int kreal = 3 * ps * ps;
    const unsigned blocks = (unsigned)((total + 255) / 256), zblocks = (unsigned)((rows * (ldp - kreal) + 255) / 256);
    if (dtype == EAVQA_F32) {
        hipLaunchKernelGGL(patchify_kernel<float>, dim3(blocks), dim3(256), 0, s, total, img, ps, g, pixels, (float*)patches, ldp);
        if (ldp > kreal) hipLaunchKernelGGL(zero_cols_kernel<float>, dim3(zblocks), dim3(256), 0, s, rows, kreal, (int)ldp, (float*)patches, ldp);
    } else if (dtype == EAVQA_BF16) {
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, total, img, ps, g, pixels, (bf16_t*)patches, ldp);
        if (ldp > kreal) hipLaunchKernelGGL(zero_cols_kernel<bf16_t>, dim3(zblocks), dim3(256), 0, s, rows, kreal, (int)ldp, (bf16_t*)patches, ldp);
    } else return EAVQA_E_DTYPE;
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_vit_assemble(int dtype, int B, int n_patch, int W, const void* patch_embed, int64_t ldpe,
                                  const float* cls, const float* pos, float* x, int64_t ldx, void* stream) {
    if (B <= 0 || n_patch <= 0 || W <= 0 || !patch_embed || !cls || !pos || !x) return EAVQA_E_ARG;
    if (W % 4 || ldpe % 4 || ldx % 4) return EAVQA_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rows = B * (n_patch + 1);
    if (dtype == EAVQA_F32)
        hipLaunchKernelGGL(vit_assemble_kernel<float>, dim3(rows), dim3(256), 0, s, n_patch, W, (const float*)patch_embed, ldpe, cls, pos, x, ldx);
    else if (dtype == EAVQA_BF16)
        hipLaunchKernelGGL(vit_assemble_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, n_patch, W, (const bf16_t*)patch_embed, ldpe, cls, pos, x, ldx);
    else return EAVQA_E_DTYPE;
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_cast_rows(int dtype, int rows, int64_t cols, const float* x, int64_t ldx, void* y, int64_t ldy, void* stream) {
    if (rows <= 0 || cols <= 0 || !x || !y) return EAVQA_E_ARG;
    if (cols % 4 || ldx % 4 || ldy % 4) return EAVQA_E_ALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int64_t chunks = (cols + 1023) / 1024;
    const int64_t want = (2048 + rows - 1) / rows;          // ~2048 blocks in total
    if (chunks > want) chunks = want;
    if (chunks > 65535) chunks = 65535;
    dim3 grid(rows, (unsigned)chunks);
    if (dtype == EAVQA_F32)
        hipLaunchKernelGGL(cast_rows_kernel<float>, grid, dim3(256), 0, s, cols, x, ldx, (float*)y, ldy);
    else if (dtype == EAVQA_BF16)
        hipLaunchKernelGGL(cast_rows_kernel<bf16_t>, grid, dim3(256), 0, s, cols, x, ldx, (bf16_t*)y, ldy);
    else return EAVQA_E_DTYPE;
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

// ---------------------------------------------------------------------------------------------------- gated activation (T5 v1.1 / T0)
// T5DenseGatedActDense (HF:models/t5/modeling_t5.py:97-123): h = act(wi_0 x) * (wi_1 x).  The two projections are ONE GEMM against the
// stacked weight [wi_0; wi_1] -> u [rows, 2 F]; forward: h[:, c] = act(u[:, c]) * u[:, F + c]; backward (frozen weights, dgrad only):
// du[:, c] = dh * u[:, F + c] * act'(u[:, c]),  du[:, F + c] = dh * act(u[:, c]).  HBM-bound elementwise passes, 4 columns per thread.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void gated_act_fwd_kernel(int rows, int F, int act, const T* u, int64_t ldu, T* h, int64_t ldh) {
    const int64_t n4 = (int64_t)rows * (F >> 2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (F >> 2);
        const int c = (int)(i - r * (F >> 2)) * 4;
        const float4 a = elem<T>::ld4(u + r * ldu + c), b = elem<T>::ld4(u + r * ldu + F + c);
        elem<T>::st4(h + r * ldh + c, make_float4(act_fwd(act, a.x) * b.x, act_fwd(act, a.y) * b.y, act_fwd(act, a.z) * b.z, act_fwd(act, a.w) * b.w));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gated_act_bwd_kernel(int rows, int F, int act, const T* u, int64_t ldu, const T* dh, int64_t lddh, T* du,
                                                            int64_t lddu) {
    const int64_t n4 = (int64_t)rows * (F >> 2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (F >> 2);
        const int c = (int)(i - r * (F >> 2)) * 4;
        const float4 a = elem<T>::ld4(u + r * ldu + c), b = elem<T>::ld4(u + r * ldu + F + c), d = elem<T>::ld4(dh + r * lddh + c);
        elem<T>::st4(du + r * lddu + c, make_float4(d.x * b.x * act_bwd(act, a.x), d.y * b.y * act_bwd(act, a.y), d.z * b.z * act_bwd(act, a.z),
                                                    d.w * b.w * act_bwd(act, a.w)));
        elem<T>::st4(du + r * lddu + F + c, make_float4(d.x * act_fwd(act, a.x), d.y * act_fwd(act, a.y), d.z * act_fwd(act, a.z), d.w * act_fwd(act, a.w)));
    }
}
}  // namespace

extern "C" int eavqa_gated_act_fwd(int dtype, int rows, int F, int act, const void* u, int64_t ldu, void* h, int64_t ldh, void* stream) {
    if (!u || !h || rows <= 0 || F <= 0) return EAVQA_E_ARG;
    if (F % 4 || ldu % 4 || ldh % 4 || ldu < 2 * F || ldh < F) return EAVQA_E_SHAPE;
    if (act < EAVQA_ACT_NONE || act > EAVQA_ACT_QUICK_GELU) return EAVQA_E_DTYPE;
    const int64_t n4 = (int64_t)rows * (F / 4);
    const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 256 * 8);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_F32) hipLaunchKernelGGL(gated_act_fwd_kernel<float>, dim3(blocks), dim3(256), 0, s, rows, F, act, reinterpret_cast<const float*>(u), ldu, reinterpret_cast<float*>(h), ldh);
    else if (dtype == EAVQA_BF16) hipLaunchKernelGGL(gated_act_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, rows, F, act, reinterpret_cast<const bf16_t*>(u), ldu, reinterpret_cast<bf16_t*>(h), ldh);
    else return EAVQA_E_DTYPE;
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

extern "C" int eavqa_gated_act_bwd(int dtype, int rows, int F, int act, const void* u, int64_t ldu, const void* dh, int64_t lddh, void* du,
                                   int64_t lddu, void* stream) {
    if (!u || !dh || !du || rows <= 0 || F <= 0) return EAVQA_E_ARG;
    if (F % 4 || ldu % 4 || lddh % 4 || lddu % 4 || ldu < 2 * F || lddu < 2 * F || lddh < F) return EAVQA_E_SHAPE;
    if (act < EAVQA_ACT_NONE || act > EAVQA_ACT_QUICK_GELU) return EAVQA_E_DTYPE;
    const int64_t n4 = (int64_t)rows * (F / 4);
    const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 256 * 8);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EAVQA_F32) hipLaunchKernelGGL(gated_act_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, rows, F, act, reinterpret_cast<const float*>(u), ldu, reinterpret_cast<const float*>(dh), lddh, reinterpret_cast<float*>(du), lddu);
    else if (dtype == EAVQA_BF16) hipLaunchKernelGGL(gated_act_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, rows, F, act, reinterpret_cast<const bf16_t*>(u), ldu, reinterpret_cast<const bf16_t*>(dh), lddh, reinterpret_cast<bf16_t*>(du), lddu);
    else return EAVQA_E_DTYPE;
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}
