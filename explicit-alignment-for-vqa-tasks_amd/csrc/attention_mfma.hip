// bf16 attention forward / backward on the matrix cores (v_mfma_f32_16x16x32_bf16), head dims 64 / 80 /
// 96 / 128, and any wider head dim (multiple of 8) when the problem is one 64 x 64 tile (the mapping network: see the
// "wide heads" section at the end).  Called from eavqa_attention_fwd / _bwd (attention.hip) for dtype bf16; the fp32 path
// and the other head dims stay on the vector-ALU kernels.
//
// One workgroup = 4 waves = 64 "lane items" of one (batch, head); a wave owns 16 of them, ONE PER LANE
// COLUMN (lane & 15), and streams the other sequence dimension ("register items") through LDS in tiles
// of 64.  Every product is oriented so that the register items are the MFMA row index:
//   forward / dQ pass : lane item = query, register item = key
//        S^T  = K  . Q^T      (A = K rows from LDS,        B = Q fragment held in registers)
//        O^T += V^T . P^T     (A = V^T from LDS,           B = P^T = the S^T accumulator itself)
//        dP^T = V  . dO^T,    dQ^T += K^T . dS^T
//   dK/dV pass        : lane item = key, register item = query
//        S = Q . K^T, dP = dO . V^T, dV^T += dO^T . P, dK^T += Q^T . dS
// An accumulator tile (rows in registers, column on the lane) is directly the B operand of the next
// product that sums over its ROW index, so P / dS never leave registers: registers 0-3 of two adjacent
// 16-row tiles form one 8-element B fragment; the A operand of that product is the staged tile TRANSPOSED,
// fetched from the row-major LDS image in the same permuted k order by two transposing reads
// (ds_read_b64_tr_b16) - no transposed copy is written.  Softmax statistics are per lane column:
// the reduction over register items is in-lane plus two wave shuffles (xor 16, 32).
// Masked scores are replaced by -FLT_MAX exactly as in the vector-ALU kernels.
#include "common.h"

namespace eavqa_attn_mfma {

struct Params {
    const void* q; const void* k; const void* v; const void* o; const void* d_o;
    void* out; void* dq; void* dk; void* dv;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    const int32_t* key_mask; int64_t ld_mask;
    const int32_t* cu;
    float* lse; float* delta;
    int B, H, Sq, Sk, hd, causal, stat_ld;
    int fused_padded;              // one-tile backward, hd 64: keep the round-2 padded-pitch kernel (eavqa_attention_bwd_ex path bit 2: A / B, parity)
    int64_t bsq, bsk;
    float scale;
    // T5 relative-position bias (forward, streamed-tile kernel only): score(i, j) += rel_bias[h * rel_ld + (j - (i + Sk - Sq)) + rel_zero]
    const float* rel_bias; int64_t rel_ld; int rel_zero;
};

// T5's additive relative-position bias of (head h, key position, query position counted from the end of the keys): 0 without a table;
// entries outside the table are clamped (they belong to masked positions only).  Used by the forward AND, since round 4, by the
// backward kernels (the frozen T5's dgrad recomputes P = softmax(q k^T scale + bias): T0_3B training spent 12 % of its step in the
// vector-ALU backward kernels because only they knew the bias).
__device__ __forceinline__ float rel_bias_at(const Params& p, int h, int key, int qpos) {
    if (!p.rel_bias) return 0.f;
    const int idx = min(max(key - qpos + p.rel_zero, 0), (int)p.rel_ld - 1);
    return p.rel_bias[(int64_t)h * p.rel_ld + idx];
}


constexpr int TILE = 64;

// s_waitcnt immediate that waits for vmcnt <= n only (lgkmcnt / expcnt untouched)
__device__ __host__ constexpr int vm_only_attn(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

template <int KS> struct Geo {
    static constexpr int HP = KS * 32;              // head dim padded to the MFMA k step
    static constexpr int PR = HP * 2 + 16;          // byte pitch of a row-major [item][d] image (16 B pad: bank spread)
    static constexpr int ROW_BYTES = TILE * PR;
};

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.f;
    return z;
}

// Stage 64 rows (items row0 .. row0+63 of the current sample/head) as a row-major [item][d] image (16-byte stores).
// The second argument is kept for call-site symmetry and must be null: products that need the tile transposed read it
// through ds_read_b64_tr_b16 (tile_accumulate) instead of keeping a transposed copy.
template <int KS>
__device__ __forceinline__ void stage(char* rowmaj, char* /*unused*/, const bf16_t* src, int64_t ld, int row0, int n_rows,
                                      int hd, int head_off) {
    constexpr int CH = Geo<KS>::HP / 8;             // 16-byte chunks per row
    for (int c = threadIdx.x; c < TILE * CH; c += blockDim.x) {
        const int r = c / CH, ch = c - r * CH;
        uint4 val = make_uint4(0u, 0u, 0u, 0u);
        if (row0 + r < n_rows && ch * 8 < hd) val = *reinterpret_cast<const uint4*>(src + (int64_t)(row0 + r) * ld + head_off + ch * 8);
        *reinterpret_cast<uint4*>(rowmaj + r * Geo<KS>::PR + ch * 16) = val;
    }
}

// B fragments of a lane item (one sequence row): lane holds d = 32 s + 8 g .. +7 of row `row`
template <int KS>
__device__ __forceinline__ void load_bfrag(bf16x8 (&f)[KS], const bf16_t* src, int64_t ld, int row, bool valid, int hd,
                                           int head_off, int g) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int d0 = 32 * s + 8 * g;
        f[s] = (valid && d0 < hd) ? *reinterpret_cast<const bf16x8*>(src + (int64_t)row * ld + head_off + d0) : zero8();
    }
}

// acc[f] (f = 0..3, 16 register items each) = sum_d A_tile[item 16 f + (lane & 15)][d] * bfrag[d]; only the first `nf`
// fragments are computed (a ragged last tile: 257 = 4 x 64 + 1 ViT tokens leave 63 of 64 register items empty), the rest
// read as zero
// SW (hd = 64 images of the streamed forward kernel): rows on a 128-byte pitch, 16-byte chunk c of row r stored at slot c ^ (r & 6) - the
// swizzle of the resident-K/V kernel, conflict-free for these row reads and for the transposing reads of tile_accumulate (the padded
// 144-byte pitch showed 22 % LDS conflict cycles on the LM forward, profiles/round3_bench_cfg2_bf16_mfma.md)
template <int KS, bool SW = false>
__device__ __forceinline__ void tile_dot(f32x4 (&acc)[4], const char* rowmaj, const bf16x8 (&bf)[KS], int x, int g, int nf = 4) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (f < nf) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 a = SW ? *reinterpret_cast<const bf16x8*>(rowmaj + (16 * f + x) * 128 + (((4 * s + g) ^ (x & 6)) << 4))
                                    : *reinterpret_cast<const bf16x8*>(rowmaj + (16 * f + x) * Geo<KS>::PR + (32 * s + 8 * g) * 2);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[s], acc[f], 0, 0, 0);
            }
        }
    }
}

// 4 x 16 transposing LDS read (ds_read_b64_tr_b16): within a 16-lane group, lane 4q + p supplies the address of row q,
// columns 4p .. 4p+3 of a block of 16-bit elements; lane i receives column i of the 4 rows (row q in element q).
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 lds_tr4(const char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(bf16x4, v);
}

// out[dm] += sum over the 64 register items of R[item][d = 16 dm + (lane & 15)] * w[item], w given as 4 accumulator
// tiles.  R is the ROW-MAJOR [item][d] image: the A operand (rows = d, k = items in the permuted order of the
// accumulator-as-B trick: items 32 s2 + 4 g .. +3 and 32 s2 + 16 + 4 g .. +3) comes out of two transposing reads, so no
// transposed copy of the tile is ever written.
template <int KS, int D16, bool SW = false>
__device__ __forceinline__ void tile_accumulate(f32x4 (&out)[D16], const char* rowmaj, const f32x4 (&w)[4], int x, int g,
                                                int nf = 4) {
    const int q = x >> 2, pp = x & 3;
    const int sw = (4 * g + q) & 6;                     // SW: swizzle key of rows 32 s2 + 4 g + q and + 16
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        if (s2 == 1 && nf <= 2) break;                  // register items 32 .. 63 are all empty
        bf16x8 b;
#pragma unroll
        for (int r = 0; r < 4; ++r) { b[r] = (bf16_t)w[2 * s2][r]; b[4 + r] = (bf16_t)w[2 * s2 + 1][r]; }
        const char* row_lo = SW ? rowmaj + (32 * s2 + 4 * g + q) * 128 + 8 * (pp & 1) : rowmaj + (32 * s2 + 4 * g + q) * Geo<KS>::PR + 8 * pp;
        const char* row_hi = row_lo + 16 * (SW ? 128 : Geo<KS>::PR);
#pragma unroll
        for (int dm = 0; dm < D16; ++dm) {
            const int o = SW ? (((2 * dm + (pp >> 1)) ^ sw) << 4) : 32 * dm;
            const bf16x4 lo = lds_tr4(row_lo + o);
            const bf16x4 hi = lds_tr4(row_hi + o);
            bf16x8 a;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = lo[r]; a[4 + r] = hi[r]; }
            out[dm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, out[dm], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ float group4_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float group4_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }

// store 4 consecutive d (16 dm + 4 g .. +3) of one row
__device__ __forceinline__ void store4(bf16_t* dst, int64_t ld, int row, int head_off, int d0, int hd, const f32x4& v, float mul) {
    if (d0 < hd) elem<bf16_t>::st4(dst + (int64_t)row * ld + head_off + d0, make_float4(v[0] * mul, v[1] * mul, v[2] * mul, v[3] * mul));
}

// shared prologue: resolve the sample's row window (packed or batched); false = nothing to do for this block
__device__ __forceinline__ bool window(Params& p, int b, int first_item, bool lanes_are_queries) {
    if (p.cu) {
        const int base = p.cu[b], len = p.cu[b + 1] - base;
        if (first_item >= len) return false;
        p.Sq = p.Sk = len;
        p.bsq = p.bsk = base;        // reused as absolute row offsets below
    } else {
        if (first_item >= (lanes_are_queries ? p.Sq : p.Sk)) return false;
        p.bsq = (int64_t)b * p.bsq;
        p.bsk = (int64_t)b * p.bsk;
    }
    return true;
}

// ------------------------------------------------------------------------------------ forward
// NW waves = 16 NW queries per workgroup: every K / V tile a workgroup stages serves 16 NW queries, so a sequence of 257 tokens
// (ViT-L/14) is 3 workgroups of 6 waves (18 wave slots for 17 used) instead of 5 of 4 (the fifth for ONE query row), 150 prompt
// positions 2 of 5 instead of 3 of 4: fewer staging passes over K / V for the same products (run() picks NW per problem).
// FAST = false: the instantiation for problems of fewer than 64 keys (the training forward at S = 42: one ragged tile) - no tile of theirs can
// take the interior-tile paths, and carrying them cost that kernel 0.6 us per launch
template <int KS, int D16, int NW, bool FAST = true>
__global__ __launch_bounds__(64 * NW) void fwd_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + Geo<KS>::ROW_BYTES;
    int* valid = reinterpret_cast<int*>(Vs + Geo<KS>::ROW_BYTES);
    constexpr int QT = 16 * NW;
    constexpr bool SW = KS == 2;                             // hd = 64: swizzled 128-byte rows (tile_dot)

    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int q0 = blockIdx.x * QT;
    if (!window(p, b, q0, true)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, g = lane >> 4;
    const int qi = q0 + wave * 16 + x;
    const bool active = qi < p.Sq;
    const bool wave_has_query = q0 + wave * 16 < p.Sq;       // a wave without queries only helps staging the tiles
    const int off = p.Sk - p.Sq, head_off = h * p.hd;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + p.bsq * p.ldq;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + p.bsk * p.ldk;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + p.bsk * p.ldv;

    bf16x8 qf[KS];
    load_bfrag<KS>(qf, Q, p.ldq, qi, active, p.hd, head_off, g);
    f32x4 acc[D16];
#pragma unroll
    for (int dm = 0; dm < D16; ++dm) acc[dm] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -FLT_MAX, lsum = 0.f;

    int k_end = p.Sk;
    if (p.causal) k_end = min(p.Sk, min(p.Sq, q0 + QT) + off);
    if (k_end < 1) k_end = min(p.Sk, 1);
    // K / V tiles travel global -> registers -> LDS; the loads of tile t + 1 are issued before the products of tile t and written to
    // LDS after them, so their latency runs under the MFMA / softmax work instead of in front of it
    constexpr int CH = Geo<KS>::HP / 8, NT = 64 * NW, NCH = (TILE * CH + NT - 1) / NT;
    uint4 kreg[NCH], vreg[NCH];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = threadIdx.x + i * NT, r = c / CH, ch = c - r * CH;
            kreg[i] = make_uint4(0u, 0u, 0u, 0u);
            vreg[i] = kreg[i];
            if (c < TILE * CH && k0 + r < p.Sk && ch * 8 < p.hd) {
                kreg[i] = *reinterpret_cast<const uint4*>(K + (int64_t)(k0 + r) * p.ldk + head_off + ch * 8);
                vreg[i] = *reinterpret_cast<const uint4*>(V + (int64_t)(k0 + r) * p.ldv + head_off + ch * 8);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = threadIdx.x + i * NT, r = c / CH, ch = c - r * CH;
            if (c < TILE * CH) {
                const int o = SW ? r * 128 + ((ch ^ (r & 6)) << 4) : r * Geo<KS>::PR + ch * 16;
                *reinterpret_cast<uint4*>(Ks + o) = kreg[i];
                *reinterpret_cast<uint4*>(Vs + o) = vreg[i];
            }
        }
    };
    gload(0);
    for (int k0 = 0; k0 < k_end; k0 += TILE) {
        __syncthreads();                                               // the previous tile is consumed
        lstore();
        for (int c = threadIdx.x; c < TILE; c += NT)
            valid[c] = (k0 + c < p.Sk) && (!p.key_mask || p.key_mask[(int64_t)b * p.ld_mask + k0 + c] != 0);
        __syncthreads();
        // every key of the tile exists and is attended: one LDS read and a ballot per wave (TILE = 64 = the wave; a __syncthreads_and in place of
        // the barrier cost the one-tile training forward 1.7 us per launch)
        const bool tile_all_keys = FAST && __builtin_amdgcn_ballot_w64(valid[lane] != 0) == ~0ull;
        if (k0 + TILE < k_end) gload(k0 + TILE);
        const int nf = min(4, (min(TILE, p.Sk - k0) + 15) >> 4);      // 16-key fragments of this tile that hold a key
        if (wave_has_query) {
        f32x4 st[4];
        tile_dot<KS, SW>(st, Ks, qf, x, g, nf);
        // a full tile without a mask (every tile but the last of the CLIP tower's 257 tokens): no per-key tests, and the scale moves
        // into the exponent - exp((s - m) scale) = exp2(s c - m c'), two instructions per score instead of ten
        // Round 4: "plain" also covers the interior tiles of masked / causal problems - every key attended (tile_all_keys) and, under a causal
        // mask, the whole tile at or before this wave's first query (wave-uniform): the few-shot prefill's tiles below the diagonal, every
        // full tile of the T5 encoder.  With T5's relative-position bias such a tile takes the third path below (bias added, exponent fused).
        const bool clean_tile = FAST && tile_all_keys && (!p.causal || k0 + TILE - 1 <= q0 + wave * 16 + off) && p.scale > 0.f;
        const bool plain_tile = clean_tile && !p.rel_bias;
        const bool bias_tile = clean_tile && p.rel_bias;
        float m_new;
        if (bias_tile) {
            const float* rel = p.rel_bias + (int64_t)h * p.rel_ld;
            const int rel_shift = p.rel_zero - (qi + off) + k0, rel_last = (int)p.rel_ld - 1;
            float tmax = -FLT_MAX;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[f][r] = fmaf(st[f][r], p.scale, rel[min(max(16 * f + 4 * g + r + rel_shift, 0), rel_last)]);
                    tmax = fmaxf(tmax, st[f][r]);
                }
            m_new = fmaxf(m, group4_max(tmax));
        } else if (plain_tile) {
            float tmax = -FLT_MAX;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, st[f][r]);
            m_new = fmaxf(m, group4_max(tmax) * p.scale);
        } else {
            // the bias table of this head, shifted so that the key position indexes it (entries outside the table: clamped, never visible)
            const float* rel = p.rel_bias ? p.rel_bias + (int64_t)h * p.rel_ld : nullptr;
            const int rel_shift = p.rel_zero - (qi + off), rel_last = (int)p.rel_ld - 1;
            float tmax = -FLT_MAX;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = 16 * f + 4 * g + r;          // key within the tile
                    const bool exists = k0 + kk < p.Sk;
                    const bool vis = valid[kk] && (!p.causal || (k0 + kk) <= qi + off);
                    float sv = st[f][r] * p.scale;
                    if (rel) sv += rel[min(max(k0 + kk + rel_shift, 0), rel_last)];
                    st[f][r] = exists ? (vis ? sv : -FLT_MAX) : -INFINITY;
                    tmax = fmaxf(tmax, st[f][r]);
                }
            m_new = fmaxf(m, group4_max(tmax));
        }
        const float corr = __expf(m - m_new);
        lsum *= corr;
#pragma unroll
        for (int dm = 0; dm < D16; ++dm) { acc[dm][0] *= corr; acc[dm][1] *= corr; acc[dm][2] *= corr; acc[dm][3] *= corr; }
        if (bias_tile) {
            const float mm = m_new * 1.4426950408889634f;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pj = __builtin_amdgcn_exp2f(fmaf(st[f][r], 1.4426950408889634f, -mm));
                    st[f][r] = pj;
                    lsum += pj;
                }
        } else if (plain_tile) {
            const float c2 = p.scale * 1.4426950408889634f, mm = m_new * 1.4426950408889634f;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pj = __builtin_amdgcn_exp2f(fmaf(st[f][r], c2, -mm));
                    st[f][r] = pj;
                    lsum += pj;
                }
        } else {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pj = __expf(st[f][r] - m_new);      // exp(-inf) = 0 for keys that do not exist
                    st[f][r] = pj;
                    lsum += pj;
                }
        }
        tile_accumulate<KS, D16, SW>(acc, Vs, st, x, g, nf);
        m = m_new;
        }
    }
    const float l = group4_sum(lsum);
    if (active) {
        const float inv = 1.f / l;
        bf16_t* O = reinterpret_cast<bf16_t*>(p.out) + p.bsq * p.ldo;
#pragma unroll
        for (int dm = 0; dm < D16; ++dm) store4(O, p.ldo, qi, head_off, 16 * dm + 4 * g, p.hd, acc[dm], inv);
        if (p.lse && g == 0) p.lse[((int64_t)b * p.H + h) * p.stat_ld + qi] = m + __logf(l);
    }
}

// ------------------------------------------------------------------------------------ forward, K / V resident in LDS (hd = 64)
// The CLIP tower's attention (ViT-L/14: 257 tokens x 16 heads x 160 images per few-shot batch, no mask) was 3 x HBM-bound in the
// streamed kernel above: each (image, head) problem was cut into three workgroups of 96 queries that each staged all of K and V
// again (on three different XCDs: consecutive blocks land on different L2s), through registers, with two barriers per 64-key
// tile.  Here ONE workgroup owns the whole (image, head) problem:
//   * all of K and V (N x 64 bf16 each, 36 KiB at N = 257, 74 KiB at N = 577) arrive ONCE by LDS-DMA (1-KiB pieces of eight
//     128-byte rows, no registers) into two row-major images that stay for the lifetime of the workgroup: one barrier in all;
//   * image rows are 128 bytes; 16-byte chunk c of row r sits at slot c ^ (r & 6) (applied on the DMA's per-lane SOURCE address,
//     the same involution on every read).  That one swizzle makes BOTH kinds of read conflict-free: the ds_read_b128 row reads of
//     K (16-lane groups hold rows {0-3, 12-15} at chunk a and rows {4-11} at chunk a ^ 1: r & 6 sends the 8 rows of either
//     parity to 8 different slots) and the ds_read_b64_tr_b16 transposed reads of V (a 32-lane half reads 8 consecutive rows x 2
//     adjacent chunks: bit 0 of the chunk index is untouched, so the pair stays adjacent, and r & 6 spreads the rows);
//   * a wave owns 32 queries (two B fragments), so every K / V^T fragment read from LDS feeds two MFMAs: half the LDS traffic per
//     FLOP of the 16-query waves above (whose 4 LDS cycles per 16-cycle MFMA on four SIMDs are exactly the LDS's whole rate);
//   * waves walk the query blocks independently (no barrier after the staging one).
// Rows beyond N in the images are copies of row N - 1 (clamped DMA source: finite values); their scores are masked to -inf.
__device__ __forceinline__ int res_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 6)) << 4); }

__global__ __launch_bounds__(1024, 4) void fwd_resident64_kernel(Params p, int npad, int nqb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + npad * 128;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x & 63, x = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int N = p.Sk, head_off = h * 64;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.bsq * p.ldq;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.bsk * p.ldk;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.bsk * p.ldv;
    {
        // piece i < npieces: K rows 8 i .. 8 i + 7; piece npieces + i: the same rows of V.  Lane l of a piece: row 8 i + (l >> 3), slot l & 7.
        const int npieces = npad >> 3, r8 = lane >> 3, slot = lane & 7;
        for (int i = wave; i < 2 * npieces; i += nw) {
            const bool isv = i >= npieces;
            const int piece = isv ? i - npieces : i;
            const int row = piece * 8 + r8;
            const bf16_t* src = (isv ? V + (int64_t)min(row, N - 1) * p.ldv : K + (int64_t)min(row, N - 1) * p.ldk) + head_off + ((slot ^ (row & 6)) << 3);
            char* dst = (isv ? Vs : Ks) + piece * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    }
    __builtin_amdgcn_s_waitcnt(vm_only_attn(0));
    __syncthreads();

    const float c2 = p.scale * 1.4426950408889634f;
    const int tq = x >> 2, pp = x & 3;
    for (int qb = wave; qb < nqb; qb += nw) {
        const int qa = qb * 32 + x, qc = qa + 16;
        bf16x8 qfa[2], qfb[2];
        load_bfrag<2>(qfa, Q, p.ldq, qa, qa < N, 64, head_off, g);
        load_bfrag<2>(qfb, Q, p.ldq, qc, qc < N, 64, head_off, g);
        f32x4 oa[4], ob[4];
#pragma unroll
        for (int dm = 0; dm < 4; ++dm) { oa[dm] = (f32x4){0.f, 0.f, 0.f, 0.f}; ob[dm] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        float ma = -FLT_MAX, mb = -FLT_MAX, la = 0.f, lb = 0.f;          // running max in SCALED units (score * scale)
        for (int k0 = 0; k0 < N; k0 += TILE) {
            const int nf = min(4, (N - k0 + 15) >> 4);                    // 16-key fragments of this tile that hold a key
            f32x4 sa[4], sb[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                sa[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
                sb[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (f < nf) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ks + res_off(k0 + 16 * f + x, 4 * s + g));
                        sa[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qfa[s], sa[f], 0, 0, 0);
                        sb[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qfb[s], sb[f], 0, 0, 0);
                    }
                }
                if (f == 1) __builtin_amdgcn_sched_barrier(0);            // at most four K fragments in flight (registers: 128 per lane)
            }
            if (k0 + TILE > N) {                                          // last tile: keys that do not exist
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + 16 * f + 4 * g + r >= N) { sa[f][r] = -INFINITY; sb[f][r] = -INFINITY; }
            }
            float ta = -FLT_MAX, tb = -FLT_MAX;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) { ta = fmaxf(ta, sa[f][r]); tb = fmaxf(tb, sb[f][r]); }
            // exp((s - m) scale) = exp2(s c2 - m'), m' = the running maximum in log2 units (scale > 0: max commutes with it)
            const float na = fmaxf(ma, group4_max(ta) * c2), nb = fmaxf(mb, group4_max(tb) * c2);
            const float ca = __builtin_amdgcn_exp2f(ma - na), cb = __builtin_amdgcn_exp2f(mb - nb);
            la *= ca; lb *= cb;
#pragma unroll
            for (int dm = 0; dm < 4; ++dm) {
                oa[dm][0] *= ca; oa[dm][1] *= ca; oa[dm][2] *= ca; oa[dm][3] *= ca;
                ob[dm][0] *= cb; ob[dm][1] *= cb; ob[dm][2] *= cb; ob[dm][3] *= cb;
            }
            ma = na; mb = nb;
            // per half of the tile (32 keys): P^T = exp2(...) as bf16 B fragments (k order of the accumulator-as-B trick: registers 0-3 of two
            // adjacent 16-key tiles), then O^T += V^T . P^T with the A operand (rows = d, k = keys in that permuted order) from two
            // transposing reads of the row-major V image
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (s2 == 1 && nf <= 2) break;                            // keys 32 .. 63 of the tile do not exist
                __builtin_amdgcn_sched_barrier(0);                        // keeps the second half's reads out of the first half (registers)
                bf16x8 pa8, pb8;
#pragma unroll
                for (int ff = 0; ff < 2; ++ff)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pa = __builtin_amdgcn_exp2f(fmaf(sa[2 * s2 + ff][r], c2, -na)), pb = __builtin_amdgcn_exp2f(fmaf(sb[2 * s2 + ff][r], c2, -nb));
                        la += pa; lb += pb;
                        pa8[4 * ff + r] = (bf16_t)pa;
                        pb8[4 * ff + r] = (bf16_t)pb;
                    }
                const int row_lo = k0 + 32 * s2 + 4 * g + tq;            // (row_lo + 16) & 6 == row_lo & 6
                const char* base = Vs + row_lo * 128 + 8 * (pp & 1);
                const int sw = row_lo & 6, half = pp >> 1;
#pragma unroll
                for (int dm = 0; dm < 4; ++dm) {
                    const int off = ((2 * dm + half) ^ sw) << 4;
                    const bf16x4 lo = lds_tr4(base + off);
                    const bf16x4 hi = lds_tr4(base + 16 * 128 + off);
                    bf16x8 a;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a[r] = lo[r]; a[4 + r] = hi[r]; }
                    oa[dm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pa8, oa[dm], 0, 0, 0);
                    ob[dm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb8, ob[dm], 0, 0, 0);
                }
            }
        }
        const float lta = group4_sum(la), ltb = group4_sum(lb);
        bf16_t* O = reinterpret_cast<bf16_t*>(p.out) + (int64_t)b * p.bsq * p.ldo;
        if (qa < N) {
            const float inv = 1.f / lta;
#pragma unroll
            for (int dm = 0; dm < 4; ++dm) store4(O, p.ldo, qa, head_off, 16 * dm + 4 * g, 64, oa[dm], inv);
            if (p.lse && g == 0) p.lse[((int64_t)b * p.H + h) * p.stat_ld + qa] = ma * 0.6931471805599453f + __logf(lta);
        }
        if (qc < N) {
            const float inv = 1.f / ltb;
#pragma unroll
            for (int dm = 0; dm < 4; ++dm) store4(O, p.ldo, qc, head_off, 16 * dm + 4 * g, 64, ob[dm], inv);
            if (p.lse && g == 0) p.lse[((int64_t)b * p.H + h) * p.stat_ld + qc] = mb * 0.6931471805599453f + __logf(ltb);
        }
    }
}

// the resident kernel's domain: self-attention (Sq == Sk) over at most 592 positions with hd = 64, no mask, no packing
bool resident_supported(const Params& p) {
    return p.hd == 64 && !p.causal && !p.key_mask && !p.cu && p.Sq == p.Sk && p.Sk <= 592 && p.scale > 0.f &&
           !(p.ldq % 8 || p.ldk % 8 || p.ldv % 8 || p.ldo % 4) && eavqa_aligned16(p.q) && eavqa_aligned16(p.k) && eavqa_aligned16(p.v);
}

int run_resident(const Params& p, hipStream_t s) {
    const int N = p.Sk, npad = (N + 31) / 32 * 32, nqb = (N + 31) / 32;
    const size_t lds = (size_t)2 * npad * 128;
    // waves per workgroup: 128 VGPRs allow 16 waves per CU; images of <= 80 KiB let two workgroups share a CU (one stages while the
    // other multiplies), so those take at most 8 waves each; wave w walks query blocks w, w + nw, ...
    const int max_nw = lds <= 80 * 1024 ? 8 : 16;
    const int rounds = (nqb + max_nw - 1) / max_nw;
    const int nw = rounds == 1 ? nqb : ((nqb - 1) % max_nw == 0 ? max_nw : (nqb + rounds - 1) / rounds);   // a lone last block rides on wave 0
    static std::atomic<bool> configured{false};
    if (lds > 64 * 1024 && !configured.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_resident64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
            return EAVQA_E_LAUNCH;
        configured.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(fwd_resident64_kernel, dim3(p.B * p.H), dim3(64 * nw), lds, s, p, npad, nqb);
    if (hipGetLastError() != hipSuccess) return EAVQA_E_LAUNCH;
    return EAVQA_OK;
}

// ------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <int KS, int D16>
__global__ __launch_bounds__(256) void bwd_dq_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = Ks + Geo<KS>::ROW_BYTES;
    int* valid = reinterpret_cast<int*>(Vs + Geo<KS>::ROW_BYTES);

    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int q0 = blockIdx.x * TILE;
    if (!window(p, b, q0, true)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, g = lane >> 4;
    const int qi = q0 + wave * 16 + x;
    const bool active = qi < p.Sq;
    const int off = p.Sk - p.Sq, head_off = h * p.hd;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + p.bsq * p.ldq;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + p.bsk * p.ldk;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + p.bsk * p.ldv;
    const bf16_t* O = reinterpret_cast<const bf16_t*>(p.o) + p.bsq * p.ldo;
    const bf16_t* DO = reinterpret_cast<const bf16_t*>(p.d_o) + p.bsq * p.lddo;

    bf16x8 qf[KS], dof[KS], of[KS];
    load_bfrag<KS>(qf, Q, p.ldq, qi, active, p.hd, head_off, g);
    load_bfrag<KS>(dof, DO, p.lddo, qi, active, p.hd, head_off, g);
    load_bfrag<KS>(of, O, p.ldo, qi, active, p.hd, head_off, g);
    float dsum = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) dsum += (float)dof[s][j] * (float)of[s][j];
    const float delta = group4_sum(dsum);
    const int64_t stat = ((int64_t)b * p.H + h) * p.stat_ld + qi;
    const float lse = active ? p.lse[stat] : 0.f;
    if (active && g == 0) p.delta[stat] = delta;

    f32x4 dq[D16];
#pragma unroll
    for (int dm = 0; dm < D16; ++dm) dq[dm] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int k_end = p.Sk;
    if (p.causal) k_end = min(p.Sk, min(p.Sq, q0 + TILE) + off);
    if (k_end < 1) k_end = min(p.Sk, 1);
    for (int k0 = 0; k0 < k_end; k0 += TILE) {
        __syncthreads();
        stage<KS>(Ks, nullptr, K, p.ldk, k0, p.Sk, p.hd, head_off);
        stage<KS>(Vs, nullptr, V, p.ldv, k0, p.Sk, p.hd, head_off);
        for (int c = threadIdx.x; c < TILE; c += 256)
            valid[c] = (k0 + c < p.Sk) && (!p.key_mask || p.key_mask[(int64_t)b * p.ld_mask + k0 + c] != 0);
        __syncthreads();
        f32x4 st[4], dp[4];
        tile_dot<KS>(st, Ks, qf, x, g);
        tile_dot<KS>(dp, Vs, dof, x, g);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 16 * f + 4 * g + r;
                const bool exists = k0 + kk < p.Sk;
                const bool vis = exists && valid[kk] && (!p.causal || (k0 + kk) <= qi + off);
                const float pj = exists ? __expf((vis ? st[f][r] * p.scale + rel_bias_at(p, h, k0 + kk, qi + off) : -FLT_MAX) - lse) : 0.f;
                st[f][r] = pj * (dp[f][r] - delta) * p.scale;      // dS^T
            }
        tile_accumulate<KS, D16>(dq, Ks, st, x, g);
    }
    if (active) {
        bf16_t* DQ = reinterpret_cast<bf16_t*>(p.dq) + p.bsq * p.lddq;
#pragma unroll
        for (int dm = 0; dm < D16; ++dm) store4(DQ, p.lddq, qi, head_off, 16 * dm + 4 * g, p.hd, dq[dm], 1.f);
    }
}

// ------------------------------------------------------------------------------------ backward: dK, dV
template <int KS, int D16>
__global__ __launch_bounds__(256) void bwd_dkv_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* DOs = Qs + Geo<KS>::ROW_BYTES;
    float* stats = reinterpret_cast<float*>(DOs + Geo<KS>::ROW_BYTES);    // lse[64], delta[64]

    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int j0 = blockIdx.x * TILE;
    if (!window(p, b, j0, false)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, g = lane >> 4;
    const int kj = j0 + wave * 16 + x;                 // this lane's key
    const bool active = kj < p.Sk;
    const int off = p.Sk - p.Sq, head_off = h * p.hd;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + p.bsq * p.ldq;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + p.bsk * p.ldk;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + p.bsk * p.ldv;
    const bf16_t* DO = reinterpret_cast<const bf16_t*>(p.d_o) + p.bsq * p.lddo;

    bf16x8 kf[KS], vf[KS];
    load_bfrag<KS>(kf, K, p.ldk, kj, active, p.hd, head_off, g);
    load_bfrag<KS>(vf, V, p.ldv, kj, active, p.hd, head_off, g);
    const bool kvalid = active && (!p.key_mask || p.key_mask[(int64_t)b * p.ld_mask + kj] != 0);
    f32x4 dk[D16], dv[D16];
#pragma unroll
    for (int dm = 0; dm < D16; ++dm) { dk[dm] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dm] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    int q_begin = 0;
    if (p.causal) q_begin = (max(0, j0 - off) / TILE) * TILE;
    for (int q0 = q_begin; q0 < p.Sq; q0 += TILE) {
        __syncthreads();
        stage<KS>(Qs, nullptr, Q, p.ldq, q0, p.Sq, p.hd, head_off);
        stage<KS>(DOs, nullptr, DO, p.lddo, q0, p.Sq, p.hd, head_off);
        for (int c = threadIdx.x; c < TILE; c += 256) {
            const bool in = q0 + c < p.Sq;
            const int64_t st = ((int64_t)b * p.H + h) * p.stat_ld + q0 + c;
            stats[c] = in ? p.lse[st] : 0.f;
            stats[TILE + c] = in ? p.delta[st] : 0.f;
        }
        __syncthreads();
        f32x4 sc[4], dp[4];
        tile_dot<KS>(sc, Qs, kf, x, g);
        tile_dot<KS>(dp, DOs, vf, x, g);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = 16 * f + 4 * g + r;          // query within the tile
                const bool exists = q0 + qq < p.Sq;
                const bool vis = exists && kvalid && (!p.causal || kj <= (q0 + qq) + off);
                const float pj = exists ? __expf((vis ? sc[f][r] * p.scale + rel_bias_at(p, h, kj, q0 + qq + off) : -FLT_MAX) - stats[qq]) : 0.f;
                sc[f][r] = pj;                                                   // P
                dp[f][r] = pj * (dp[f][r] - stats[TILE + qq]) * p.scale;        // dS
            }
        tile_accumulate<KS, D16>(dv, DOs, sc, x, g);
        tile_accumulate<KS, D16>(dk, Qs, dp, x, g);
    }
    if (active) {
        bf16_t* DK = reinterpret_cast<bf16_t*>(p.dk) + p.bsk * p.lddk;
        bf16_t* DV = reinterpret_cast<bf16_t*>(p.dv) + p.bsk * p.lddv;
#pragma unroll
        for (int dm = 0; dm < D16; ++dm) {
            store4(DK, p.lddk, kj, head_off, 16 * dm + 4 * g, p.hd, dk[dm], 1.f);
            store4(DV, p.lddv, kj, head_off, 16 * dm + 4 * g, p.hd, dv[dm], 1.f);
        }
    }
}

// ------------------------------------------------------------------------------------ backward, one tile: dQ, dK, dV
// Sq, Sk <= 64 (every training sequence of the hot path: prefix + caption <= 42 positions): the whole (batch, head)
// problem is one tile, so one workgroup produces all three gradients and S / P / dS are computed once instead of twice.
// Orientation of bwd_dkv_kernel (lane item = key) for P, dS, dV, dK; dS is then handed over through LDS as a row-major
// [query][key] bf16 image and dQ^T += K^T . dS^T runs with lane item = query (A = K^T in
// natural k order, B = 8 consecutive keys of the lane's query row).  All A operands that sum over sequence items come
// from row-major images through transposing LDS reads.
template <int KS, int D16>
__global__ __launch_bounds__(256) void bwd_fused_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* DOs = Qs + Geo<KS>::ROW_BYTES;
    char* Ks = DOs + Geo<KS>::ROW_BYTES;
    char* DSs = Ks + Geo<KS>::ROW_BYTES;                                  // [64 queries][64 keys] bf16, pitch DS_PITCH
    constexpr int DS_PITCH = TILE * 2 + 16;
    float* stats = reinterpret_cast<float*>(DSs + TILE * DS_PITCH);       // lse[64], delta[64]

    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    if (!window(p, b, 0, false)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, g = lane >> 4;
    const int item = wave * 16 + x;                   // this lane's key (first half) and query (second half)
    const int off = p.Sk - p.Sq, head_off = h * p.hd;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + p.bsq * p.ldq;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + p.bsk * p.ldk;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + p.bsk * p.ldv;
    const bf16_t* O = reinterpret_cast<const bf16_t*>(p.o) + p.bsq * p.ldo;
    const bf16_t* DO = reinterpret_cast<const bf16_t*>(p.d_o) + p.bsq * p.lddo;

    stage<KS>(Qs, nullptr, Q, p.ldq, 0, p.Sq, p.hd, head_off);
    stage<KS>(DOs, nullptr, DO, p.lddo, 0, p.Sq, p.hd, head_off);
    stage<KS>(Ks, nullptr, K, p.ldk, 0, p.Sk, p.hd, head_off);
    // every tile is fetched from memory once: Q, dO, K as LDS images (whose rows also serve as this lane's fragments), V and
    // O only as fragments
    const bool qa = item < p.Sq, kactive = item < p.Sk;
    bf16x8 vf[KS], of[KS];
    load_bfrag<KS>(vf, V, p.ldv, item, kactive, p.hd, head_off, g);
    load_bfrag<KS>(of, O, p.ldo, item, qa, p.hd, head_off, g);
    const int64_t stat_at = ((int64_t)b * p.H + h) * p.stat_ld + item;
    if (g == 0) stats[item] = qa ? p.lse[stat_at] : 0.f;
    const bool kvalid = kactive && (!p.key_mask || p.key_mask[(int64_t)b * p.ld_mask + item] != 0);
    __syncthreads();
    bf16x8 kf[KS];
    {
        float dsum = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kf[s] = *reinterpret_cast<const bf16x8*>(Ks + item * Geo<KS>::PR + (32 * s + 8 * g) * 2);
            const bf16x8 dof = *reinterpret_cast<const bf16x8*>(DOs + item * Geo<KS>::PR + (32 * s + 8 * g) * 2);
#pragma unroll
            for (int j = 0; j < 8; ++j) dsum += (float)dof[j] * (float)of[s][j];
        }
        const float delta = group4_sum(dsum);        // rowsum(dO * O) of query `item`
        if (g == 0) {
            stats[TILE + item] = qa ? delta : 0.f;
            if (qa) p.delta[stat_at] = delta;
        }
    }
    __syncthreads();

    f32x4 sc[4], dp[4];
    tile_dot<KS>(sc, Qs, kf, x, g);
    tile_dot<KS>(dp, DOs, vf, x, g);
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = 16 * f + 4 * g + r;
            const bool exists = qq < p.Sq;
            const bool vis = exists && kvalid && (!p.causal || item <= qq + off);
            const float pj = exists ? __expf((vis ? sc[f][r] * p.scale + rel_bias_at(p, h, item, qq + off) : -FLT_MAX) - stats[qq]) : 0.f;
            sc[f][r] = pj;                                                   // P
            dp[f][r] = pj * (dp[f][r] - stats[TILE + qq]) * p.scale;        // dS
            *reinterpret_cast<bf16_t*>(DSs + qq * DS_PITCH + item * 2) = (bf16_t)(kactive ? dp[f][r] : 0.f);
        }
    {
        f32x4 dk[D16], dv[D16];
#pragma unroll
        for (int dm = 0; dm < D16; ++dm) { dk[dm] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dm] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        tile_accumulate<KS, D16>(dv, DOs, sc, x, g);
        tile_accumulate<KS, D16>(dk, Qs, dp, x, g);
        if (kactive) {
            bf16_t* DK = reinterpret_cast<bf16_t*>(p.dk) + p.bsk * p.lddk;
            bf16_t* DV = reinterpret_cast<bf16_t*>(p.dv) + p.bsk * p.lddv;
#pragma unroll
            for (int dm = 0; dm < D16; ++dm) {
                store4(DK, p.lddk, item, head_off, 16 * dm + 4 * g, p.hd, dk[dm], 1.f);
                store4(DV, p.lddv, item, head_off, 16 * dm + 4 * g, p.hd, dv[dm], 1.f);
            }
        }
    }
    __syncthreads();                                  // the dS image is complete

    f32x4 dq[D16];
#pragma unroll
    for (int dm = 0; dm < D16; ++dm) dq[dm] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const int q = x >> 2, pp = x & 3;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(DSs + item * DS_PITCH + (32 * s2 + 8 * g) * 2);
            const char* row_lo = Ks + (32 * s2 + 8 * g + q) * Geo<KS>::PR + 8 * pp;      // keys 32 s2 + 8 g .. +3
            const char* row_hi = row_lo + 4 * Geo<KS>::PR;                                 // keys .. +4 .. +7
#pragma unroll
            for (int dm = 0; dm < D16; ++dm) {
                const bf16x4 lo = lds_tr4(row_lo + 32 * dm);
                const bf16x4 hi = lds_tr4(row_hi + 32 * dm);
                bf16x8 a;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a[r] = lo[r]; a[4 + r] = hi[r]; }
                dq[dm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfrag, dq[dm], 0, 0, 0);
            }
        }
    }
    if (item < p.Sq) {
        bf16_t* DQ = reinterpret_cast<bf16_t*>(p.dq) + p.bsq * p.lddq;
#pragma unroll
        for (int dm = 0; dm < D16; ++dm) store4(DQ, p.lddq, item, head_off, 16 * dm + 4 * g, p.hd, dq[dm], 1.f);
    }
}

// ------------------------------------------------------------------------------------ backward, one tile, hd = 64: swizzled images
// The training step's attention backward (GPT-2-large / OPT-1.3B heads, packed sequences of 18 .. 42 positions): the kernel above
// with (a) 128-byte image rows and the chunk swizzle of the resident forward instead of the 144-byte padded pitch (round 2 measured
// 69 % of its LDS cycles as bank conflicts): every ds_read_b128 row read and every ds_read_b64_tr_b16 transposed read is
// conflict-free, the dS image [query][64 keys] uses chunk ^ ((row >> 1) & 7) for its 8-byte row reads; (b) images of R = 16
// ceil(S / 16) rows only (S = the longest sequence of the launch): 25 KiB at S <= 48 instead of 37 KiB, so five to six workgroups
// share a CU and the 1 280 problems of a cfg2 step are one round instead of a round and a quarter; (c) only the 16-item fragments that
// hold items are multiplied.  The dS B fragment of the dQ product is read in the permuted k order of the accumulator-as-B trick, so
// K^T comes out of the standard transposed-read pattern too.
__device__ __forceinline__ int ds_off(int row, int key) { return row * 128 + ((((key >> 3) ^ (row >> 1)) & 7) << 4) + (key & 7) * 2; }

__global__ __launch_bounds__(256, 5) void bwd_fused64_kernel(Params p, int R) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* DOs = Qs + R * 128;
    char* Ks = DOs + R * 128;
    char* DSs = Ks + R * 128;
    float* stats = reinterpret_cast<float*>(DSs + R * 128);              // lse[64], delta[64]

    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    if (!window(p, b, 0, false)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, g = lane >> 4;
    const int item = wave * 16 + x;                   // this lane's key (first half) and query (second half)
    const int off = p.Sk - p.Sq, head_off = h * 64;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + p.bsq * p.ldq;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + p.bsk * p.ldk;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + p.bsk * p.ldv;
    const bf16_t* O = reinterpret_cast<const bf16_t*>(p.o) + p.bsq * p.ldo;
    const bf16_t* DO = reinterpret_cast<const bf16_t*>(p.d_o) + p.bsq * p.lddo;
    const int nfq = (p.Sq + 15) >> 4, nfk = (p.Sk + 15) >> 4;            // 16-item fragments that hold a query / a key
    const bool wave_has_key = wave < nfk, wave_has_query = wave < nfq;

    // stage Q, dO, K: rows < R (rows beyond the sequence zero), chunk c of row r at slot c ^ (r & 6)
    for (int c = threadIdx.x; c < R * 8; c += 256) {
        const int r = c >> 3, ch = c & 7, so = res_off(r, ch);
        uint4 vq = make_uint4(0u, 0u, 0u, 0u), vd = vq, vk = vq;
        if (r < p.Sq) {
            vq = *reinterpret_cast<const uint4*>(Q + (int64_t)r * p.ldq + head_off + ch * 8);
            vd = *reinterpret_cast<const uint4*>(DO + (int64_t)r * p.lddo + head_off + ch * 8);
        }
        if (r < p.Sk) vk = *reinterpret_cast<const uint4*>(K + (int64_t)r * p.ldk + head_off + ch * 8);
        *reinterpret_cast<uint4*>(Qs + so) = vq;
        *reinterpret_cast<uint4*>(DOs + so) = vd;
        *reinterpret_cast<uint4*>(Ks + so) = vk;
    }
    const bool qa = item < p.Sq, kactive = item < p.Sk;
    bf16x8 vf[2], of[2];
    load_bfrag<2>(vf, V, p.ldv, item, kactive, 64, head_off, g);
    load_bfrag<2>(of, O, p.ldo, item, qa, 64, head_off, g);
    const int64_t stat_at = ((int64_t)b * p.H + h) * p.stat_ld + item;
    if (g == 0) stats[item] = qa ? p.lse[stat_at] : 0.f;
    const bool kvalid = kactive && (!p.key_mask || p.key_mask[(int64_t)b * p.ld_mask + item] != 0);
    __syncthreads();
    bf16x8 kf[2];
    {
        float dsum = 0.f;
        if (item < R) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                kf[s] = *reinterpret_cast<const bf16x8*>(Ks + res_off(item, 4 * s + g));
                const bf16x8 dof = *reinterpret_cast<const bf16x8*>(DOs + res_off(item, 4 * s + g));
#pragma unroll
                for (int j = 0; j < 8; ++j) dsum += (float)dof[j] * (float)of[s][j];
            }
        } else {
            kf[0] = zero8(); kf[1] = zero8();
        }
        const float delta = group4_sum(dsum);        // rowsum(dO * O) of query `item`
        if (g == 0) {
            stats[TILE + item] = qa ? delta : 0.f;
            if (qa) p.delta[stat_at] = delta;
        }
    }
    __syncthreads();

    const int tq = x >> 2, pp = x & 3;
    f32x4 sc[4], dp[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { sc[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    if (wave_has_key) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
            if (f < nfq) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 aq = *reinterpret_cast<const bf16x8*>(Qs + res_off(16 * f + x, 4 * s + g));
                    const bf16x8 ad = *reinterpret_cast<const bf16x8*>(DOs + res_off(16 * f + x, 4 * s + g));
                    sc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, kf[s], sc[f], 0, 0, 0);
                    dp[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad, vf[s], dp[f], 0, 0, 0);
                }
            }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = 16 * f + 4 * g + r;
            const bool exists = qq < p.Sq;
            const bool vis = exists && kvalid && (!p.causal || item <= qq + off);
            const float pj = exists ? __expf((vis ? sc[f][r] * p.scale + rel_bias_at(p, h, item, qq + off) : -FLT_MAX) - stats[qq]) : 0.f;
            sc[f][r] = pj;                                                   // P
            dp[f][r] = pj * (dp[f][r] - stats[TILE + qq]) * p.scale;        // dS
            if (qq < R && item < TILE) *reinterpret_cast<bf16_t*>(DSs + ds_off(qq, item)) = (bf16_t)(kactive ? dp[f][r] : 0.f);
        }
    if (wave_has_key) {
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dm = 0; dm < 4; ++dm) { dk[dm] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dm] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 == 1 && nfq <= 2) break;                                  // queries 32 .. 63 do not exist
            bf16x8 bp, bd;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bp[r] = (bf16_t)sc[2 * s2][r]; bp[4 + r] = (bf16_t)sc[2 * s2 + 1][r];
                bd[r] = (bf16_t)dp[2 * s2][r]; bd[4 + r] = (bf16_t)dp[2 * s2 + 1][r];
            }
            const int row_lo = 32 * s2 + 4 * g + tq, sw = row_lo & 6, half = pp >> 1;
            const int base = row_lo * 128 + 8 * (pp & 1);
            const bool hi_ok = nfq > 2 * s2 + 1;                             // rows row_lo + 16 lie inside the R-row images
#pragma unroll
            for (int dm = 0; dm < 4; ++dm) {
                const int o = base + (((2 * dm + half) ^ sw) << 4);
                const bf16x4 lo1 = lds_tr4(DOs + o), lo2 = lds_tr4(Qs + o);
                bf16x4 hi1 = lo1, hi2 = lo2;                                 // (multiplied by P = dS = 0 when the rows do not exist)
                if (hi_ok) { hi1 = lds_tr4(DOs + o + 16 * 128); hi2 = lds_tr4(Qs + o + 16 * 128); }
                bf16x8 a1, a2;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a1[r] = lo1[r]; a1[4 + r] = hi1[r]; a2[r] = lo2[r]; a2[4 + r] = hi2[r]; }
                dv[dm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bp, dv[dm], 0, 0, 0);
                dk[dm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bd, dk[dm], 0, 0, 0);
            }
        }
        if (kactive) {
            bf16_t* DK = reinterpret_cast<bf16_t*>(p.dk) + p.bsk * p.lddk;
            bf16_t* DV = reinterpret_cast<bf16_t*>(p.dv) + p.bsk * p.lddv;
#pragma unroll
            for (int dm = 0; dm < 4; ++dm) {
                store4(DK, p.lddk, item, head_off, 16 * dm + 4 * g, 64, dk[dm], 1.f);
                store4(DV, p.lddv, item, head_off, 16 * dm + 4 * g, 64, dv[dm], 1.f);
            }
        }
    }
    __syncthreads();                                  // the dS image is complete

    if (wave_has_query) {
        f32x4 dq[4];
#pragma unroll
        for (int dm = 0; dm < 4; ++dm) dq[dm] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 == 1 && nfk <= 2) break;                                  // keys 32 .. 63 do not exist
            // dS of query `item`, keys 32 s2 + 4 g .. +3 and 32 s2 + 16 + 4 g .. +3 (the permuted k order of the transposed reads below)
            const uint2 blo = *reinterpret_cast<const uint2*>(DSs + ds_off(item, 32 * s2 + 4 * g));
            const uint2 bhi = *reinterpret_cast<const uint2*>(DSs + ds_off(item, 32 * s2 + 16 + 4 * g));
            const uint4 bb = make_uint4(blo.x, blo.y, bhi.x, bhi.y);
            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, bb);
            const int row_lo = 32 * s2 + 4 * g + tq, sw = row_lo & 6, half = pp >> 1;
            const int base = row_lo * 128 + 8 * (pp & 1);
            const bool hi_ok = nfk > 2 * s2 + 1;
#pragma unroll
            for (int dm = 0; dm < 4; ++dm) {
                const int o = base + (((2 * dm + half) ^ sw) << 4);
                const bf16x4 lo = lds_tr4(Ks + o);
                bf16x4 hi = lo;
                if (hi_ok) hi = lds_tr4(Ks + o + 16 * 128);
                bf16x8 a;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a[r] = lo[r]; a[4 + r] = hi[r]; }
                dq[dm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfrag, dq[dm], 0, 0, 0);
            }
        }
        if (qa) {
            bf16_t* DQ = reinterpret_cast<bf16_t*>(p.dq) + p.bsq * p.lddq;
#pragma unroll
            for (int dm = 0; dm < 4; ++dm) store4(DQ, p.lddq, item, head_off, 16 * dm + 4 * g, 64, dq[dm], 1.f);
        }
    }
}

template <int KS, int D16>
int launch(int which, const Params& p, hipStream_t s) {
    const size_t row = Geo<KS>::ROW_BYTES;
    if (which == 0) {
        // waves per workgroup: the fewest wave slots for Sq queries, larger workgroups on ties (fewer passes over K / V).  Non-causal
        // only (the CLIP tower: 259.7 against 277.8 us per ViT-L layer at 160 images): with a causal mask the 64-query tiles skip more
        // key tiles than wider ones (few-shot prefill, 150 positions: 51.8 us with 4 waves, 74.2 us with 5)
        int nw = 4, best = ((p.Sq + 63) / 64) * 4;
        if (!p.causal)
            for (int cand : {5, 6, 8}) {
                const int slots = ((p.Sq + 16 * cand - 1) / (16 * cand)) * cand;
                if (slots <= best && p.Sq > 64) { best = slots; nw = cand; }
            }
        const dim3 grid((p.Sq + 16 * nw - 1) / (16 * nw), p.B * p.H);
        const size_t lds = 2 * row + TILE * 4;
        if (nw == 4 && p.Sk < TILE) hipLaunchKernelGGL((fwd_kernel<KS, D16, 4, false>), grid, dim3(256), lds, s, p);
        else if (nw == 4) hipLaunchKernelGGL((fwd_kernel<KS, D16, 4>), grid, dim3(256), lds, s, p);
        else if (nw == 5) hipLaunchKernelGGL((fwd_kernel<KS, D16, 5>), grid, dim3(320), lds, s, p);
        else if (nw == 6) hipLaunchKernelGGL((fwd_kernel<KS, D16, 6>), grid, dim3(384), lds, s, p);
        else hipLaunchKernelGGL((fwd_kernel<KS, D16, 8>), grid, dim3(512), lds, s, p);
    } else if (which == 1) {
        dim3 grid((p.Sq + TILE - 1) / TILE, p.B * p.H);
        hipLaunchKernelGGL((bwd_dq_kernel<KS, D16>), grid, dim3(256), 2 * row + TILE * 4, s, p);
    } else if (which == 3 && KS == 2 && !p.fused_padded) {
        const int R = ((max(p.Sq, p.Sk) + 15) / 16) * 16;
        hipLaunchKernelGGL(bwd_fused64_kernel, dim3(1, p.B * p.H), dim3(256), (size_t)4 * R * 128 + 2 * TILE * 4, s, p, R);
    } else if (which == 3) {
        static std::atomic<bool> configured{false};        // atomic: concurrent first calls only repeat an idempotent call
        const size_t lds = 3 * row + TILE * (TILE * 2 + 16) + 2 * TILE * 4;
        if (lds > 64 * 1024 && !configured.load(std::memory_order_acquire)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_fused_kernel<KS, D16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return EAVQA_E_LAUNCH;
            configured.store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL((bwd_fused_kernel<KS, D16>), dim3(1, p.B * p.H), dim3(256), lds, s, p);
    } else {
        static std::atomic<bool> configured{false};        // atomic: concurrent first calls only repeat an idempotent call        // hd = 128: 2 x 17 KiB + 2 x 17 KiB + stats > 64 KiB
        const size_t lds = 2 * row + 2 * TILE * 4;
        if (lds > 64 * 1024 && !configured.load(std::memory_order_acquire)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_dkv_kernel<KS, D16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return EAVQA_E_LAUNCH;
            configured.store(true, std::memory_order_release);
        }
        dim3 grid((p.Sk + TILE - 1) / TILE, p.B * p.H);
        hipLaunchKernelGGL((bwd_dkv_kernel<KS, D16>), grid, dim3(256), lds, s, p);
    }
    if (hipGetLastError() != hipSuccess) return EAVQA_E_LAUNCH;
    return EAVQA_OK;
}

bool supported(int hd) { return hd == 64 || hd == 80 || hd == 96 || hd == 128; }

// which: 0 forward, 1 dQ (+delta), 2 dK/dV, 3 all three gradients of a one-tile problem (Sq, Sk <= 64)
int run(int which, const Params& p, hipStream_t s) {
    switch (p.hd) {
        case 64: return launch<2, 4>(which, p, s);
        case 80: return launch<3, 5>(which, p, s);
        case 96: return launch<3, 6>(which, p, s);
        case 128: return launch<4, 8>(which, p, s);
        default: return EAVQA_E_SHAPE;
    }
}

// ------------------------------------------------------------------------------------ wide heads, one tile
// The transformer mapping network (reference: src/models/clip_cap.py Transformer / MultiHeadAttention: 8 heads over
// dim_embedding = the LM width) has head dims 160 (GPT-2-large) .. 512 (OPT-6.7B) over clip_length + prefix_length <= 64
// positions.  The whole (batch, head) problem is one 64 x 64 score tile, so the head dim is walked in CHUNKS of 128 with the
// geometry of the hd = 128 kernels: S (and dP) accumulate over the chunks, O / dQ / dK / dV are produced chunk by chunk, and no
// online-softmax rescaling exists because there is only one key tile.
constexpr int WKS = 4, WD16 = 8, WCH = 128;

template <int KS>
__device__ __forceinline__ void tile_dot_acc(f32x4 (&acc)[4], const char* rowmaj, const bf16x8 (&bf)[KS], int x, int g) {
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(rowmaj + (16 * f + x) * Geo<KS>::PR + (32 * s + 8 * g) * 2);
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[s], acc[f], 0, 0, 0);
        }
}

__global__ __launch_bounds__(256) void fwd_wide_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + Geo<WKS>::ROW_BYTES;
    int* valid = reinterpret_cast<int*>(Vs + Geo<WKS>::ROW_BYTES);

    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    if (!window(p, b, 0, true)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, g = lane >> 4;
    const int qi = wave * 16 + x;
    const bool active = qi < p.Sq;
    const int off = p.Sk - p.Sq, head_off = h * p.hd;
    const int nch = (p.hd + WCH - 1) / WCH;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + p.bsq * p.ldq;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + p.bsk * p.ldk;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + p.bsk * p.ldv;

    for (int c = threadIdx.x; c < TILE; c += 256)
        valid[c] = (c < p.Sk) && (!p.key_mask || p.key_mask[(int64_t)b * p.ld_mask + c] != 0);
    f32x4 st[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) st[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nch; ++c) {
        const int hc = min(WCH, p.hd - c * WCH), ho = head_off + c * WCH;
        char* buf = (c & 1) ? Vs : Ks;                       // alternate images: chunk c + 1 is staged while chunk c is consumed
        stage<WKS>(buf, nullptr, K, p.ldk, 0, p.Sk, hc, ho);
        bf16x8 qf[WKS];
        load_bfrag<WKS>(qf, Q, p.ldq, qi, active, hc, ho, g);
        __syncthreads();
        tile_dot_acc<WKS>(st, buf, qf, x, g);
    }
    float tmax = -FLT_MAX;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = 16 * f + 4 * g + r;
            const bool exists = kk < p.Sk;
            const bool vis = valid[kk] && (!p.causal || kk <= qi + off);
            st[f][r] = exists ? (vis ? st[f][r] * p.scale : -FLT_MAX) : -INFINITY;
            tmax = fmaxf(tmax, st[f][r]);
        }
    const float m = group4_max(tmax);
    float lsum = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pj = __expf(st[f][r] - m);
            st[f][r] = pj;
            lsum += pj;
        }
    const float l = group4_sum(lsum);
    const float inv = 1.f / l;
    bf16_t* O = reinterpret_cast<bf16_t*>(p.out) + p.bsq * p.ldo;
    for (int c = 0; c < nch; ++c) {
        const int hc = min(WCH, p.hd - c * WCH), ho = head_off + c * WCH;
        char* buf = ((c + nch) & 1) ? Vs : Ks;
        __syncthreads();                                     // two chunks back is consumed: its image may be overwritten
        stage<WKS>(buf, nullptr, V, p.ldv, 0, p.Sk, hc, ho);
        __syncthreads();
        f32x4 acc[WD16];
#pragma unroll
        for (int dm = 0; dm < WD16; ++dm) acc[dm] = (f32x4){0.f, 0.f, 0.f, 0.f};
        tile_accumulate<WKS, WD16>(acc, buf, st, x, g);
        if (active) {
#pragma unroll
            for (int dm = 0; dm < WD16; ++dm) store4(O, p.ldo, qi, ho, 16 * dm + 4 * g, hc, acc[dm], inv);
        }
    }
    if (active && p.lse && g == 0) p.lse[((int64_t)b * p.H + h) * p.stat_ld + qi] = m + __logf(l);
}

__global__ __launch_bounds__(256) void bwd_wide_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* DOs = Qs + Geo<WKS>::ROW_BYTES;
    char* Ks = DOs + Geo<WKS>::ROW_BYTES;
    char* DSs = Ks + Geo<WKS>::ROW_BYTES;                                 // [64 queries][64 keys] bf16, pitch DS_PITCH
    constexpr int DS_PITCH = TILE * 2 + 16;
    float* stats = reinterpret_cast<float*>(DSs + TILE * DS_PITCH);       // lse[64], delta[64]

    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    if (!window(p, b, 0, false)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, g = lane >> 4;
    const int item = wave * 16 + x;                   // this lane's key (P, dS, dV, dK) and query (delta, dQ)
    const int off = p.Sk - p.Sq, head_off = h * p.hd;
    const int nch = (p.hd + WCH - 1) / WCH;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + p.bsq * p.ldq;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + p.bsk * p.ldk;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + p.bsk * p.ldv;
    const bf16_t* O = reinterpret_cast<const bf16_t*>(p.o) + p.bsq * p.ldo;
    const bf16_t* DO = reinterpret_cast<const bf16_t*>(p.d_o) + p.bsq * p.lddo;
    const bool qa = item < p.Sq, kactive = item < p.Sk;
    const int64_t stat_at = ((int64_t)b * p.H + h) * p.stat_ld + item;
    const bool kvalid = kactive && (!p.key_mask || p.key_mask[(int64_t)b * p.ld_mask + item] != 0);

    // ---- pass 1 over the chunks: delta = rowsum(dO * O) of query `item`; S and dP of key `item` against all queries
    f32x4 sc[4], dp[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { sc[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    float dsum = 0.f;
    for (int c = 0; c < nch; ++c) {
        const int hc = min(WCH, p.hd - c * WCH), ho = head_off + c * WCH;
        __syncthreads();
        stage<WKS>(Qs, nullptr, Q, p.ldq, 0, p.Sq, hc, ho);
        stage<WKS>(DOs, nullptr, DO, p.lddo, 0, p.Sq, hc, ho);
        bf16x8 kf[WKS], vf[WKS], of[WKS];
        load_bfrag<WKS>(kf, K, p.ldk, item, kactive, hc, ho, g);
        load_bfrag<WKS>(vf, V, p.ldv, item, kactive, hc, ho, g);
        load_bfrag<WKS>(of, O, p.ldo, item, qa, hc, ho, g);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < WKS; ++s) {
            const bf16x8 dof = *reinterpret_cast<const bf16x8*>(DOs + item * Geo<WKS>::PR + (32 * s + 8 * g) * 2);
#pragma unroll
            for (int j = 0; j < 8; ++j) dsum += (float)dof[j] * (float)of[s][j];
        }
        tile_dot_acc<WKS>(sc, Qs, kf, x, g);
        tile_dot_acc<WKS>(dp, DOs, vf, x, g);
    }
    {
        const float delta = group4_sum(dsum);
        if (g == 0) {
            stats[item] = qa ? p.lse[stat_at] : 0.f;
            stats[TILE + item] = qa ? delta : 0.f;
            if (qa) p.delta[stat_at] = delta;
        }
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = 16 * f + 4 * g + r;
            const bool exists = qq < p.Sq;
            const bool vis = exists && kvalid && (!p.causal || item <= qq + off);
            const float pj = exists ? __expf((vis ? sc[f][r] * p.scale + rel_bias_at(p, h, item, qq + off) : -FLT_MAX) - stats[qq]) : 0.f;
            sc[f][r] = pj;                                                   // P
            dp[f][r] = pj * (dp[f][r] - stats[TILE + qq]) * p.scale;        // dS
            *reinterpret_cast<bf16_t*>(DSs + qq * DS_PITCH + item * 2) = (bf16_t)(kactive ? dp[f][r] : 0.f);
        }
    bf16x8 dsf[2];                                    // B fragments of dQ^T += K^T . dS^T: 8 consecutive keys of query `item`
    __syncthreads();                                  // the dS image is complete
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) dsf[s2] = *reinterpret_cast<const bf16x8*>(DSs + item * DS_PITCH + (32 * s2 + 8 * g) * 2);

    // ---- pass 2 over the chunks: dV, dK of key `item`, dQ of query `item`
    bf16_t* DK = reinterpret_cast<bf16_t*>(p.dk) + p.bsk * p.lddk;
    bf16_t* DV = reinterpret_cast<bf16_t*>(p.dv) + p.bsk * p.lddv;
    bf16_t* DQ = reinterpret_cast<bf16_t*>(p.dq) + p.bsq * p.lddq;
    for (int c = nch - 1; c >= 0; --c) {              // the last chunk of pass 1 is still staged in Qs / DOs
        const int hc = min(WCH, p.hd - c * WCH), ho = head_off + c * WCH;
        if (c != nch - 1) {
            __syncthreads();
            stage<WKS>(Qs, nullptr, Q, p.ldq, 0, p.Sq, hc, ho);
            stage<WKS>(DOs, nullptr, DO, p.lddo, 0, p.Sq, hc, ho);
        }
        stage<WKS>(Ks, nullptr, K, p.ldk, 0, p.Sk, hc, ho);
        __syncthreads();
        {
            f32x4 dv[WD16];
#pragma unroll
            for (int dm = 0; dm < WD16; ++dm) dv[dm] = (f32x4){0.f, 0.f, 0.f, 0.f};
            tile_accumulate<WKS, WD16>(dv, DOs, sc, x, g);
            if (kactive) {
#pragma unroll
                for (int dm = 0; dm < WD16; ++dm) store4(DV, p.lddv, item, ho, 16 * dm + 4 * g, hc, dv[dm], 1.f);
            }
        }
        {
            f32x4 dk[WD16];
#pragma unroll
            for (int dm = 0; dm < WD16; ++dm) dk[dm] = (f32x4){0.f, 0.f, 0.f, 0.f};
            tile_accumulate<WKS, WD16>(dk, Qs, dp, x, g);
            if (kactive) {
#pragma unroll
                for (int dm = 0; dm < WD16; ++dm) store4(DK, p.lddk, item, ho, 16 * dm + 4 * g, hc, dk[dm], 1.f);
            }
        }
        {
            f32x4 dq[WD16];
#pragma unroll
            for (int dm = 0; dm < WD16; ++dm) dq[dm] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int q = x >> 2, pp = x & 3;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const char* row_lo = Ks + (32 * s2 + 8 * g + q) * Geo<WKS>::PR + 8 * pp;     // keys 32 s2 + 8 g .. +3
                const char* row_hi = row_lo + 4 * Geo<WKS>::PR;                                // keys .. +4 .. +7
#pragma unroll
                for (int dm = 0; dm < WD16; ++dm) {
                    const bf16x4 lo = lds_tr4(row_lo + 32 * dm);
                    const bf16x4 hi = lds_tr4(row_hi + 32 * dm);
                    bf16x8 a;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a[r] = lo[r]; a[4 + r] = hi[r]; }
                    dq[dm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, dsf[s2], dq[dm], 0, 0, 0);
                }
            }
            if (qa) {
#pragma unroll
                for (int dm = 0; dm < WD16; ++dm) store4(DQ, p.lddq, item, ho, 16 * dm + 4 * g, hc, dq[dm], 1.f);
            }
        }
    }
}

bool supported_wide(int hd, int Sq, int Sk) { return hd > 128 && hd % 8 == 0 && Sq <= TILE && Sk <= TILE; }

// which: 0 forward, 3 all three gradients
int run_wide(int which, const Params& p, hipStream_t s) {
    const size_t row = Geo<WKS>::ROW_BYTES;
    if (which == 0) {
        hipLaunchKernelGGL(fwd_wide_kernel, dim3(1, p.B * p.H), dim3(256), 2 * row + TILE * 4, s, p);
    } else if (which == 3) {
        hipLaunchKernelGGL(bwd_wide_kernel, dim3(1, p.B * p.H), dim3(256), 3 * row + TILE * (TILE * 2 + 16) + 2 * TILE * 4, s, p);
    } else {
        return EAVQA_E_ARG;
    }
    if (hipGetLastError() != hipSuccess) return EAVQA_E_LAUNCH;
    return EAVQA_OK;
}

}  // namespace eavqa_attn_mfma
