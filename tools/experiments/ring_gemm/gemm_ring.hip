// 256 x 256 bf16 tile on a deep LDS-DMA ring (included by gemm.hip, inside its anonymous namespace).
//
// Why a second 256 x 256 kernel: probes of the round-1 kernel (profiles/round3_big_kernel_probes.md; M = 41 120, N = 4 096) gave, per
// 64-deep K-tile of a workgroup: MFMAs alone 1.05 us (the matrix pipe at the ~1.96 GHz it holds under this load), its fragment reads
// + MFMAs without any DMA 1.47 us (the compiler waits lgkmcnt(0) right after each group of ds_reads, four times per tile, and all
// four waves of a SIMD stand at the same point), its DMA alone 1.33 us (two 64 KiB stages = ONE tile in flight: a strictly serial
// issue -> land -> barrier round trip), together 1.57-1.8 us; and 9.5 us per tile outside the K loop, 5.5 of them the C epilogue
// (eight __syncthreads per tile, each of which also drains the global stores of the slab before).  This kernel changes all three:
//   * ring of NS slots of 32 KiB (k = 32: 64-byte rows, the swizzle of the round-1 shaped tiles); ONE barrier per slot, placed
//     after the first row of MFMAs of a phase - by then every fragment of the slot has arrived in registers (the MFMAs consumed
//     the last one), so the slot is free and the DMA of phase q + NS goes out there: NS - 1 phases (4 x 32 KiB at NS = 5) are in
//     flight with a lead of NS - 1 phases, instead of one tile with a lead of one;
//   * fragments rotate through ONE register set: A[i] of the next phase is read as soon as row i of this phase has issued, B[j] after
//     the last MFMA that uses it (row 3), so reads run under the MFMAs of the same wave (no second register set: 16 waves leave
//     128 registers per lane);
//   * operands swapped in the MFMA (weights first): a lane then owns four consecutive COLUMNS of a row, the C slab is staged with
//     16 ds_write_b128 per wave instead of 64 ds_write_b32, and the slab barriers wait for LDS only (lgkmcnt), not for the stores.
constexpr int RBK = 32;
constexpr int ROPER = 256 * RBK * 2;               // 16 KiB per operand per slot
constexpr int RSLOT = 2 * ROPER;                   // 32 KiB

__device__ __forceinline__ void lds_barrier() {    // LDS traffic of this wave done, then the workgroup barrier; global stores stay in flight
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
}

#define EAVQA_RING_MFMA(i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
#define EAVQA_RING_ROW(i) EAVQA_RING_MFMA(i, 0) EAVQA_RING_MFMA(i, 1) EAVQA_RING_MFMA(i, 2) EAVQA_RING_MFMA(i, 3)
#define EAVQA_RING_LDA(i) fa[i] = *reinterpret_cast<const bf16x8*>(nx + a_off + (i) * 1024);
#define EAVQA_RING_LDB(j) fb[j] = *reinterpret_cast<const bf16x8*>(nx + b_off + (j) * 1024);

__device__ __forceinline__ void wait_vm_upto(int n) {          // s_waitcnt vmcnt(n) for a wave-uniform n in 0 .. 12 (the count is an immediate)
    switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(vm_only(0)); break;
        case 1: __builtin_amdgcn_s_waitcnt(vm_only(1)); break;
        case 2: __builtin_amdgcn_s_waitcnt(vm_only(2)); break;
        case 3: __builtin_amdgcn_s_waitcnt(vm_only(3)); break;
        case 4: __builtin_amdgcn_s_waitcnt(vm_only(4)); break;
        case 5: __builtin_amdgcn_s_waitcnt(vm_only(5)); break;
        case 6: __builtin_amdgcn_s_waitcnt(vm_only(6)); break;
        case 7: __builtin_amdgcn_s_waitcnt(vm_only(7)); break;
        case 8: __builtin_amdgcn_s_waitcnt(vm_only(8)); break;
        case 9: __builtin_amdgcn_s_waitcnt(vm_only(9)); break;
        case 10: __builtin_amdgcn_s_waitcnt(vm_only(10)); break;
        case 11: __builtin_amdgcn_s_waitcnt(vm_only(11)); break;
        default: __builtin_amdgcn_s_waitcnt(vm_only(12)); break;
    }
}

constexpr int RING_PFD = 6;                        // the L2 prefetch runs this many phases ahead of the DMA issue point

// ABL: timing probes (1 no DMA inside the loop, 2 DMA only, 3 DMA only and every workgroup on tile (0, 0)); results wrong when non-zero.
// PF: L2 prefetch.  The 32 workgroups of an XCD that run beside each other ask for the same panel lines at the same time, so the first
// touch of every line - a trip to the Infinity Cache or HBM - is on everybody's path (probe: all-L2-hit DMA 1.0 us per 64-deep tile,
// the real mix 1.45).  Each wave therefore touches, every second phase, six lines (2 rows of A, 4 of B: this workgroup's share of the
// panels it has in common with its neighbours) of the K-tile RING_PFD phases beyond the DMA issue point with a plain load whose
// result nobody reads - of its own tile, or of the tile 32 places on (the next round on this XCD) once its own K range is used up.
template <int NS, int ABL = 0, bool PF = false>
__global__ __launch_bounds__(1024) void gemm_bf16_ring_kernel(GemmParams p, int gx, int gy, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tm, tn, grp_r, grp_n;
    if (!big_tile_at(p, gx, gy, tiles_m, tiles_n, blockIdx.x & 7, blockIdx.x >> 3, tm, tn, grp_r, grp_n)) return;
    if (ABL == 3) tm = tn = 0;                     // probe: every workgroup streams the same panels (all L2 hits)
    const int m0 = tm * GBM, n0 = tn * GBN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

    // DMA: wave w fills rows 16 w .. 16 w + 15 of both operand images of a slot (1 KiB each): lane -> row 16 w + lane / 4, physical
    // chunk lane & 3, which holds the logical chunk fswz puts there
    const int drow = wave * 16 + (lane >> 2);
    const int lchunk = (lane & 3) ^ ((-(drow >> 2)) & 3);
    const bf16_t* asrc = A + (int64_t)min(m0 + drow, p.M - 1) * p.lda + lchunk * 8;
    const bf16_t* bsrc = B + (int64_t)min(n0 + drow, p.N - 1) * p.ldb + lchunk * 8;
    const int dma_off = wave * 1024;
    auto issue = [&](int q, int slot) {
        char* st = smem + slot * RSLOT + dma_off;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + q * RBK),
                                         (__attribute__((address_space(3))) void*)st, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + q * RBK),
                                         (__attribute__((address_space(3))) void*)(st + ROPER), 16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nq = p.K / RBK;
    const int frow = lane & 15, fk = lane >> 4;
    const int a_off = fswz(wm * 64 + frow, fk);                  // + i * 16 rows * 64 B
    const int b_off = ROPER + fswz(wn * 64 + frow, fk);          // + j * 16 rows * 64 B
    bf16x8 fa[4], fb[4];

#pragma unroll
    for (int q = 0; q < NS; ++q)
        if (q < nq) issue(q, q);
    if (nq >= NS) __builtin_amdgcn_s_waitcnt(vm_only(2 * (NS - 1)));
    else __builtin_amdgcn_s_waitcnt(vm_only(0));
    __builtin_amdgcn_s_barrier();
    {
        const char* nx = smem;
        EAVQA_RING_LDA(0) EAVQA_RING_LDA(1) EAVQA_RING_LDA(2) EAVQA_RING_LDA(3)
        EAVQA_RING_LDB(0) EAVQA_RING_LDB(1) EAVQA_RING_LDB(2) EAVQA_RING_LDB(3)
    }

    // prefetch addresses (bytes): lanes 0, 1 -> rows of A, lanes 2 .. 5 (and, redundantly, the rest) -> rows of B
    const char* pf_cur = nullptr;
    const char* pf_nxt = nullptr;
    bool has_nxt = false;
    if (PF) {
        const int pl = min(lane, 5);
        auto row_ptr = [&](int tm_, int tn_, int r_, int gn_) -> const char* {
            const int sa = (r_ % gn_) & 7, sb = (r_ / gn_) & 3;          // this workgroup among the 8 that share its A panel / the 4 that share its B panel
            if (pl < 2) return reinterpret_cast<const char*>(A + (int64_t)min(tm_ * GBM + sa * 32 + wave * 2 + pl, p.M - 1) * p.lda);
            return reinterpret_cast<const char*>(B + (int64_t)min(tn_ * GBN + sb * 64 + wave * 4 + (pl - 2), p.N - 1) * p.ldb);
        };
        pf_cur = row_ptr(tm, tn, grp_r, grp_n);
        int tm2, tn2, r2, gn2;
        has_nxt = big_tile_at(p, gx, gy, tiles_m, tiles_n, blockIdx.x & 7, (blockIdx.x >> 3) + 32, tm2, tn2, r2, gn2);
        pf_nxt = has_nxt ? row_ptr(tm2, tn2, r2, gn2) : pf_cur;
    }
    const int nk64 = nq >> 1;

    int slot = 0;                                                // slot of phase q
    for (int q = 0; q + 1 < nq; ++q) {
        const int nslot = slot + 1 == NS ? 0 : slot + 1;
        const char* nx = smem + nslot * RSLOT;
        __builtin_amdgcn_sched_barrier(0);
        EAVQA_RING_ROW(0)
        __builtin_amdgcn_sched_barrier(0);
        // phase q + 1 must have landed (this wave's share; the barrier makes it everybody's); phases q + 2 .. q + NS - 1 stay in flight
        if (!PF) {
            if (q + NS - 1 < nq) __builtin_amdgcn_s_waitcnt(vm_only(2 * (NS - 2)));
            else __builtin_amdgcn_s_waitcnt(vm_only(0));
        } else {
            // younger than the DMA of phase q + 1 and allowed to stay out: the DMAs of phases q + 2 .. and the prefetches issued at the
            // even barriers after that DMA's own (q + 2 - NS .. q - 1)
            const int lo = max(q + 2 - NS, 0), hi = q - 1;
            const int npf = hi >= lo ? (hi >> 1) - ((lo + 1) >> 1) + 1 : 0;
            wait_vm_upto(2 * min(NS - 2, nq - q - 2) + npf);
        }
        __builtin_amdgcn_s_barrier();                            // ... and every wave holds all fragments of slot `slot`
        if (PF && !(q & 1)) {
            const int t = (q + NS + RING_PFD) >> 1;              // 64-deep K-tile to touch
            const bool own = t < nk64, nxt = !own && has_nxt && t - nk64 < nk64;
            const char* a = (own ? pf_cur : pf_nxt) + (own ? t : nxt ? t - nk64 : 0) * 128;      // neither: any valid line (keeps the count uniform)
            // 4 bytes per lane by LDS-DMA into a 256-byte dump behind the ring: no destination register that a late return could overwrite
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                             (__attribute__((address_space(3))) void*)(smem + NS * RSLOT), 4, 0, 0);
        }
        if (ABL != 1 && q + NS < nq) issue(q + NS, slot);
        if (ABL >= 2) { slot = nslot; continue; }
        EAVQA_RING_LDA(0)
        __builtin_amdgcn_sched_barrier(0);
        EAVQA_RING_ROW(1)
        EAVQA_RING_LDA(1)
        __builtin_amdgcn_sched_barrier(0);
        EAVQA_RING_ROW(2)
        EAVQA_RING_LDA(2)
        __builtin_amdgcn_sched_barrier(0);
        EAVQA_RING_MFMA(3, 0)
        EAVQA_RING_LDB(0)
        __builtin_amdgcn_sched_barrier(0);
        EAVQA_RING_MFMA(3, 1)
        EAVQA_RING_LDB(1)
        __builtin_amdgcn_sched_barrier(0);
        EAVQA_RING_MFMA(3, 2)
        EAVQA_RING_LDB(2)
        __builtin_amdgcn_sched_barrier(0);
        EAVQA_RING_MFMA(3, 3)
        EAVQA_RING_LDA(3)                                        // B[3] last: the MFMA that consumes it (row 0 of the next phase) then
        EAVQA_RING_LDB(3)                                        // certifies that every read of the slot has returned
        slot = nslot;
    }
    __builtin_amdgcn_sched_barrier(0);
    EAVQA_RING_ROW(0) EAVQA_RING_ROW(1) EAVQA_RING_ROW(2) EAVQA_RING_ROW(3)
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();                                               // ring reads over (every DMA has landed: the last wait covered phase nq - 1)

    // C tile through LDS one 64-row slab at a time (the accumulators of one wave row); a lane holds row (lane & 15) of a 16 x 16
    // fragment, columns 4 (lane >> 4) .. + 3
    float* Cs = reinterpret_cast<float*>(smem);
    for (int slab = 0; slab < 4; ++slab) {
        if (wm == slab) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4*>(&Cs[(i * 16 + (lane & 15)) * GCS_PITCH + wn * 64 + j * 16 + (lane >> 4) * 4]) = acc[i][j];
        }
        lds_barrier();
        epilogue<bf16_t, EpiGeo256>(p, Cs, m0 + slab * 64, n0);
        if (slab < 3) lds_barrier();
    }
}
#undef EAVQA_RING_MFMA
#undef EAVQA_RING_ROW
#undef EAVQA_RING_LDA
#undef EAVQA_RING_LDB

template <int NS, int ABL = 0, bool PF = false>
int launch_ring(const GemmParams& p, hipStream_t stream) {
    static std::atomic<bool> configured{false};        // atomic: concurrent first calls only repeat an idempotent call
    if (!configured.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_ring_kernel<NS, ABL, PF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                NS * RSLOT + (PF ? 256 : 0)) != hipSuccess)
            return EAVQA_E_LAUNCH;
        configured.store(true, std::memory_order_release);
    }
    const int tiles_m = (p.M + GBM - 1) / GBM, tiles_n = (p.N + GBN - 1) / GBN;
    int best_gx = 8, best_cost = 1 << 30;
    const int cand[4] = {8, 4, 2, 1};
    for (int c = 0; c < 4; ++c) {
        const int gx = cand[c], gy = 8 / gx;
        const int cost = (tiles_m + gx - 1) / gx + (tiles_n + gy - 1) / gy;
        if (cost < best_cost) { best_cost = cost; best_gx = gx; }
    }
    const int gx = best_gx, gy = 8 / gx;
    const int per_xcd = ((tiles_m + gx - 1) / gx) * ((tiles_n + gy - 1) / gy);
    hipLaunchKernelGGL((gemm_bf16_ring_kernel<NS, ABL, PF>), dim3(per_xcd * 8), dim3(1024), NS * RSLOT + (PF ? 256 : 0), stream, p, gx, gy, tiles_m, tiles_n);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}
