// bf16 MFMA GEMM, full-cache-line operand staging (included by gemm.hip, inside its anonymous namespace).
//
// Why this family exists (tools/intake_probe.hip, profiles/round2_intake_probe.md): what bounds every GEMM of the hot path is
// the rate at which ONE compute unit takes operand bytes in from its XCD's L2, and that rate depends on the granule of the
// request, not on the bytes: LDS-DMA of 64-byte row pieces (BK = 32, the round-1 kernels) 57-59 GB/s per CU, of whole 128-byte
// lines (BK = 64) 83-88 GB/s, of 1 KiB contiguous per wave instruction 90-101 GB/s (vector-L1 hits: 140-148).  So:
//   * BK = 64: every row piece of both operands is one whole 128-byte line (8 rows per wave instruction);
//   * LDS image [rows][128 B], 16-byte chunk kc of row r stored at slot kc ^ (r & 7) (applied on the per-lane SOURCE address,
//     LDS-DMA writes linearly; the same involution on the fragment read): ds_read_b128 conflict-free;
//   * a ring of NST stages with counted vmcnt and ONE s_barrier per K-step, DMA in flight across barriers;
//   * two 32-deep MFMA sub-steps per stage; with DB the fragments of the next sub-step are fetched into a second register
//     set while the current one multiplies (4-wave and 8-wave tiles, <= 2 waves per SIMD); without DB (16 resident waves'
//     worth of accumulators, the 256 x 256 tile) the co-resident waves cover each other's LDS latency;
//   * tile = (16 MF WM) x (16 NF WN), WM x WN waves; the B stage moves as BFULL full-wave pieces plus one BREM-lane piece per
//     wave so that every wave issues the same NDMA pieces per stage and the counted waits stay exact;
//   * rows beyond M / N are clamped to the last valid row (their products only reach outputs that are never stored);
//   * C leaves through the ring's LDS in passes of whole wave rows (16-byte row-contiguous global accesses, shared epilogue).
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(vm_only(N)); }

// (Round 2's NON-specialised BK = 64 family - every wave both issuing DMA and multiplying - measured within +-10 % of the round-1
// kernels and was superseded by the loader / consumer specialised kernels below; round 3 removed it from the library:
// profiles/round2_gemm_k64.md keeps its numbers.)

// ================================================================ loader / consumer specialisation ===
// Round-2 finding (k64 sweep, profiles/round2_gemm_k64.md): with every wave both issuing its share of the stage's LDS-DMA pieces
// and multiplying, a K-step costs the SUM of the two - an LDS-DMA piece holds its wave's instruction stream for ~100 cycles
// (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"), 7 pieces per K-step on the 128 x 80 tile = 700 cycles beside 320 cycles of
// MFMA - which is why the full-line staging alone did not reach the probe's 83-91 GB/s per CU.  Here the workgroup carries LW
// extra LOADER waves that only issue the DMA (and wait for it), while the WM x WN CONSUMER waves only read fragments and
// multiply; one s_barrier per K-step joins the two roles:
//   loader   step t:  wait until tile t+1 has landed -> barrier -> issue tile t+NST into the stage tile t occupied
//   consumer step t:  fragments of (t, k 32..63) -> MFMA (t, k 0..31) -> [all reads of stage t returned] barrier ->
//                     fragments of (t+1, k 0..31) -> MFMA (t, k 32..63)
// so a K-step costs max(issue, multiply).  The stage is one image [TBM A rows | TBN B rows] x 128 B cut into pieces of 8 rows;
// loader w moves pieces w, w + LW, ...; loaders with one piece more wait on their own count (wave-uniform branch).
// All waves take part in the C staging / store passes of the epilogue.
template <int WM, int WN, int MF, int NF, int NST_, int LW_> struct K64SGeo {
    static constexpr int NST = NST_, LW = LW_;
    static constexpr int NC = WM * WN;                                   // consumer waves
    static constexpr int NT = 64 * (NC + LW);
    static constexpr int TBM = 16 * MF * WM, TBN = 16 * NF * WN;
    static constexpr int PT = (TBM + TBN) / 8;                           // 1-KiB pieces (8 rows x 128 B) per stage
    static constexpr int P_LO = PT / LW, P_HI = (PT + LW - 1) / LW, N_HI = PT % LW;   // loaders w < N_HI move P_HI pieces
    static constexpr int AOPER = TBM * 128, STAGE = (TBM + TBN) * 128, RING = NST * STAGE;
    static constexpr int PITCH = TBN + 4;
    static constexpr int SP = (16 * MF * WM * PITCH * 4 <= RING) ? WM : ((16 * MF * (WM / 2) * PITCH * 4 <= RING && WM >= 2) ? WM / 2 : 1);
    static constexpr int MFC = (16 * MF * PITCH * 4 <= RING) ? MF : MF / 2;
    static constexpr int PROWS = 16 * MFC * SP;
    static constexpr int TPR = TBN / 4;
    static constexpr int RPP = NT / TPR, NPASS = (PROWS + RPP - 1) / RPP;
    static_assert((TBM + TBN) % 8 == 0 && P_LO >= 1, "stage must cut into whole pieces, at least one per loader");
    static_assert(16 * MFC * PITCH * 4 <= RING && MF % MFC == 0 && (MFC == MF || SP == 1), "C staging pass must fit the ring");
    static_assert(WM % SP == 0 && RING <= 160 * 1024 && NT <= 1024, "geometry");
    static_assert(P_HI * (NST - 1) <= 63 && NST >= 2 && NST <= 6, "vmcnt field");
};

// The loader role (shared by the bf16 and the fp8 kernels): operands are addressed in BYTES - a stage row is 128 bytes of the
// operand's K extent, whatever the element type (64 bf16 or 128 fp8).
template <class G>
__device__ __forceinline__ void k64s_loader_role(const char* A, const char* B, int64_t lda_bytes, int64_t ldb_bytes, int M, int N, int m0,
                                                 int n0, int nk, char* smem, int lw, int lane) {
    constexpr int NST = G::NST;
    const bool hi = lw < G::N_HI;
    const char* src[G::P_HI];
    int dst[G::P_HI];
#pragma unroll
    for (int i = 0; i < G::P_HI; ++i) {
        const int piece = min(lw + G::LW * i, G::PT - 1);                   // (the extra slot of a P_LO loader is never issued)
        const int row = piece * 8 + (lane >> 3), slot = lane & 7;           // row & 7 == lane >> 3
        const int kc = (slot ^ (lane >> 3)) * 16;
        src[i] = row < G::TBM ? A + (int64_t)min(m0 + row, M - 1) * lda_bytes + kc
                              : B + (int64_t)min(n0 + row - G::TBM, N - 1) * ldb_bytes + kc;
        dst[i] = piece * 1024;
    }
    auto issue = [&](int kt, int stage) {
        char* st = smem + stage * G::STAGE;
#pragma unroll
        for (int i = 0; i < G::P_HI; ++i)
            if (i < G::P_LO || hi)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (int64_t)kt * 128),
                                                 (__attribute__((address_space(3))) void*)(st + dst[i]), 16, 0, 0);
    };
    auto wait_tiles = [&](int tiles) {                                      // at most `tiles` later tiles of this wave still in flight
        if (hi) {
            if (NST >= 6 && tiles >= 5) wait_vm<(NST >= 6 ? 5 : 0) * G::P_HI>();
            else if (NST >= 5 && tiles >= 4) wait_vm<(NST >= 5 ? 4 : 0) * G::P_HI>();
            else if (NST >= 4 && tiles >= 3) wait_vm<(NST >= 4 ? 3 : 0) * G::P_HI>();
            else if (NST >= 3 && tiles >= 2) wait_vm<(NST >= 3 ? 2 : 0) * G::P_HI>();
            else if (tiles >= 1) wait_vm<G::P_HI>();
            else wait_vm<0>();
        } else {
            if (NST >= 6 && tiles >= 5) wait_vm<(NST >= 6 ? 5 : 0) * G::P_LO>();
            else if (NST >= 5 && tiles >= 4) wait_vm<(NST >= 5 ? 4 : 0) * G::P_LO>();
            else if (NST >= 4 && tiles >= 3) wait_vm<(NST >= 4 ? 3 : 0) * G::P_LO>();
            else if (NST >= 3 && tiles >= 2) wait_vm<(NST >= 3 ? 2 : 0) * G::P_LO>();
            else if (tiles >= 1) wait_vm<G::P_LO>();
            else wait_vm<0>();
        }
    };
#pragma unroll
    for (int i = 0; i < NST; ++i)
        if (i < nk) issue(i, i);
    wait_tiles(min(nk, NST) - 1);
    __builtin_amdgcn_s_barrier();                                            // tile 0 visible to the consumers
    int stage = 0;
    for (int t = 0; t + 1 < nk; ++t) {
        wait_tiles(min(nk - 1, t + NST - 1) - (t + 1));                      // tile t+1 landed
        __builtin_amdgcn_s_barrier();                                        // ... and every consumer is done with stage t
        if (t + NST < nk) issue(t + NST, stage);
        stage = (stage + 1 == NST) ? 0 : stage + 1;
    }
}

// C tile of the consumer waves -> LDS (the ring) -> shared epilogue, in passes; ALL waves of the workgroup take part.
template <class G, int WM, int WN, int MF, int NF, bool LNX = false>
__device__ __forceinline__ void k64s_store_tile(const GemmParams& p, const f32x4 (&acc)[MF][NF], char* smem, int wave, int lane, int m0, int n0,
                                                const float2* rowstat = nullptr) {
    const int wm = wave / WN, wn = wave % WN;
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CH = MF / G::MFC;
    constexpr int NPASSES = (WM / G::SP) * CH;
    for (int pass = 0; pass < NPASSES; ++pass) {
        const int wgrp = pass / CH, chunk = pass % CH;
        if (wave < G::NC && wm / G::SP == wgrp) {
            const int r0 = (wm % G::SP) * 16 * G::MFC;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                if (i / G::MFC != chunk) continue;
#pragma unroll
                for (int j = 0; j < NF; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = r0 + (i % G::MFC) * 16 + (lane >> 4) * 4 + r;
                        const int col = wn * 16 * NF + j * 16 + (lane & 15);
                        Cs[row * G::PITCH + col] = acc[i][j][r];
                    }
            }
        }
        __syncthreads();
        const int row_base = wgrp * G::SP * 16 * MF + chunk * 16 * G::MFC;
        epilogue<bf16_t, EpiGeo<G::TPR, G::RPP, G::NPASS, G::PITCH, G::PROWS>, LNX>(p, Cs, m0 + row_base, n0, LnArgs{rowstat, row_base});
        if (pass + 1 < NPASSES) __syncthreads();
    }
}

// LNX: the eavqa_gemm_ln form (its own instantiation: the plain kernels stay what they were, instruction for instruction)
template <int WM, int WN, int MF, int NF, int NST, int LW, bool LNX>
__global__ __launch_bounds__(64 * (WM * WN + LW)) void gemm_bf16_k64s_kernel(typename KernArg<LNX>::type pk, int gx, int gy, int tiles_m, int tiles_n) {
    const GemmParams p = widen(pk);
    using G = K64SGeo<WM, WN, MF, NF, NST, LW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tm, tn;
    if (!shaped_tile(gx, gy, tiles_m, tiles_n, tm, tn)) return;
    const int m0 = tm * G::TBM, n0 = tn * G::TBN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // p.ablate (eavqa_gemm_ex, timing only - results are wrong): bit 0 = no C staging / stores, bit 1 = one K-step only,
    // bit 2 = nothing but the launch (every wave returns at once)
    if (p.ablate & 4) return;
    const int nk = (p.ablate & 2) ? 1 : (p.K >> 6);
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wm = wave / WN, wn = wave % WN;                               // meaningful for consumers (wave < NC)
    float2* rowstat = reinterpret_cast<float2*>(smem + G::RING);            // present when launched with ln_lds(p) extra bytes

    if (wave >= G::NC) {
        k64s_loader_role<G>(reinterpret_cast<const char*>(p.A), reinterpret_cast<const char*>(p.B), p.lda * 2, p.ldb * 2, p.M, p.N, m0, n0, nk,
                            smem, wave - G::NC, lane);
    } else {
        // ------------------------------------------------------------------ consumer
        const int frow = lane & 15, fk = lane >> 4;
        const int sw0 = ((fk ^ (frow & 7)) << 4);
        const int a_off = (wm * 16 * MF + frow) * 128;
        const int b_off = G::AOPER + (wn * 16 * NF + frow) * 128;
        bf16x8 fa[2][MF], fb[2][NF];
        auto read_frags = [&](int set, const char* st, int sw) {
#pragma unroll
            for (int i = 0; i < MF; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(st + a_off + i * 2048 + sw);
#pragma unroll
            for (int j = 0; j < NF; ++j) fb[set][j] = *reinterpret_cast<const bf16x8*>(st + b_off + j * 2048 + sw);
        };
        auto mfma_all = [&](int set) {
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
        };
        if (LNX) ln_rowstat_fill(p, rowstat, m0, n0, G::TBM, tid, 64 * G::NC);   // eavqa_gemm_ln: under the first tile's round trip
        __builtin_amdgcn_s_barrier();
        read_frags(0, smem, sw0);
        int stage = 0;
        for (int t = 0; t < nk; ++t) {
            const char* st = smem + stage * G::STAGE;
            const int nstage = (stage + 1 == NST) ? 0 : stage + 1;
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);                             // lgkmcnt(0): set 0 complete
            read_frags(1, st, sw0 ^ 64);
            mfma_all(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);                             // set 1 complete: done reading stage t
            if (t + 1 < nk) {
                __builtin_amdgcn_s_barrier();
                read_frags(0, smem + nstage * G::STAGE, sw0);
            }
            mfma_all(1);
            stage = nstage;
        }
    }
    __syncthreads();   // ring free (every DMA was waited for by its loader before the last K-step's barrier)
    if (p.ablate & 1) {                                                     // keep the accumulators alive without the C pass
        if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.C)[0] = acc[0][0][0];
        return;
    }
    k64s_store_tile<G, WM, WN, MF, NF, LNX>(p, acc, smem, wave, lane, m0, n0, rowstat);
}

// WITH_LN: also instantiate the eavqa_gemm_ln form of this tile (the tiles the dispatcher picks from; the knob-only experiments do without -
// every instantiation costs half a minute of compile time - and answer EAVQA_E_SHAPE to an eavqa_gemm_ln_ex call that forces them)
template <int WM, int WN, int MF, int NF, int NST, int LW, bool WITH_LN = false>
int launch_k64s(const GemmParams& p, hipStream_t stream) {
    using G = K64SGeo<WM, WN, MF, NF, NST, LW>;
    const bool ln = p.ln_stats || p.stats_out || p.copy_out;
    if (ln && !WITH_LN) return EAVQA_E_SHAPE;
    static std::atomic<bool> configured{false};        // atomic: concurrent first calls only repeat an idempotent call
    if (!configured.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_k64s_kernel<WM, WN, MF, NF, NST, LW, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, G::RING) != hipSuccess)
            return EAVQA_E_LAUNCH;
        if constexpr (WITH_LN) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_k64s_kernel<WM, WN, MF, NF, NST, LW, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G::RING + LN_ROWSTAT_BYTES) != hipSuccess)
                return EAVQA_E_LAUNCH;
        }
        configured.store(true, std::memory_order_release);
    }
    static_assert(!WITH_LN || G::RING + LN_ROWSTAT_BYTES <= 160 * 1024, "ring + row statistics of eavqa_gemm_ln");
    const int tiles_m = (p.M + G::TBM - 1) / G::TBM, tiles_n = (p.N + G::TBN - 1) / G::TBN;
    const GridPlan g = plan_grid(tiles_m, tiles_n, G::TBM, G::TBN);
    if constexpr (WITH_LN) {
        if (ln) {
            hipLaunchKernelGGL((gemm_bf16_k64s_kernel<WM, WN, MF, NF, NST, LW, true>), dim3(g.per_xcd * 8), dim3(G::NT), G::RING + ln_lds(p), stream, p,
                               g.gx, g.gy, tiles_m, tiles_n);
            EAVQA_LAUNCH_CHECK();
            return EAVQA_OK;
        }
    }
    hipLaunchKernelGGL((gemm_bf16_k64s_kernel<WM, WN, MF, NF, NST, LW, false>), dim3(g.per_xcd * 8), dim3(G::NT), G::RING, stream, p, g.gx, g.gy,
                       tiles_m, tiles_n);
    EAVQA_LAUNCH_CHECK();
    return EAVQA_OK;
}

// `rate`: ns per 128-byte operand row and K-step (64) of the fullest CU (calibrated on MI355X with tools/gemm_bench.py).
struct K64Choice { int bm, bn, wg_per_cu; float rate; int (*launch)(const GemmParams&, hipStream_t); };
const K64Choice K64_SHAPES[] = {
    // loader / consumer specialised (knob ids 2..)
    {128, 80, 2, 1.1f, launch_k64s<4, 1, 2, 5, 3, 2>},          //  2
    {128, 80, 2, 1.1f, launch_k64s<4, 1, 2, 5, 3, 4>},          //  3
    {128, 128, 2, 1.1f, launch_k64s<2, 2, 4, 4, 2, 4>},         //  4
    {256, 128, 1, 1.97f, launch_k64s<4, 2, 4, 4, 3, 4, true>},  //  5
    {256, 160, 1, 1.93f, launch_k64s<4, 2, 4, 5, 3, 4, true>},  //  6
    {128, 80, 1, 1.56f, launch_k64s<4, 1, 2, 5, 4, 2, true>},   //  7: four stages (one workgroup per CU)
    {128, 96, 1, 1.1f, launch_k64s<4, 1, 2, 6, 3, 2>},          //  8
    {256, 192, 1, 1.1f, launch_k64s<4, 2, 4, 6, 2, 4>},         //  9
    {256, 256, 1, 1.1f, launch_k64s<2, 4, 8, 4, 2, 4>},         // 10
    {128, 128, 1, 1.73f, launch_k64s<2, 2, 4, 4, 3, 2, true>},  // 11: three stages
    {128, 256, 1, 1.97f, launch_k64s<2, 4, 4, 4, 3, 4, true>},  // 12
};
constexpr int N_K64 = sizeof(K64_SHAPES) / sizeof(K64_SHAPES[0]);
// the tiles the dispatcher chooses among (indices into K64_SHAPES): s128x80 (4 stages), s256x128, s256x160, s128x128 (3 stages), s128x256
constexpr int K64_AUTO[] = {5, 3, 4, 9, 10};       // table indices: 128x80 (4 stages), 256x128, 256x160, 128x128 (3 stages), 128x256
constexpr int N_K64_AUTO = sizeof(K64_AUTO) / sizeof(K64_AUTO[0]);


