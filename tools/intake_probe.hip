// intake_probe: how many bytes per second ONE compute unit of an MI355X can take in from its XCD's L2, by load form.
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/intake_probe tools/intake_probe.hip && gpurun_out/intake_probe
//
// Why: DESIGN.md section 6 found every hot GEMM pinned at ~52 GB/s per CU of operand intake (LDS-DMA from L2).  Before
// redesigning the GEMM around another load path this measures, with the GEMM's own access pattern (a workgroup marches over
// K through `rows` operand rows, the 32 workgroups of an XCD share A / B panels like a 4 x 8 rectangle of 128 x 128 tiles),
// the rate of each candidate path with nothing else in the kernel:
//   mode 0  LDS-DMA (global_load_lds_dwordx4), 64-byte row pieces (BK = 32), what the fast / shaped kernels issue today
//   mode 1  LDS-DMA, 128-byte row pieces (BK = 64, whole cache lines), what the 256 x 256 kernel issues
//   mode 2  global_load_dwordx4 to VGPRs, 64-byte row pieces
//   mode 3  global_load_dwordx4 to VGPRs, 128-byte row pieces
//   mode 4  global_load_dwordx4 to VGPRs from a PRE-PACKED operand: every wave instruction reads 1 KiB contiguous
//   mode 5  LDS-DMA from the pre-packed operand (1 KiB contiguous per wave instruction)
//   mode 6  half the bytes as mode 0 (A through LDS), half as mode 4 (pre-packed B to VGPRs): do the two paths add?
//   mode 7  every workgroup re-reads the same 16 KiB (vector-L1 hits): the TA / L1 ceiling
// Sweeps workgroup size / workgroups per CU / loads in flight.  Prints GB/s per CU (requested bytes / time / 256).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int K_BYTES = 5120 * 2;          // operand row pitch (K = 5120 bf16)
constexpr int ROWS_A = 512, ROWS_B = 1024; // rows of the A / B panels one XCD's rectangle touches (4 x 8 tiles of 128)

__device__ __forceinline__ void dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// One "k-step" moves PIECES x (NT x 16) bytes per workgroup; D steps are kept in flight.
// LINE: bytes of one row piece (64 or 128); packed: consecutive 16-byte chunks of a wave instruction are consecutive in memory.
template <int MODE, int NT, int PIECES, int D>
__global__ __launch_bounds__(NT) void probe(const char* A, const char* B, const char* Ap, const char* Bp, int steps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, local = (blockIdx.x >> 3) & 31;
    const int tm = local & 3, tn = local >> 2;                 // 4 x 8 rectangle of 128-row tiles inside the XCD's panels
    constexpr int LINE = (MODE == 1 || MODE == 3) ? 128 : 64;
    constexpr int LPR = LINE / 16;                             // lanes per row piece
    constexpr int STEP_BYTES = PIECES * NT * 16;
    constexpr int ROWS_STEP = STEP_BYTES / LINE;               // rows per step (A half + B half)
    // per-thread sources for the row-piece modes: piece p covers chunk c = tid + NT p -> row c / LPR, slot c % LPR
    const char* src[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        const int c = tid + NT * p, row = c / LPR, slot = c % LPR;
        const bool isA = row < ROWS_STEP / 2;
        const int r = isA ? (tm * 128 + row) % ROWS_A : (tn * 128 + (row - ROWS_STEP / 2)) % ROWS_B;
        src[p] = (isA ? A : B) + (size_t)xcd * 0 + (size_t)r * K_BYTES + slot * 16;
    }
    // pre-packed operand: [tile][kstep][STEP_BYTES/2] contiguous per operand
    const char* psrcA = Ap + (size_t)tm * (size_t)steps * (STEP_BYTES / 2);
    const char* psrcB = Bp + (size_t)tn * (size_t)steps * (STEP_BYTES / 2);
    unsigned acc = 0;
    if (MODE == 0 || MODE == 1 || MODE == 5 || MODE == 7) {
        // LDS-DMA ring of D stages
        auto issue = [&](int s) {
            char* st = smem + (s % D) * STEP_BYTES;
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                const char* g;
                if (MODE == 5) g = (p < PIECES / 2 ? psrcA + (size_t)s * (STEP_BYTES / 2) + (p * NT + tid) * 16
                                                    : psrcB + (size_t)s * (STEP_BYTES / 2) + ((p - PIECES / 2) * NT + tid) * 16);
                else if (MODE == 7) g = A + (p * NT + tid) * 16;
                else g = src[p] + (size_t)s * LINE;
                dma16(g, st + p * NT * 16 + wave * 1024);
            }
        };
        for (int s = 0; s < D - 1 && s < steps; ++s) issue(s);
        for (int s = 0; s < steps; ++s) {
            if (s + D - 1 < steps) issue(s + D - 1);
            // leave D-1 steps in flight (counted wait), like the GEMM ring
            if (D == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((1 * PIECES) & 15) | (((1 * PIECES) >> 4) << 14));
            else if (D == 4) __builtin_amdgcn_s_waitcnt(0x0F70 | ((3 * PIECES) & 15) | (((3 * PIECES) >> 4) << 14));
            else __builtin_amdgcn_s_waitcnt(0x0F70 | ((7 * PIECES) & 15) | (((7 * PIECES) >> 4) << 14));
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        acc = *reinterpret_cast<unsigned*>(smem + tid * 4);
    } else if (MODE == 2 || MODE == 3 || MODE == 4) {
        uint4 ring[D][PIECES];
        auto ld = [&](int s, int p) -> uint4 {
            const char* g;
            if (MODE == 4) g = (p < PIECES / 2 ? psrcA + (size_t)s * (STEP_BYTES / 2) + (p * NT + tid) * 16
                                                : psrcB + (size_t)s * (STEP_BYTES / 2) + ((p - PIECES / 2) * NT + tid) * 16);
            else g = src[p] + (size_t)s * LINE;
            return *reinterpret_cast<const uint4*>(g);
        };
#pragma unroll
        for (int s = 0; s < D - 1; ++s)
#pragma unroll
            for (int p = 0; p < PIECES; ++p) ring[s][p] = ld(s, p);
        for (int s0 = 0; s0 < steps; s0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int s = s0 + u;
                if (s + D - 1 < steps) {
#pragma unroll
                    for (int p = 0; p < PIECES; ++p) ring[(u + D - 1) % D][p] = ld(s + D - 1, p);
                }
#pragma unroll
                for (int p = 0; p < PIECES; ++p) acc ^= ring[u][p].x ^ ring[u][p].y ^ ring[u][p].z ^ ring[u][p].w;
            }
        }
    } else if (MODE == 6 || MODE == 8) {
        // MODE 6: A half by LDS-DMA (64-byte row pieces), B half pre-packed to VGPRs.  MODE 8: both halves pre-packed to VGPRs.
        // The VGPR loads are inline asm with hand-counted vmcnt (beside an LDS-DMA in flight hipcc would wait vmcnt(0) for
        // every ordinary load and drain the ring).
        constexpr int HP = PIECES / 2;
        constexpr int NV = MODE == 6 ? HP : PIECES;           // VGPR loads per step
        u32x4 ring[D][NV];
        auto issueA = [&](int s) {
            char* st = smem + (s % D) * (STEP_BYTES / 2);
#pragma unroll
            for (int p = 0; p < HP; ++p) {
                const int c = tid + NT * p, row = c / 4, slot = c % 4;
                const char* g = A + (size_t)((tm * 128 + row) % ROWS_A) * K_BYTES + slot * 16 + (size_t)s * 64;
                dma16(g, st + p * NT * 16 + wave * 1024);
            }
        };
        auto ldv = [&](u32x4& r, int s, int p) {
            const char* g = (MODE == 8 && p >= HP) ? psrcA + (size_t)s * (STEP_BYTES / 2) + ((p - HP) * NT + tid) * 16
                                                    : psrcB + (size_t)s * (STEP_BYTES / 2) + (p * NT + tid) * 16;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(g) : "memory");
        };
#pragma unroll
        for (int s = 0; s < D - 1; ++s) {
            if (MODE == 6) issueA(s);
#pragma unroll
            for (int p = 0; p < NV; ++p) ldv(ring[s][p], s, p);
        }
        for (int s0 = 0; s0 < steps; s0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int s = s0 + u;
                if (s + D - 1 < steps) {
                    if (MODE == 6) issueA(s + D - 1);
#pragma unroll
                    for (int p = 0; p < NV; ++p) ldv(ring[(u + D - 1) % D][p], s + D - 1, p);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PIECES) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
#pragma unroll
                for (int p = 0; p < NV; ++p) acc ^= ring[u][p][0] ^ ring[u][p][1] ^ ring[u][p][2] ^ ring[u][p][3];
                if (MODE == 6) __builtin_amdgcn_s_barrier();
            }
        }
        if (MODE == 6) {
            __syncthreads();
            acc ^= *reinterpret_cast<unsigned*>(smem + tid * 4);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;     // never true in practice: keeps the loads alive
}

template <int MODE, int NT, int PIECES, int D>
void run(const char* name, int wg_per_cu, const char* A, const char* B, const char* Ap, const char* Bp, unsigned* sink) {
    constexpr int STEP_BYTES = PIECES * NT * 16;
    constexpr int LINE = (MODE == 1 || MODE == 3) ? 128 : 64;
    const int steps = (K_BYTES / LINE) / D * D;      // march over the whole K once (a multiple of the ring depth)
    const size_t lds = (MODE == 0 || MODE == 1 || MODE == 5 || MODE == 7) ? (size_t)D * STEP_BYTES : (MODE == 6 ? (size_t)D * STEP_BYTES / 2 : 0);
    // occupancy: request enough dynamic LDS that exactly wg_per_cu workgroups fit a CU (160 KiB)
    size_t lds_req = lds;
    const size_t floor_req = 160 * 1024 / (wg_per_cu + 1) + 1024;
    if (lds_req < floor_req) lds_req = floor_req;
    if (lds_req * wg_per_cu > 160 * 1024) { printf("%s: %d workgroups of %zu B LDS do not fit a CU, skipped\n", name, wg_per_cu, lds_req); return; }
    auto kern = probe<MODE, NT, PIECES, D>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds_req, 0, A, B, Ap, Bp, steps, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds_req, 0, A, B, Ap, Bp, steps, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid * steps * STEP_BYTES;
    const double us = ms * 1e3 / reps;
    printf("%-44s NT=%4d wg/CU=%d step=%3d KB D=%d  %8.1f us  %7.1f GB/s per CU  %6.2f TB/s chip\n", name, NT, wg_per_cu, STEP_BYTES / 1024, D, us,
           bytes / (us * 1e-6) / 256 / 1e9, bytes / (us * 1e-6) / 1e12);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs, clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    const size_t nA = (size_t)ROWS_A * K_BYTES, nB = (size_t)ROWS_B * K_BYTES;
    const size_t nP = (size_t)8 * K_BYTES * 1024;          // packed copies: generous
    char *A, *B, *Ap, *Bp; unsigned* sink;
    CK(hipMalloc(&A, nA)); CK(hipMalloc(&B, nB)); CK(hipMalloc(&Ap, nP)); CK(hipMalloc(&Bp, nP)); CK(hipMalloc(&sink, 64));
    std::vector<unsigned> h(nP / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u) | 1u;
    CK(hipMemcpy(A, h.data(), nA, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), nB, hipMemcpyHostToDevice));
    CK(hipMemcpy(Ap, h.data(), nP, hipMemcpyHostToDevice)); CK(hipMemcpy(Bp, h.data(), nP, hipMemcpyHostToDevice));

    // 256 threads: 16 KiB per step = 4 pieces (128 x 128 tile at BK 32: A 8 KiB + B 8 KiB)
    run<0, 256, 4, 4>("0 LDS-DMA 64B pieces", 1, A, B, Ap, Bp, sink);
    run<0, 256, 4, 4>("0 LDS-DMA 64B pieces", 2, A, B, Ap, Bp, sink);
    run<0, 256, 4, 8>("0 LDS-DMA 64B pieces", 1, A, B, Ap, Bp, sink);
    run<1, 256, 8, 2>("1 LDS-DMA 128B pieces", 1, A, B, Ap, Bp, sink);
    run<1, 256, 8, 2>("1 LDS-DMA 128B pieces", 2, A, B, Ap, Bp, sink);
    run<1, 256, 8, 4>("1 LDS-DMA 128B pieces", 1, A, B, Ap, Bp, sink);
    run<5, 256, 4, 4>("5 LDS-DMA packed 1KiB", 1, A, B, Ap, Bp, sink);
    run<5, 256, 4, 4>("5 LDS-DMA packed 1KiB", 2, A, B, Ap, Bp, sink);
    run<5, 256, 4, 8>("5 LDS-DMA packed 1KiB", 1, A, B, Ap, Bp, sink);
    run<2, 256, 4, 4>("2 VGPR 64B pieces", 1, A, B, Ap, Bp, sink);
    run<2, 256, 4, 4>("2 VGPR 64B pieces", 2, A, B, Ap, Bp, sink);
    run<2, 256, 4, 4>("2 VGPR 64B pieces", 4, A, B, Ap, Bp, sink);
    run<3, 256, 8, 2>("3 VGPR 128B pieces", 1, A, B, Ap, Bp, sink);
    run<3, 256, 8, 2>("3 VGPR 128B pieces", 2, A, B, Ap, Bp, sink);
    run<3, 256, 8, 2>("3 VGPR 128B pieces", 4, A, B, Ap, Bp, sink);
    run<4, 256, 4, 4>("4 VGPR packed 1KiB", 1, A, B, Ap, Bp, sink);
    run<4, 256, 4, 4>("4 VGPR packed 1KiB", 2, A, B, Ap, Bp, sink);
    run<4, 256, 4, 4>("4 VGPR packed 1KiB", 4, A, B, Ap, Bp, sink);
    run<4, 256, 4, 8>("4 VGPR packed 1KiB", 2, A, B, Ap, Bp, sink);
    run<6, 256, 4, 4>("6 A LDS-DMA 64B + B VGPR packed", 1, A, B, Ap, Bp, sink);
    run<6, 256, 4, 4>("6 A LDS-DMA 64B + B VGPR packed", 2, A, B, Ap, Bp, sink);
    run<8, 256, 4, 4>("8 VGPR packed 1KiB (asm, counted vmcnt)", 1, A, B, Ap, Bp, sink);
    run<8, 256, 4, 4>("8 VGPR packed 1KiB (asm, counted vmcnt)", 2, A, B, Ap, Bp, sink);
    run<8, 256, 4, 8>("8 VGPR packed 1KiB (asm, counted vmcnt)", 1, A, B, Ap, Bp, sink);
    run<7, 256, 4, 4>("7 LDS-DMA same 16 KiB (L1 hits)", 1, A, B, Ap, Bp, sink);
    run<7, 256, 4, 4>("7 LDS-DMA same 16 KiB (L1 hits)", 2, A, B, Ap, Bp, sink);
    // 512 / 1024 threads, one workgroup per CU
    run<0, 512, 4, 4>("0 LDS-DMA 64B pieces", 1, A, B, Ap, Bp, sink);
    run<1, 512, 4, 2>("1 LDS-DMA 128B pieces", 1, A, B, Ap, Bp, sink);
    run<1, 1024, 4, 2>("1 LDS-DMA 128B pieces", 1, A, B, Ap, Bp, sink);
    run<5, 1024, 4, 2>("5 LDS-DMA packed 1KiB", 1, A, B, Ap, Bp, sink);
    run<4, 512, 4, 4>("4 VGPR packed 1KiB", 1, A, B, Ap, Bp, sink);
    run<4, 1024, 2, 4>("4 VGPR packed 1KiB", 1, A, B, Ap, Bp, sink);
    run<3, 1024, 4, 2>("3 VGPR 128B pieces", 1, A, B, Ap, Bp, sink);
    run<6, 512, 4, 4>("6 A LDS-DMA 64B + B VGPR packed", 1, A, B, Ap, Bp, sink);
    printf("done\n");
    return 0;
}
