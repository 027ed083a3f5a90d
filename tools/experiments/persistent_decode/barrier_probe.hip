// Grid-barrier probe: how long does a device-wide barrier between 256 co-resident workgroups take on MI355X (8 XCDs, one L2 each)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/barrier_probe tools/barrier_probe.hip && tools/_bin/barrier_probe
// The barrier is a monotonic counter in global memory (agent-scope atomics) with an agent-scope release before the arrive and an
// acquire after the wait (on gfx950 these write back / invalidate the XCD's L2: the L2s of different XCDs are not coherent).
// Variants: the bare barrier; the barrier with a producer/consumer exchange through global memory (every workgroup writes 4 KiB,
// after the barrier reads the 4 KiB of another workgroup on another XCD and checks them) - the exchange is the correctness check
// the decode-step design needs (partial sums written by one CU, read by another after the barrier).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: release / acquire fences at agent scope + sleeping spin; 1: same fences, tight spin; 2: NO fences (relaxed agent-scope
// atomics only: correct only when the exchanged data bypasses the non-coherent caches, i.e. lives in uncached / fine-grained memory),
// tight spin
template <int MODE>
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (MODE < 2) __atomic_thread_fence(__ATOMIC_RELEASE);
        else __builtin_amdgcn_s_waitcnt(0);                                         // this wave's stores have left
        __hip_atomic_fetch_add(counter, 1u, MODE < 2 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, MODE < 2 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (MODE == 0) __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) { ok = false; break; }                        // never hang
        }
        if (MODE < 2) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    return ok;
}

template <int MODE>
__global__ __launch_bounds__(512) void bare(unsigned* counter, int rounds, unsigned* fail) {
    for (int r = 1; r <= rounds; ++r)
        if (!grid_barrier<MODE>(counter, (unsigned)r * gridDim.x)) { if (threadIdx.x == 0) atomicAdd(fail, 1u); return; }
}

template <int MODE>
__global__ __launch_bounds__(512) void exchange(unsigned* counter, int rounds, unsigned* buf, unsigned* fail) {
    // buf: [2][grid][1024] words (double-buffered by round parity)
    const unsigned nb = gridDim.x, me = blockIdx.x;
    unsigned bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        unsigned* mine = buf + ((size_t)(r & 1) * nb + me) * 1024;
        for (int i = threadIdx.x; i < 1024; i += blockDim.x) mine[i] = (unsigned)r * 0x10001u + me * 7u + i;
        if (!grid_barrier<MODE>(counter, (unsigned)r * nb)) { if (threadIdx.x == 0) atomicAdd(fail, 1u << 16); return; }
        const unsigned other = (me + 37u) % nb;                                     // 37: another XCD under round-robin dispatch
        const unsigned* theirs = buf + ((size_t)(r & 1) * nb + other) * 1024;
        for (int i = threadIdx.x; i < 1024; i += blockDim.x)
            if (theirs[i] != (unsigned)r * 0x10001u + other * 7u + i) ++bad;
    }
    if (bad) atomicAdd(fail, bad);
}

// MODE 2 barrier (no fences), but every exchanged word is written / read with an agent-scope relaxed ATOMIC store / load (sc1: served at
// the level all XCDs share): is that a correct and cheap hand-over?
__global__ __launch_bounds__(512) void exchange_atomic(unsigned* counter, int rounds, unsigned* buf, unsigned* fail) {
    const unsigned nb = gridDim.x, me = blockIdx.x;
    unsigned bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        unsigned* mine = buf + ((size_t)(r & 1) * nb + me) * 1024;
        for (int i = threadIdx.x; i < 1024; i += blockDim.x)
            __hip_atomic_store(mine + i, (unsigned)r * 0x10001u + me * 7u + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!grid_barrier<2>(counter, (unsigned)r * nb)) { if (threadIdx.x == 0) atomicAdd(fail, 1u << 16); return; }
        const unsigned other = (me + 37u) % nb;
        const unsigned* theirs = buf + ((size_t)(r & 1) * nb + other) * 1024;
        for (int i = threadIdx.x; i < 1024; i += blockDim.x)
            if (__hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)r * 0x10001u + other * 7u + i) ++bad;
    }
    if (bad) atomicAdd(fail, bad);
}

template <int MODE>
int run(const char* name, unsigned* counter, unsigned* fail, unsigned* buf, int nb) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; ++variant) {
        for (int rounds : {64, 256}) {
            CK(hipMemset(counter, 0, 4)); CK(hipMemset(fail, 0, 4));
            void* args_bare[] = {&counter, (void*)&rounds, &fail};
            void* args_ex[] = {&counter, (void*)&rounds, &buf, &fail};
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (variant == 0) CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(bare<MODE>), dim3(nb), dim3(512), args_bare, 0, 0));
            else CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(exchange<MODE>), dim3(nb), dim3(512), args_ex, 0, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
            printf("%-34s %-9s rounds=%4d  %8.1f us total  %6.2f us per barrier  failures=%u\n", name, variant ? "exchange" : "bare", rounds,
                   ms * 1e3, ms * 1e3 / rounds, f);
            fflush(stdout);
        }
    }
    return 0;
}

int main() {
    unsigned *counter, *fail, *buf, *ubuf = nullptr, *ucounter = nullptr;
    const int nb = 256;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&buf, (size_t)2 * nb * 1024 * 4));
    if (run<0>("fences, sleeping spin", counter, fail, buf, nb)) return 1;
    if (run<1>("fences, tight spin", counter, fail, buf, nb)) return 1;
    if (run<2>("no fences, cached buffer (UNSAFE)", counter, fail, buf, nb)) return 1;
    if (hipExtMallocWithFlags((void**)&ubuf, (size_t)2 * nb * 1024 * 4, hipDeviceMallocUncached) == hipSuccess &&
        hipExtMallocWithFlags((void**)&ucounter, 4, hipDeviceMallocUncached) == hipSuccess) {
        if (run<2>("no fences, UNCACHED buffer", ucounter, fail, ubuf, nb)) return 1;
    } else printf("hipDeviceMallocUncached not available\n");
    unsigned* fbuf = nullptr;
    if (hipExtMallocWithFlags((void**)&fbuf, (size_t)2 * nb * 1024 * 4, hipDeviceMallocFinegrained) == hipSuccess) {
        if (run<2>("no fences, FINE-GRAINED buffer", counter, fail, fbuf, nb)) return 1;
    } else printf("hipDeviceMallocFinegrained not available\n");
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rounds : {64, 256}) {
            CK(hipMemset(counter, 0, 4)); CK(hipMemset(fail, 0, 4));
            void* args[] = {&counter, (void*)&rounds, &buf, &fail};
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(exchange_atomic), dim3(nb), dim3(512), args, 0, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
            printf("%-34s %-9s rounds=%4d  %8.1f us total  %6.2f us per barrier  failures=%u\n", "no fences, ATOMIC word exchange", "exchange", rounds,
                   ms * 1e3, ms * 1e3 / rounds, f);
        }
    }
    printf("done\n");
    return 0;
}
