import sys, torch
sys.path.insert(0, '/root/repo')
from eavqa_amd import ops
def timed(fn, iters):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
print("M N K tiles256 big k64_256x128 k64_256x160 k64_128x256 auto")
for M in (2048, 3200, 4800, 8192, 16448, 41120):
    for N in (1024, 2560, 4096, 7680, 10240):
        for K in (1024, 2560, 4096, 10240):
            if M * N * K > 41120 * 4096 * 4096: continue
            a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
            b = (torch.randn(N, K, device='cuda') * 0.02).to(torch.bfloat16)
            out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
            iters = max(4, min(20, int(5e11 / (2.0 * M * N * K)) + 3))
            res = []
            for sel in ((2 << 14), (14 << 8), (15 << 8), (22 << 8), 0):
                ops.KernelSelect.gemm = sel
                res.append(timed(lambda: ops.gemm(a, b, out=out), iters))
            ops.KernelSelect.gemm = 0
            tiles = ((M + 255) // 256) * ((N + 255) // 256)
            print(M, N, K, tiles, " ".join(f"{t:.1f}" for t in res), flush=True)
