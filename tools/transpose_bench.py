import sys, torch
sys.path.insert(0, '/root/repo')
from eavqa_amd import ops
for rows, cols in ((2048, 4096), (2048, 8192), (12800, 6400), (1864, 1280)):
    x = torch.randn(rows, cols, device='cuda').to(torch.bfloat16)
    y = torch.empty(cols, rows, device='cuda', dtype=torch.bfloat16)
    for _ in range(3): ops.transpose(x, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.transpose(x, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"transpose [{rows},{cols}] bf16: {us:.1f} us  {2 * rows * cols * 2 / us / 1e6:.2f} TB/s")
