"""The data-parallel exchange code over RCCL on the one GPU of the test box: a process group of ONE rank with backend "nccl".

A one-GPU box cannot run two RCCL ranks (one rank per device), so these tests cannot show scaling; they show that the collectives the
N > 1 path issues - all_reduce on a side stream, all_gather_into_tensor, reduce_scatter_tensor, in-place bf16 all-gather - are accepted
by RCCL for the tensors this code hands over (device buffers, slices of the flat parameter buffer, bf16 shadows) and leave the same
numbers as the single-process optimiser.  The arithmetic across ranks is covered by the world_size-2 gloo tests (test_data_parallel.py).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def rccl_group_of_one():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        yield
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def test_exchange_collectives_run_on_rccl(rccl_group_of_one):
    from eavqa_amd.trainers.data_parallel import GradSync, all_gather_rows
    assert dist.get_backend() == "nccl"
    # factor exchange of the MLP mapper: one ncclAllGather straight into the result
    for dtype in (torch.bfloat16, torch.float32):
        t = torch.randn(5, 12, device=DEV).to(dtype)[:, :8]          # non-contiguous view, like a slice of a wider activation
        got = all_gather_rows(t)
        assert got.shape == (5, 8) and torch.equal(got, t)
    # flat-gradient all-reduce on the side stream
    buf = torch.randn(1 << 20, device=DEV)
    want = buf.clone()
    sync = GradSync(buf, collectives_in_group_of_one=True)
    assert sync.stream is not None and sync.grad_scale == 1.0
    for _ in range(2):
        sync.start()
        sync.finish()
    torch.cuda.synchronize()
    assert torch.equal(buf, want)                                      # the sum over one rank


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_sharded_adamw_over_rccl_equals_fused_adamw(rccl_group_of_one, cdt):
    """reduce_scatter_tensor -> eavqa_adamw on the shard -> in-place all_gather_into_tensor of the operand copy, bucket by bucket on
    the side stream, against the single-kernel optimiser on the same gradients: identical bits (same kernel, same elements)."""
    from eavqa_amd.models.clipcap import FlatParams
    from eavqa_amd.trainers.optim import FusedAdamW, ShardedAdamW
    shapes = [("linear.weight", (768, 512)), ("linear.bias", (768,)), ("norm.weight", (512,)), ("attn.weight", (1536, 512)),
              ("prefix_const", (10, 512))]
    a, b = FlatParams(shapes, DEV, cdt), FlatParams(shapes, DEV, cdt)
    init = torch.randn(a.numel, device=DEV)
    for fl in (a, b):
        fl.master.copy_(init)
        if fl.shadow is not fl.master:
            fl.shadow.copy_(init.to(cdt))
    sharded = ShardedAdamW(a, lr=0.01, n_buckets=3, collectives_in_group_of_one=True)
    fused = FusedAdamW(b, lr=0.01)
    assert sharded.multi and sharded.stream is not None
    for step in range(3):
        g = torch.randn(a.numel, device=DEV)
        a.grad.copy_(g)
        b.grad.copy_(g)
        sharded.start()
        sharded.finish()
        fused.step()
    torch.cuda.synchronize()
    assert torch.equal(a.master, b.master) and torch.equal(a.shadow, b.shadow)
    assert torch.equal(sharded.gather_master(), b.master)
