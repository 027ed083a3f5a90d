"""CPU, world_size 2 over gloo: the data-parallel gradient exchange used for mapper training
(eavqa_amd.trainers.data_parallel).  The N > 1 bench path is the same code over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      EAVQA_DIST_BACKEND="gloo")
    from eavqa_amd.trainers.data_parallel import GradSync, init_from_env
    import oracle
    r, local, w = init_from_env()
    assert (r, w) == (rank, world)
    # each rank: gradient of its own mean loss on its own shard (a tiny MLP mapper through the oracle)
    torch.manual_seed(0)
    D, H, O, B = 8, 16, 12, 6
    params = {"model.0.weight": torch.randn(H, D), "model.0.bias": torch.randn(H), "model.2.weight": torch.randn(O, H),
              "model.2.bias": torch.randn(O)}
    flat = torch.cat([p.flatten() for p in params.values()])
    g = torch.Generator().manual_seed(100)
    x_all, y_all = torch.randn(world * B, D, generator=g), torch.randn(world * B, O, generator=g)

    def grad_of(xs, ys):
        ps = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        loss = ((oracle.mlp_mapper(xs, ps) - ys) ** 2).mean()
        loss.backward()
        return torch.cat([p.grad.flatten() for p in ps.values()])

    local_grad = grad_of(x_all[rank * B:(rank + 1) * B], y_all[rank * B:(rank + 1) * B])
    buf = local_grad.clone()
    sync = GradSync(buf, world)
    sync.start()
    sync.finish()
    mean_grad = buf * sync.grad_scale                       # what the fused AdamW consumes (grad_scale folded in)
    # equal shard sizes: the cross-rank mean of per-rank mean-loss gradients == gradient on the 2x batch
    full = grad_of(x_all, y_all)
    ok = torch.allclose(mean_grad, full, atol=1e-6)
    # a second exchange on the same buffer (next step) is independent of the first
    buf.copy_(local_grad * 2)
    sync.start(); sync.finish()
    ok2 = torch.allclose(buf * sync.grad_scale, 2 * full, atol=1e-6)
    gathered = [torch.zeros_like(mean_grad) for _ in range(world)]
    dist.all_gather(gathered, mean_grad)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    # factor exchange (models/clipcap.py _MLPFunction.backward): all-gathering the per-sample factors and forming the
    # weight gradient of the global batch equals the all-reduced sum of the per-rank weight gradients
    from eavqa_amd.trainers.data_parallel import all_gather_rows
    gg = torch.Generator().manual_seed(7 + rank)
    dy, h = torch.randn(B, O, generator=gg), torch.randn(B, H, generator=gg)
    dy_all, h_all = all_gather_rows(dy), all_gather_rows(h)
    assert dy_all.shape == (world * B, O) and torch.equal(dy_all[rank * B:(rank + 1) * B], dy)
    local = dy.T @ h
    dist.all_reduce(local)
    ok3 = torch.allclose(dy_all.T @ h_all, local, atol=1e-5)
    nosync = GradSync(buf, world, exchange=False)
    before = buf.clone()
    nosync.start(); nosync.finish()
    ok3 = ok3 and torch.equal(buf, before) and nosync.grad_scale == 1.0 / world
    results[rank] = (bool(ok), bool(ok2), bool(same and ok3))
    dist.destroy_process_group()


def test_gradsync_two_ranks_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
        assert dict(results) == {0: (True, True, True), 1: (True, True, True)}


def test_gradsync_single_rank_is_a_noop():
    from eavqa_amd.trainers.data_parallel import GradSync
    buf = torch.arange(5.0)
    s = GradSync(buf, 1)
    s.start(); s.finish()
    assert s.grad_scale == 1.0 and torch.equal(buf, torch.arange(5.0))
