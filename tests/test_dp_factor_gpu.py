"""GPU, two ranks sharing one card over gloo: the MLP mapper's gradient-factor exchange (all-gather dy / h / dh / x, weight
gradient of the global batch formed locally) gives the gradients of the all-reduce path, identically on every rank."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      EAVQA_DIST_BACKEND="gloo", EAVQA_FORCE_DEVICE="0")
    import torch.distributed as dist
    from eavqa_amd.models.clipcap import MLP
    from eavqa_amd.trainers.data_parallel import GradSync, init_from_env
    init_from_env()
    dev = "cuda:0"
    out = {}
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 3e-2)):
        torch.manual_seed(0)                                  # same parameters on both ranks
        mlp = MLP((32, 64, 128), device=dev, dtype=dtype)
        g = torch.Generator().manual_seed(10 + rank)          # different samples per rank
        x = torch.randn(8, 32, generator=g).to(dev)
        w = torch.randn(8, 128, generator=g).to(dev)

        def grads(factor):
            mlp.dp_factor_exchange = factor
            mlp.zero_grad(set_to_none=True)
            mlp.flat.grad_live = False
            y = mlp(x)
            (y.float() * w).sum().backward()
            sync = GradSync(mlp.flat.grad, world, exchange=not factor)
            sync.start(); sync.finish()
            torch.cuda.synchronize()
            return mlp.flat.grad.clone()

        g_ar, g_fx = grads(False), grads(True)
        scale = g_ar.abs().max().item()
        both = [torch.empty_like(g_fx).cpu() for _ in range(world)]
        dist.all_gather(both, g_fx.cpu())
        out[str(dtype)] = (bool((g_ar - g_fx).abs().max().item() <= tol * max(1.0, scale)), bool(torch.equal(both[0], both[1])),
                           bool(scale > 0))
    results[rank] = out
    dist.destroy_process_group()


def test_factor_exchange_matches_all_reduce_two_ranks():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
        res = dict(results)
    for rank in range(world):
        for k, v in res[rank].items():
            assert v == (True, True, True), (rank, k, v)
