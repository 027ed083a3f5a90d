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


def _cpu_adamw(param, grad, m, v, step, lr, beta1, beta2, eps, weight_decay, grad_scale, shadow=None):
    """torch restatement of eavqa_adamw for the CPU tests (the product default, ops.adamw, has no CPU path)."""
    import oracle
    oracle.adamw_step(param, grad * grad_scale, m, v, step, lr, beta1, beta2, eps, weight_decay)
    if shadow is not None:
        shadow.copy_(param.to(shadow.dtype))


def _sharded_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      EAVQA_DIST_BACKEND="gloo")
    import oracle
    from eavqa_amd.models.clipcap import FlatParams
    from eavqa_amd.trainers.data_parallel import init_from_env
    from eavqa_amd.trainers.optim import ShardedAdamW
    init_from_env()
    out = {}
    for cdt in (torch.float32, torch.bfloat16):
        shapes = [("linear.weight", (24, 16)), ("linear.bias", (24,)), ("norm.weight", (16,)), ("attn.weight", (48, 16)), ("prefix_const", (5, 16))]
        flat = FlatParams(shapes, "cpu", cdt)
        g0 = torch.Generator().manual_seed(1)
        init = torch.randn(flat.numel, generator=g0)
        flat.master.copy_(init)
        if flat.shadow is not flat.master:
            flat.shadow.copy_(init.to(cdt))
        opt = ShardedAdamW(flat, lr=0.05, group=None, n_buckets=2, adamw=_cpu_adamw)
        assert opt.world == world and opt.n_buckets == 2 and flat.small_numel == 24 + 16
        # reference: plain AdamW on the cross-rank MEAN gradient, whole buffer
        ref_p, ref_m, ref_v = init.clone(), torch.zeros(flat.numel), torch.zeros(flat.numel)
        ok = True
        for step in range(1, 4):
            grads = [torch.randn(flat.numel, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world)]
            flat.grad.copy_(grads[rank])
            opt.start()
            opt.finish()
            oracle.adamw_step(ref_p, sum(grads) / world, ref_m, ref_v, step, 0.05)
            # every rank holds the whole updated operand copy; the fp32 master is exact on the owned shards and the small region
            tol = 1e-6 if cdt == torch.float32 else 1e-2
            ok = ok and bool((flat.shadow.float() - ref_p).abs().max().item() <= tol * max(1.0, ref_p.abs().max().item()))
            ok = ok and bool(torch.allclose(flat.master[:flat.small_numel], ref_p[:flat.small_numel], atol=1e-6))
            for b in range(opt.n_buckets):
                lo, hi, mine = opt._bucket(b)
                ok = ok and bool(torch.allclose(flat.master[mine:mine + opt.piece], ref_p[mine:mine + opt.piece], atol=1e-6))
        # a checkpoint taken between start() and finish(): gather_master() waits for the running update itself
        grads = [torch.randn(flat.numel, generator=torch.Generator().manual_seed(900 + r)) for r in range(world)]
        flat.grad.copy_(grads[rank])
        opt.start()
        full = opt.gather_master()
        oracle.adamw_step(ref_p, sum(grads) / world, ref_m, ref_v, 4, 0.05)
        ok = ok and bool(torch.allclose(full, ref_p, atol=1e-6)) and not opt._busy
        # the blocking fallback (synchronous=True / EAVQA_DP_SYNC=1) is the same arithmetic: bit-equal parameters
        f2 = FlatParams(shapes, "cpu", cdt)
        f2.master.copy_(init)
        if f2.shadow is not f2.master:
            f2.shadow.copy_(init.to(cdt))
        o2 = ShardedAdamW(f2, lr=0.05, group=None, n_buckets=2, adamw=_cpu_adamw, synchronous=True)
        for step in range(1, 4):
            grads = [torch.randn(flat.numel, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world)]
            f2.grad.copy_(grads[rank])
            o2.step()
        f2.grad.copy_([torch.randn(flat.numel, generator=torch.Generator().manual_seed(900 + r)) for r in range(world)][rank])
        o2.step()
        ok = ok and bool(torch.equal(o2.gather_master(), full)) and bool(torch.equal(f2.shadow, flat.shadow))
        # reduce-scatters issued DURING the backward (arm + the mapper's notify_grad reports, here replayed in backward order: the last
        # matrix first, then everything): bit-equal to the exchange after the backward; and bf16 gradient transport within its rounding
        f3, f4 = FlatParams(shapes, "cpu", cdt), FlatParams(shapes, "cpu", cdt)
        for f in (f3, f4):
            f.master.copy_(init)
            if f.shadow is not f.master:
                f.shadow.copy_(init.to(cdt))
        o3 = ShardedAdamW(f3, lr=0.05, group=None, n_buckets=2, adamw=_cpu_adamw)
        o4 = ShardedAdamW(f4, lr=0.05, group=None, n_buckets=2, adamw=_cpu_adamw, grad_transport=torch.bfloat16)
        issued_early = 0
        for step in list(range(1, 4)) + [9]:
            grads = [torch.randn(flat.numel, generator=torch.Generator().manual_seed((100 * step if step < 9 else 900) + r)) for r in range(world)]
            f3.grad.copy_(grads[rank]); f4.grad.copy_(grads[rank])
            o3.arm()
            lo, hi, _ = o3._bucket(o3.n_buckets - 1)
            f3.notify_grad(lo, hi)                               # "the last layer's gradients are final"
            issued_early += len(o3._rs)
            f3.notify_grad(0, f3.numel)
            o3.step(); o4.step()
        ok = ok and issued_early == 4 and bool(torch.equal(o3.gather_master(), full)) and bool(torch.equal(f3.shadow, flat.shadow))
        d16 = (o4.gather_master() - full).abs()
        # AdamW normalises the gradient: an entry whose summed gradient is within bf16 rounding of zero may take the other sign (one
        # update of size lr: measured max 1.0 lr on 0.1 % of the entries), every other entry moves as with the fp32 exchange
        ok = ok and 0 < d16.max().item() <= 0.05 * 4 * 1.01 and d16.mean().item() <= 0.05 * 0.02 and (d16 > 0.005).float().mean().item() <= 0.01
        out[str(cdt)] = ok
    results[rank] = out
    dist.destroy_process_group()


def test_sharded_adamw_two_ranks_gloo():
    """Reduce-scatter + sharded AdamW + all-gather (the exchange for the transformer mapper / very large mappers) gives, on every
    rank, the parameters of plain AdamW on the mean gradient."""
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_sharded_worker, args=(world, port, results), nprocs=world, join=True)
        res = dict(results)
    assert all(all(v.values()) for v in res.values()) and len(res) == world, res


def test_exchange_choice_rule():
    """gradient factors while the whole-batch weight gradient + full AdamW pass is cheaper than moving 6 B per
    parameter over one xGMI link; the sharded optimiser for mappers without per-sample factors."""
    from eavqa_amd.trainers.optim import choose_dp_exchange, dp_exchange_costs
    assert choose_dp_exchange(85_000_000, 85_000_000, 64, 1) == "none"
    assert choose_dp_exchange(85_000_000, 85_000_000, 64, 8) == "factors"               # cfg2 MLP mapper
    assert choose_dp_exchange(218_000_000, 218_000_000, 64, 8) == "factors"             # cfg3
    assert choose_dp_exchange(1_170_000_000, 0, 32, 8) == "sharded"                     # cfg5 transformer mapper: no factor path
    c = dp_exchange_costs(8_640_000_000, 8_640_000_000, 32, 8)
    assert c["factors"] < c["sharded"] < c["allreduce"]
    c7 = dp_exchange_costs(8_640_000_000, 8_640_000_000, 32, 8, links=7)
    assert c7["sharded"] < c7["factors"]                                                # a direct 7-link algorithm would flip it


def test_all_gather_rows_nccl_branch_shapes_and_order(monkeypatch):
    """The RCCL branch of all_gather_rows (one all_gather_into_tensor straight into the result) cannot run here (no GPU) nor with
    two ranks on one card (RCCL refuses duplicate devices): exercise it against a recording stand-in for the process group -
    output shape [world * rows, cols], rank-major order, a contiguous input."""
    from eavqa_amd.trainers import data_parallel as dp
    calls = {}

    def fake_all_gather_into_tensor(out, t, group=None):
        calls["out_shape"], calls["in_contig"] = tuple(out.shape), t.is_contiguous()
        world = out.shape[0] // t.shape[0]
        for r in range(world):                       # what ncclAllGather produces: rank r's rows at [r * rows, (r + 1) * rows)
            out[r * t.shape[0]:(r + 1) * t.shape[0]] = t + 100 * r
    monkeypatch.setattr(dp.dist, "get_world_size", lambda group=None: 4)
    monkeypatch.setattr(dp.dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(dp.dist, "all_gather_into_tensor", fake_all_gather_into_tensor)
    x = torch.arange(12.0).view(3, 4).t()            # non-contiguous [4, 3] view
    out = dp.all_gather_rows(x)
    assert tuple(out.shape) == (16, 3) and calls == {"out_shape": (16, 3), "in_contig": True}
    for r in range(4):
        assert torch.equal(out[4 * r:4 * r + 4], x + 100 * r)
