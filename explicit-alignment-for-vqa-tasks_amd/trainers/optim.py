"""Fused AdamW over the mapper's flat parameter buffer.

``torch.optim.AdamW(params, lr=config.train.lr)`` with torch defaults (betas 0.9/0.999, eps 1e-8,
weight_decay 0.01) is what the reference configures (src/trainers/clipcap_exector.py:79-81); here the
whole update is one HBM-bound kernel (``eavqa_adamw``) over the contiguous master / grad / moment
buffers, which also refreshes the bf16 shadow used by the next forward.
"""
from __future__ import annotations

import torch
from typing import Optional

from .. import ops


class FusedAdamW:
    def __init__(self, flat, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        self.flat = flat
        self.param_groups = [dict(lr=lr, initial_lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.m = torch.zeros_like(flat.master)
        self.v = torch.zeros_like(flat.master)
        self.step_count = 0

    def step(self, grad_scale: float = 1.0, chunks=None) -> None:
        """``chunks`` (``mapper.update_chunks()``: a partition of the flat buffers in the order the forward reads them): the update runs
        chunk by chunk on the optimiser's own stream and leaves one event per chunk with the parameters (``FlatParams.wait_ready``), so the
        next forward starts on the first layer while the later layers are still being updated - AdamW is HBM-bound (30 B / parameter),
        the mapper's GEMMs are not.  Elementwise arithmetic: the result is bit-equal to the one-launch update."""
        g = self.param_groups[0]
        self.step_count += 1
        fl = self.flat
        lowp = fl.shadow is not fl.master
        kw = dict(step=self.step_count, lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"],
                  grad_scale=grad_scale)
        if chunks and fl.master.is_cuda:
            if sorted(chunks)[0][0] != 0 or sum(h - l for l, h in chunks) != fl.numel:
                raise ValueError("chunks must partition the flat parameter buffer")
            if getattr(self, "_stream", None) is None:
                self._stream = torch.cuda.Stream()
            fl.wait_ready()                                          # (a previous pipelined update nobody has waited for yet)
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                for lo, hi in chunks:
                    ops.adamw(fl.master[lo:hi], fl.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], shadow=fl.shadow[lo:hi] if lowp else None, **kw)
                    ev = torch.cuda.Event()
                    ev.record(self._stream)
                    fl._ready.append((lo, hi, ev))
        else:
            ops.adamw(fl.master, fl.grad, self.m, self.v, shadow=fl.shadow if lowp else None, **kw)
        fl.mark_shadow_fresh()

    def zero_grad(self, set_to_none: bool = True) -> None:
        """No memset: the next backward overwrites the flat gradient instead of accumulating."""
        self.flat.grad_live = False

    def state_dict(self):
        self.flat.wait_ready()
        return dict(step=self.step_count, m=self.m, v=self.v, param_groups=self.param_groups, layout=self.flat.layout_tag())

    def load_state_dict(self, sd) -> None:
        _check_layout(self.flat, sd, self.m)
        self.step_count = int(sd["step"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.param_groups = sd["param_groups"]


def _check_layout(flat, sd, m) -> None:
    """The moments are raw flat buffers: refuse a state written under another parameter layout (FlatParams orders 1-D parameters
    first and pads the matrix region; a state from an older layout would load misaligned where the sizes happen to match)."""
    tag = sd.get("layout")
    if tag is not None and tag != flat.layout_tag():
        raise ValueError("optimiser state was written under another flat parameter layout (layout tag mismatch)")
    if tag is None and tuple(sd["m"].shape) != tuple(m.shape):
        raise ValueError(f"optimiser state without a layout tag and of another size ({tuple(sd['m'].shape)} vs {tuple(m.shape)})")


class ConstantScheduleWithWarmup:
    """``get_constant_schedule_with_warmup`` (the reference's ``"scheduler": "none"`` branch,
    clipcap_exector.py:102-110): lr * min(1, step / warmup)."""

    def __init__(self, optimizer: FusedAdamW, num_warmup_steps: int = 0):
        self.opt, self.warmup, self.n = optimizer, num_warmup_steps, 0
        self._apply()

    def _factor(self) -> float:
        return 1.0 if self.warmup <= 0 or self.n >= self.warmup else float(self.n) / float(max(1, self.warmup))

    def _apply(self) -> None:
        for g in self.opt.param_groups:
            g["lr"] = g["initial_lr"] * self._factor()

    def step(self) -> None:
        self.n += 1
        self._apply()

    def resume(self, global_step: int) -> None:
        """Continue from ``global_step`` optimiser steps (the reference passes ``last_epoch=self.global_step`` when it
        builds the scheduler, clipcap_exector.py:96-124)."""
        self.n = int(global_step)
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.opt.param_groups]


class LinearScheduleWithWarmup(ConstantScheduleWithWarmup):
    """``get_linear_schedule_with_warmup`` (``"scheduler": "linear"``, clipcap_exector.py:83-92): linear warm-up to lr,
    then linear decay to 0 at ``num_training_steps``."""

    def __init__(self, optimizer: FusedAdamW, num_warmup_steps: int, num_training_steps: int):
        self.total = max(1, num_training_steps)
        super().__init__(optimizer, num_warmup_steps)

    def _factor(self) -> float:
        if self.n < self.warmup:
            return float(self.n) / float(max(1, self.warmup))
        return max(0.0, float(self.total - self.n) / float(max(1, self.total - self.warmup)))


class CosineAnnealing(ConstantScheduleWithWarmup):
    """``optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=epochs, eta_min=1e-5)`` (``"scheduler": "cosine"``,
    clipcap_exector.py:93-101), closed form."""

    def __init__(self, optimizer: FusedAdamW, t_max: int, eta_min: float = 1e-5):
        import math
        self._math, self.t_max, self.eta_min = math, max(1, t_max), eta_min
        super().__init__(optimizer, 0)

    def _apply(self) -> None:
        for g in self.opt.param_groups:
            base = g["initial_lr"]
            g["lr"] = self.eta_min + (base - self.eta_min) * (1 + self._math.cos(self._math.pi * self.n / self.t_max)) / 2


class ShardedAdamW:
    """Data-parallel mapper update with the optimiser SHARDED over the ranks (ZeRO-1 shaped for xGMI), for mappers whose gradient
    is too large to all-reduce and whose weight gradient is not a product of a few per-sample factors (the transformer mapper),
    or whose factor exchange would repeat too much work (the 8.64 B-parameter MLP mapper of BASELINE configs[4],
    src/models/clipcap.py:256-262).  Per step, on a side stream, bucket by bucket:

        reduce-scatter(sum) of the bucket's flat fp32 gradient        -> every rank holds 1/world of the summed gradient
        fused AdamW on that shard (fp32 master + moments of the shard) -> updated master shard + its compute-dtype copy
        all-gather of the compute-dtype shards                          -> every rank has the whole updated operand copy

    Bytes per rank and step: (world-1)/world x (4 + 2) B/parameter instead of the all-reduce's 2 x (world-1)/world x 4, optimiser
    traffic and moment memory 1/world; xGMI is point-to-point, so both collectives are per-link bound.  All
    collectives are issued asynchronously (``_run``): the reduce-scatters queue up front on c10d's collective stream, bucket b's
    AdamW runs on the side stream while bucket b+1's reduce-scatter is on the wire, each all-gather is queued as soon as its shard
    is updated.  The 1-D parameters (biases, LayerNorm affine: ``FlatParams.small_numel`` leading elements, read in fp32 by the
    kernels) stay replicated: one small asynchronous all-reduce issued first and a redundant update on every rank (nothing
    blocks on it).  UNMEASURED on more than one GPU (no multi-GPU box has run this tree): correct under gloo world size 2 and
    RCCL in a group of one.

    Layout of the matrix region [small, numel): ``n_buckets`` contiguous buckets of ``world x piece`` elements; rank r owns
    [r x piece, (r+1) x piece) of every bucket, so each collective works on one contiguous range (in place for the gather).
    ``adamw`` is the update kernel (``ops.adamw``; the CPU tests inject a torch restatement - the product path has no CPU fallback)."""

    def __init__(self, flat, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, *, group=None,
                 n_buckets: int = 4, adamw=None, collectives_in_group_of_one: bool = False, synchronous: Optional[bool] = None,
                 grad_transport: Optional[torch.dtype] = None):
        """``synchronous`` (default: the environment variable ``EAVQA_DP_SYNC=1``): the blocking order of round 2 - reduce-scatter,
        AdamW, all-gather bucket by bucket, every collective waited for before the next call - as a fallback should the
        asynchronous issue order (never run on more than one GPU) misbehave on a real RCCL group; same arithmetic, same result.
        ``grad_transport=torch.bfloat16``: the matrix gradients travel in bf16 (4 instead of 6 B / parameter over xGMI: the bucket is
        cast before its reduce-scatter, the summed shard cast back before AdamW); the sum itself is then rounded to 8 significant
        bits per hop - bounded in tests/test_data_parallel.py, off by default.
        ``arm()`` before a backward lets the reduce-scatters start DURING it: the mapper's backward reports every layer whose weight
        gradients are final (``FlatParams.notify_grad``) and each bucket those reports cover completely is put on the wire at once, in
        backward order; ``start()`` then issues what is left.  Same collectives on the same data: bit-equal to the exchange after
        ``backward()`` (the north_star's "all-reduce overlapped with backward")."""
        import os
        import torch.distributed as dist
        self.synchronous = (os.environ.get("EAVQA_DP_SYNC", "0") == "1") if synchronous is None else bool(synchronous)
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.param_groups = [dict(lr=lr, initial_lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.adamw = adamw if adamw is not None else ops.adamw
        self.small = flat.small_numel
        big = flat.numel - self.small
        while n_buckets > 1 and big % (n_buckets * self.world * flat.ALIGN):
            n_buckets -= 1
        if big % (n_buckets * self.world * flat.ALIGN):
            raise ValueError(f"matrix region of {big} elements does not cut into {self.world} aligned shards")
        self.n_buckets = n_buckets
        self.piece = big // (n_buckets * self.world)
        dev = flat.master.device
        self.m_small = torch.zeros(self.small, device=dev)
        self.v_small = torch.zeros(self.small, device=dev)
        self.m = torch.zeros(n_buckets * self.piece, device=dev)
        self.v = torch.zeros(n_buckets * self.piece, device=dev)
        self.gshard = torch.empty(n_buckets * self.piece, device=dev)
        if grad_transport not in (None, torch.float32, torch.bfloat16):
            raise ValueError("grad_transport must be None / float32 / bfloat16")
        self.transport = None if grad_transport in (None, torch.float32) else grad_transport
        if self.transport is not None:
            self.g16 = torch.empty(n_buckets * self.world * self.piece, device=dev, dtype=self.transport)
            self.gshard16 = torch.empty(n_buckets * self.piece, device=dev, dtype=self.transport)
        self._armed, self._listening, self._rs, self._covered = False, False, {}, []
        self.step_count = 0
        # a group of one needs no collectives; `collectives_in_group_of_one` issues them anyway (the RCCL path on a one-GPU box)
        self.multi = self.world > 1 or (collectives_in_group_of_one and dist.is_initialized())
        self.stream = torch.cuda.Stream() if (flat.master.is_cuda and self.multi) else None
        self._busy = False
        self._master_stale = False
        self._ev = None                                  # (start, end) HIP events of the last exchange + update, on the side stream

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def _bucket(self, b: int):
        lo = self.small + b * self.world * self.piece
        return lo, lo + self.world * self.piece, lo + self.rank * self.piece

    # ---- reduce-scatter of one bucket (asynchronous); `early`: from inside the backward, behind what the current stream holds
    def _issue_rs(self, b: int, early: bool = False) -> None:
        import torch.distributed as dist
        fl = self.flat
        lo, hi, _ = self._bucket(b)

        def go():
            src, dst = fl.grad[lo:hi], self.gshard[b * self.piece:(b + 1) * self.piece]
            if self.transport is not None:
                src = self.g16[lo - self.small:hi - self.small]
                if fl.grad.is_cuda:
                    ops.cast_rows(fl.grad[lo:hi].view(1, hi - lo), self.transport, out=src.view(1, hi - lo))
                else:
                    src.copy_(fl.grad[lo:hi])
                dst = self.gshard16[b * self.piece:(b + 1) * self.piece]
            self._rs[b] = dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        if early and self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                if not self._rs:
                    self._e0_early = torch.cuda.Event(enable_timing=True)
                    self._e0_early.record(self.stream)
                go()
        else:
            go()

    def arm(self) -> None:
        """Call before the backward whose gradients the next ``start()`` exchanges (the LAST micro-batch of an accumulation window)."""
        if self.synchronous or not self.multi:
            return
        if not self._listening:
            self.flat.grad_listeners.append(self._on_grad)
            self._listening = True
        self._armed, self._rs, self._covered, self._e0_early = True, {}, [0] * self.n_buckets, None

    def _on_grad(self, lo: int, hi: int) -> None:
        if not self._armed:
            return
        for b in reversed(range(self.n_buckets)):                 # the backward completes the flat buffer from its end
            if b in self._rs:
                continue
            blo, bhi, _ = self._bucket(b)
            if lo <= blo and hi >= bhi:
                self._covered[b] = bhi - blo
            else:
                self._covered[b] += max(0, min(hi, bhi) - max(lo, blo))
            if self._covered[b] >= bhi - blo:
                self._issue_rs(b, early=True)

    def _run(self, grad_scale: float) -> None:
        import torch.distributed as dist
        fl, g = self.flat, self.param_groups[0]
        self.step_count += 1
        kw = dict(step=self.step_count, lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"],
                  grad_scale=grad_scale)
        lowp = fl.shadow is not fl.master
        multi = self.multi
        # Every collective is ASYNC: c10d runs them in issue order on its own stream, so all reduce-scatters (and, first, the small
        # all-reduce of the replicated 1-D parameters) are queued up front; bucket b's AdamW waits only for ITS reduce-scatter and
        # runs on this stream while bucket b+1's reduce-scatter is on the wire; its all-gather is queued as soon as the shard is
        # updated.  Nothing blocks the host (RCCL work.wait() is a stream wait); gloo (CPU tests) blocks in wait(), same order.
        w_small = None
        if self.small and multi:
            w_small = dist.all_reduce(fl.grad[:self.small], op=dist.ReduceOp.SUM, group=self.group, async_op=not self.synchronous)
        self._armed = False
        if multi and self.synchronous:
            for b in range(self.n_buckets):
                lo, hi, mine = self._bucket(b)
                gs = self.gshard[b * self.piece:(b + 1) * self.piece]
                dist.reduce_scatter_tensor(gs, fl.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
                self.adamw(fl.master[mine:mine + self.piece], gs, self.m[b * self.piece:(b + 1) * self.piece],
                           self.v[b * self.piece:(b + 1) * self.piece], shadow=fl.shadow[mine:mine + self.piece] if lowp else None, **kw)
                dist.all_gather_into_tensor(fl.shadow[lo:hi], fl.shadow[mine:mine + self.piece], group=self.group)
            if self.small:
                self.adamw(fl.master[:self.small], fl.grad[:self.small], self.m_small, self.v_small, shadow=fl.shadow[:self.small] if lowp else None, **kw)
            self._master_stale = lowp
            fl.mark_shadow_fresh()
            return
        if multi:
            for b in range(self.n_buckets):
                if b not in self._rs:                             # (armed: some - or all - are already on the wire since the backward)
                    self._issue_rs(b)
        ag = []
        for b in range(self.n_buckets):
            lo, hi, mine = self._bucket(b)
            if multi:
                self._rs[b].wait()
                gs = self.gshard[b * self.piece:(b + 1) * self.piece]
                if self.transport is not None:
                    gs.copy_(self.gshard16[b * self.piece:(b + 1) * self.piece])      # dtype-converting copy of 1 / world of the bucket
            else:
                gs = fl.grad[lo:hi]
            self.adamw(fl.master[mine:mine + self.piece], gs, self.m[b * self.piece:(b + 1) * self.piece], self.v[b * self.piece:(b + 1) * self.piece],
                       shadow=fl.shadow[mine:mine + self.piece] if lowp else None, **kw)
            if multi:
                # in place: rank r's input is its own slice of the output
                ag.append(dist.all_gather_into_tensor(fl.shadow[lo:hi], fl.shadow[mine:mine + self.piece], group=self.group, async_op=True))
        if self.small:
            if w_small is not None:
                w_small.wait()
            self.adamw(fl.master[:self.small], fl.grad[:self.small], self.m_small, self.v_small, shadow=fl.shadow[:self.small] if lowp else None, **kw)
        for w in ag:
            w.wait()
        self._rs = {}
        self._master_stale = multi and lowp          # non-owned master shards are not updated in bf16-operand mode: gather_master()
        fl.mark_shadow_fresh()

    def start(self, grad_scale: float = None) -> None:
        """Enqueue exchange + update behind everything queued on the current stream (call right after backward)."""
        scale = self.grad_scale if grad_scale is None else grad_scale
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
                early = getattr(self, "_e0_early", None) if self._rs else None
                self._run(scale)
                e1.record(self.stream)
                self._ev = (early or e0, e1)                      # from the first reduce-scatter (possibly issued inside the backward)
        else:
            self._run(scale)
        self._busy = True

    def finish(self) -> None:
        """Make the current stream wait for the updated operand copy (call before the next mapper forward)."""
        if self._busy and self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self._busy = False

    def last_exchange_ms(self):
        """Device milliseconds of the last exchange + sharded update (HIP events on the side stream), or None."""
        if self._ev is None:
            return None
        self._ev[1].synchronize()
        return round(self._ev[0].elapsed_time(self._ev[1]), 3)

    def step(self, grad_scale: float = None) -> None:
        """Optimiser-style entry: exchange + update + wait."""
        self.start(grad_scale)
        self.finish()

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.flat.grad_live = False

    def gather_master(self) -> torch.Tensor:
        """The whole fp32 master copy on every rank (checkpoints): non-owned shards are stale between steps in bf16 mode.
        COLLECTIVE: every rank of the group must call it (one all-gather per bucket) - a rank-0-only checkpoint would hang the
        others' next collective.  An update still running on the side stream (``start()`` without ``finish()``) is waited for first:
        the gathers are queued on the current stream and would otherwise read half-updated shards."""
        import torch.distributed as dist
        fl = self.flat
        self.finish()
        if self.multi and fl.shadow is not fl.master:
            for b in range(self.n_buckets):
                lo, hi, mine = self._bucket(b)
                dist.all_gather_into_tensor(fl.master[lo:hi], fl.master[mine:mine + self.piece], group=self.group)
        self._master_stale = False
        return fl.master

    def state_dict(self):
        return dict(step=self.step_count, m=self.m, v=self.v, m_small=self.m_small, v_small=self.v_small, param_groups=self.param_groups,
                    world=self.world, rank=self.rank, n_buckets=self.n_buckets, layout=self.flat.layout_tag())

    def load_state_dict(self, sd) -> None:
        if (sd["world"], sd["rank"], sd["n_buckets"]) != (self.world, self.rank, self.n_buckets):
            raise ValueError("sharded optimiser state belongs to another world size / rank / bucket count")
        _check_layout(self.flat, sd, self.m)
        self.step_count = int(sd["step"])
        for name in ("m", "v", "m_small", "v_small"):
            getattr(self, name).copy_(sd[name])
        self.param_groups = sd["param_groups"]


# ----------------------------------------------------------------------------------------------------------------- exchange choice
# Measured / documented rates behind the rule below: the mapper wgrad GEMM at K = world x B runs at ~0.4
# PFLOP/s (tools/gemm_bench.py, transposed-factor path), fused AdamW streams 30 B/parameter at ~4.4 TB/s, and a ring collective
# over xGMI is bound by ONE link (~153 GB/s; 0.7 achieved assumed until a multi-GPU box measures it).
WGRAD_FLOPS, ADAMW_BW, LINK_BW = 0.4e15, 4.4e12, 0.7 * 153e9


def dp_exchange_costs(n_params: int, n_factor_params: int, per_gpu_batch: int, world: int, links: int = 1):
    """Modelled seconds per step of the three exchanges for a mapper of ``n_params`` parameters, ``n_factor_params`` of them in
    Linear layers whose weight gradient is an outer product of per-sample factors (all of an MLP mapper, none of a transformer
    mapper's attention / LayerNorm - its Linear layers see L tokens per sample, so their factors are L times larger):
      factors   all-gather of the factors (negligible) + the weight gradient of the GLOBAL batch on every rank
                (world x the local wgrad FLOPs) + the full AdamW pass on every rank
      sharded   reduce-scatter fp32 + all-gather bf16 over `links` links + 1/world of the AdamW pass
      allreduce ring all-reduce of the fp32 gradient + the full AdamW pass"""
    w = max(world, 1)
    frac = (w - 1) / w
    adam = 30.0 * n_params / ADAMW_BW
    costs = {"allreduce": 2 * frac * 4.0 * n_params / (links * LINK_BW) + adam,
             "sharded": frac * 6.0 * n_params / (links * LINK_BW) + adam / w}
    if n_factor_params == n_params:
        costs["factors"] = (w - 1) * 2.0 * per_gpu_batch * n_factor_params / WGRAD_FLOPS + adam
    return costs


def choose_dp_exchange(n_params: int, n_factor_params: int, per_gpu_batch: int, world: int, links: int = 1) -> str:
    """The cheapest exchange under :func:`dp_exchange_costs` ("factors" | "sharded" | "allreduce"); ``world == 1`` -> "none"."""
    if world <= 1:
        return "none"
    costs = dp_exchange_costs(n_params, n_factor_params, per_gpu_batch, world, links)
    return min(costs, key=costs.get)
