"""Data-parallel gradient exchange for mapper training: one process per GPU, RCCL over xGMI.

Samples are independent and both the ViT and the LM are frozen replicas, so the only exchange is
the mapper gradient - ONE flat float32 buffer, averaged across ranks (Lightning-DDP
semantics: per-rank mean loss, gradients averaged).  The collective runs on its own stream; the
caller overlaps it with the next batch's ViT encode, which does not depend on the mapper, and
applies AdamW afterwards (mapper gradients arrive last in backward, so there is nothing else to
hide behind).  The division by world size is folded into the fused AdamW (``grad_scale``).

``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """(rank, local_rank, world) from torchrun's environment; initialises the process group when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        backend = backend or os.environ.get("EAVQA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("EAVQA_FORCE_DEVICE", local)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def all_gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """[rows, cols] on every rank -> [world * rows, cols] in rank order (identical on every rank).  NCCL/RCCL gathers on
    the current stream; gloo (CPU tests, single-GPU rehearsals) has no GPU all_gather and is staged through the host."""
    world = dist.get_world_size(group)
    t = t.contiguous()
    if t.is_cuda and dist.get_backend(group) == "gloo":
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)]
        dist.all_gather(parts, t.cpu(), group=group)
        return torch.cat(parts, 0).to(t.device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, t, group=group)           # one ncclAllGather straight into the result
    else:
        dist.all_gather(list(out.chunk(world, 0)), t, group=group)
    return out


class GradSync:
    """Sum-all-reduce of a flat gradient buffer on a side stream (no-op for a single rank).  ``exchange=False``: the
    gradients in the buffer are already summed over the ranks (the MLP mapper's factor exchange, models/clipcap.py), only
    ``grad_scale`` is still needed."""

    def __init__(self, flat_grad: torch.Tensor, world: Optional[int] = None, group=None, exchange: bool = True,
                 collectives_in_group_of_one: bool = False):
        self.buf = flat_grad
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        # a group of one needs no exchange; `collectives_in_group_of_one` issues the collective anyway (the RCCL path on a one-GPU box:
        # tests/test_rccl_gpu.py)
        self.exchange = exchange and (self.world > 1 or collectives_in_group_of_one)
        self.on_gpu = flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if (self.on_gpu and self.exchange) else None
        self._pending = None
        self._ev = None                                  # (start, end) HIP events of the last exchange, on the exchange stream

    @property
    def grad_scale(self) -> float:
        """Factor that turns the summed gradient into the cross-rank mean."""
        return 1.0 / self.world

    def start(self) -> None:
        """Enqueue the all-reduce after everything already queued on the current stream."""
        if not self.exchange:
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
                work = dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if dist.get_backend(self.group) == "nccl":
                    # RCCL runs the collective on c10d's internal stream; wait() only makes THIS side stream wait for it (the
                    # host does not block), so the end marker brackets the collective's device time
                    work.wait()
                    e1.record(self.stream)
                    self._ev = (e0, e1)
                    work = None
                self._pending = work
                self._started = True
        else:
            self._pending = dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._started = True

    def finish(self) -> None:
        """Make the current stream wait for the exchange (call right before the optimiser step)."""
        if not getattr(self, "_started", False):
            return
        if self._pending is not None:
            self._pending.wait()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self._pending = None
        self._started = False

    def last_exchange_ms(self):
        """Device milliseconds of the last all-reduce (HIP events on the exchange stream; waits for the end event), or None."""
        if self._ev is None:
            return None
        self._ev[1].synchronize()
        return round(self._ev[0].elapsed_time(self._ev[1]), 3)
