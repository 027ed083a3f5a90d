"""Drop-in counterparts of the reference's ``src/models/clipcap.py`` classes, executed by HIP kernels.

Same names, constructor keywords, ``forward`` / ``generate`` signatures and state-dict keys as the
reference (``MLP`` :31-42, ``TransformerMapper`` :213-237, ``ClipCaptionModel`` :240-471,
``ClipCaptionPrefix`` :590-599), so ``config.model_config.ModelClass`` / ``model_args`` select them
unchanged (src/trainers/clipcap_exector.py:52-53).  Differences, all deliberate:

* the LM may be GPT-2 *or* OPT (the reference hard-wires ``GPT2LMHeadModel``; BASELINE configs
  3-5 need OPT) and is held as a pre-packed :class:`FrozenCausalLM`, exposed as ``model.gpt``;
* masks/labels are created on the inputs' device (the reference uses a module-global ``device``
  that is always ``cuda:0``, clipcap.py:23,303-306);
* ``generate`` keeps a KV cache by default (``use_cache=False`` restores the reference's
  full re-forward per token, clipcap.py:414-419); the emitted ids follow the same rules.

The mapper is the only trainable part.  Its parameters live in one flat float32 buffer (with a
flat gradient buffer and, in bf16 mode, a flat bf16 shadow used as GEMM operand) so that the fused
AdamW and the data-parallel all-reduce each touch one contiguous range.
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from .lm import KNOWN_CONFIGS, FrozenCausalLM, LMConfig, load_local_hf, random_init_state_dict, synthetic_weights_notice

Tensor = torch.Tensor


# =========================================================================== flat parameter store
class FlatParams:
    """Owns flat master / grad / shadow buffers; module parameters are views into ``master``."""

    ALIGN = 8  # elements: keeps every view 16-byte aligned in bf16 and 32-byte aligned in fp32
    SHARD_ALIGN = 4096   # the matrix region is a multiple of this: it cuts evenly into world x buckets shards of aligned length

    def __init__(self, named_shapes: List[Tuple[str, Tuple[int, ...]]], device, compute_dtype: torch.dtype):
        """Layout: first every 1-D parameter (biases, LayerNorm affine: read in float32 by the kernels, ``f``), then the
        matrices (read through the compute-dtype shadow, ``w``).  ``small_numel`` is the length of the first region: the sharded
        optimiser (trainers/optim.py ``ShardedAdamW``) keeps that region replicated and shards only the matrices."""
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        ordered = [ns for ns in named_shapes if len(ns[1]) <= 1] + [ns for ns in named_shapes if len(ns[1]) > 1]
        self.small_numel = 0
        seen_matrix = False          # (a boolean, not `small_numel == 0`: a mapper without 1-D parameters has an EMPTY small region)
        for name, shape in ordered:
            if len(shape) > 1 and not seen_matrix:
                self.small_numel, seen_matrix = off, True
            n = int(math.prod(shape))
            self.offsets[name] = (off, tuple(shape))
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        if not seen_matrix:
            self.small_numel = off   # 1-D parameters only
        big = off - self.small_numel
        off = self.small_numel + (big + self.SHARD_ALIGN - 1) // self.SHARD_ALIGN * self.SHARD_ALIGN
        self.numel = off
        self.compute_dtype = compute_dtype
        self.master = torch.zeros(off, device=device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=device, dtype=torch.float32)
        self.shadow = self.master if compute_dtype == torch.float32 else torch.zeros(off, device=device, dtype=compute_dtype)
        self._shadow_version = -1
        self.grad_live = False   # False: the next backward overwrites ``grad`` instead of accumulating
        # pipelined optimiser (FusedAdamW.step(chunks=...)): (lo, hi, event) of parameter ranges whose update is still running on the
        # optimiser's stream - the mapper's forward waits for the ranges it is about to read (``wait_ready``)
        self._ready: List[Tuple[int, int, object]] = []
        # data parallel (ShardedAdamW.arm): callables (lo, hi) told by the mapper's backward that grad[lo:hi] is final for this step
        self.grad_listeners: List = []

    def range_of(self, *names: str) -> Tuple[int, int]:
        """[lo, hi) of the flat buffers spanned by the named parameters (contiguous in the layout when they belong to one layer)."""
        los = [self.offsets[n][0] for n in names]
        his = [self.offsets[n][0] + int(math.prod(self.offsets[n][1])) for n in names]
        return min(los), max(his)

    def wait_ready(self, lo: int = 0, hi: Optional[int] = None) -> None:
        """Make the current stream wait for every pending update that overlaps [lo, hi) (no-op without a pipelined optimiser)."""
        if not self._ready:
            return
        hi = self.numel if hi is None else hi
        keep = []
        for l, h, ev in self._ready:
            if l < hi and lo < h:
                torch.cuda.current_stream().wait_event(ev)
            else:
                keep.append((l, h, ev))
        self._ready = keep

    def notify_grad(self, lo: int, hi: int) -> None:
        for cb in self.grad_listeners:
            cb(lo, hi)

    def layout_tag(self) -> str:
        """Digest of the flat layout (names, offsets, shapes, total length): optimiser moments are raw flat buffers, so a checkpoint
        written under another layout must be refused rather than loaded misaligned."""
        import hashlib
        h = hashlib.sha256(repr((sorted(self.offsets.items()), self.numel, self.small_numel)).encode())
        return h.hexdigest()[:16]

    def view(self, buf: Tensor, name: str) -> Tensor:
        off, shape = self.offsets[name]
        return buf[off:off + int(math.prod(shape))].view(shape)

    def w(self, name: str) -> Tensor:
        """Parameter as GEMM operand (compute dtype)."""
        return self.view(self.shadow, name)

    def f(self, name: str) -> Tensor:
        """Parameter in float32 (biases, LayerNorm affine)."""
        return self.view(self.master, name)

    def g(self, name: str) -> Tensor:
        return self.view(self.grad, name)

    def refresh_shadow(self) -> None:
        """Re-cast the shadow when something other than the fused AdamW touched the master copy
        (torch optimisers, load_state_dict): tracked through the tensor version counter."""
        if self.shadow is self.master:
            return
        v = self.master._version
        if v != self._shadow_version:
            n = self.numel
            ops.cast_rows(self.master.view(1, n), self.compute_dtype, out=self.shadow.view(1, n))
            self._shadow_version = v

    def mark_shadow_fresh(self) -> None:
        self._shadow_version = self.master._version


# =========================================================================== mappers
class _MapperBase(nn.Module):
    """Common plumbing: parameters are registered as views of one :class:`FlatParams`."""

    def _adopt(self, flat: FlatParams, prefix: str, mods: Dict[str, nn.Parameter]) -> None:
        self.flat = flat
        self._names = []
        for name, p in mods.items():
            full = prefix + name
            v = flat.view(flat.master, full)
            v.copy_(p.data.to(v.device))
            p.data = v
            p.grad = None
            self._names.append((full, p))

    def attach_grads(self) -> None:
        for full, p in self._names:
            if p.grad is None:
                p.grad = self.flat.g(full)

    def grads_were_reset(self) -> bool:
        """True when the caller dropped ``.grad`` (zero_grad(set_to_none=True)): overwrite, don't accumulate."""
        return any(p.grad is None for _, p in self._names)


class MLP(_MapperBase):
    """``MLP`` clipcap.py:31-42 built as at :256-262: Linear(D, E*L/2) -> Tanh -> Linear(E*L/2, E*L).
    State-dict keys ``model.0.weight/bias``, ``model.2.weight/bias`` as ``nn.Sequential`` gives them."""

    def __init__(self, sizes: Tuple[int, ...], bias: bool = True, act=nn.Tanh, *, device="cuda", dtype=torch.bfloat16):
        super().__init__()
        if len(sizes) != 3 or not bias or act is not nn.Tanh:
            raise NotImplementedError("the hot path builds MLP((D, E*L//2, E*L)) with bias and Tanh only")
        layers = [nn.Linear(sizes[0], sizes[1]), nn.Tanh(), nn.Linear(sizes[1], sizes[2])]   # default init, :39
        self.model = nn.Sequential(*layers)
        self.sizes = tuple(sizes)
        named = [(n, tuple(p.shape)) for n, p in self.model.named_parameters(prefix="model")]
        flat = FlatParams(named, device, dtype)
        self._adopt(flat, "", {n: p for n, p in self.model.named_parameters(prefix="model")})
        self.dtype = dtype
        self._w2t = None      # scratch: transposed bf16 copy of the second Linear for its dgrad
        # data parallel: a process group here switches the weight gradients to the factor exchange (see _MLPFunction)
        self.dp_group = None
        self.dp_factor_exchange = False

    def update_chunks(self) -> List[Tuple[int, int]]:
        """Ranges of the flat buffers in the order the forward reads them (``FusedAdamW.step(chunks=...)``)."""
        fl = self.flat
        a, b = fl.range_of("model.0.weight"), fl.range_of("model.2.weight")
        return _cover(fl, [(0, fl.small_numel), a, b])

    def forward(self, x: Tensor) -> Tensor:
        """``x``: [..., D] float -> [..., E*L] in the compute dtype (differentiable w.r.t. the parameters)."""
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        self.flat.refresh_shadow()
        y = _MLPFunction.apply(x2, self, *[p for _, p in self._names])
        return y.view(*lead, self.sizes[2])


class _MLPFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod: "MLP", *params):
        fl, T = mod.flat, mod.dtype
        xT = _to_compute(x, T)
        u = torch.empty((xT.shape[0], mod.sizes[1]), device=xT.device, dtype=T)
        fl.wait_ready(0, fl.small_numel)
        fl.wait_ready(*fl.range_of("model.0.weight"))
        h = ops.gemm(xT, fl.w("model.0.weight"), bias=fl.f("model.0.bias"), act="tanh", aux_out=u)
        fl.wait_ready()
        y = ops.gemm(h, fl.w("model.2.weight"), bias=fl.f("model.2.bias"))
        ctx.mod = mod
        ctx.save_for_backward(xT, u, h)
        return y

    @staticmethod
    def backward(ctx, dy):
        mod: MLP = ctx.mod
        fl = mod.flat
        xT, u, h = ctx.saved_tensors
        dy = dy.contiguous()
        acc = fl.grad_live and not mod.grads_were_reset()
        # Data parallel, factor exchange: dW = dy^T h is a sum over samples, so instead of all-reducing the 82 M-element dW2
        # (340 MB for cfg2) every rank all-gathers the per-sample factors - dy [B, E L] and h [B, E L / 2], a few MB - and
        # computes the weight gradient of the GLOBAL batch itself (K = world x B instead of B).  Same sum, every rank gets
        # bitwise the same result, no gradient all-reduce is left.
        gather = None
        if mod.dp_factor_exchange:
            from ..trainers.data_parallel import all_gather_rows
            gather = lambda t: all_gather_rows(t, mod.dp_group)
        dy_w, h_w = (gather(dy), gather(h)) if gather else (dy, h)
        # layer 2: dW2[N,K] = dy^T h ; db2 = colsum(dy) ; dh = (dy W2) * tanh'(u)
        _wgrad(dy_w, h_w, fl.g("model.2.weight"), acc)
        ops.colsum(dy_w, fl.g("model.2.bias"), acc)
        fl.notify_grad(*fl.range_of("model.2.weight"))
        # dh = (dy W2) * tanh'(u): W2 is [N=E*L, K=H] (k-contiguous for the forward); its dgrad sums over N, so stream a
        # transposed copy (one HBM pass) through the k-contiguous kernels instead of transposing tile by tile in LDS
        w2 = fl.w("model.2.weight")
        if dy.shape[0] <= 64 and dy.dtype == torch.bfloat16 and w2.shape[0] % 32 == 0:
            if mod._w2t is None:
                mod._w2t = torch.empty((w2.shape[1], w2.shape[0]), device=w2.device, dtype=w2.dtype)
            ops.transpose(w2, out=mod._w2t)
            dh = ops.gemm(dy, mod._w2t, act="tanh", aux_in=u)
        else:
            dh = ops.gemm(dy, w2, b_kc=False, act="tanh", aux_in=u)
        dh_w, x_w = (gather(dh), gather(xT)) if gather else (dh, xT)
        _wgrad(dh_w, x_w, fl.g("model.0.weight"), acc)
        ops.colsum(dh_w, fl.g("model.0.bias"), acc)
        fl.notify_grad(0, fl.numel)
        fl.grad_live = True
        mod.attach_grads()
        return (None, None) + (None,) * len(mod._names)


def _cover(fl: FlatParams, ranges: List[Tuple[int, int]]) -> List[Tuple[int, int]]:
    """``ranges`` (in use order) completed to a partition of [0, numel): padding gaps and anything not named go to the end."""
    ranges = [r for r in ranges if r[1] > r[0]]
    covered = sorted(ranges)
    rest, at = [], 0
    for lo, hi in covered:
        if lo > at:
            rest.append((at, lo))
        at = max(at, hi)
    if at < fl.numel:
        rest.append((at, fl.numel))
    return ranges + rest


def _to_compute(x: Tensor, T: torch.dtype) -> Tensor:
    x = x.contiguous()
    if x.dtype == T:
        return x
    if x.dtype == torch.float32:
        return ops.cast_rows(x, T)
    return x.to(T)   # e.g. fp64 input: host-side convenience only


def _wgrad(dy: Tensor, x: Tensor, gview: Tensor, accumulate: bool) -> None:
    """gview[N,K] (+)= dy[M,N]^T @ x[M,K]  - both operands transposed in memory (m contiguous dim is not k).  In bf16 the
    two (small) factors are transposed first so that the product runs on the k-contiguous LDS-DMA kernels: 162 vs 621 us for
    the MLP mapper's dW2 at M = 512 (eight ranks' gathered factors), 88 vs 115 us at M = 64."""
    if dy.dtype == torch.bfloat16 and dy.shape[0] % 32 == 0 and dy.is_contiguous() and x.is_contiguous():
        ops.gemm(ops.transpose(dy), ops.transpose(x), out=gview, residual=gview if accumulate else None)
    else:
        ops.gemm(dy, x, a_kc=False, b_kc=False, out=gview, residual=gview if accumulate else None)


def _dgrad(x: Tensor, w: Tensor, **kw) -> Tensor:
    """``x @ w`` for a trainable weight ``w [N, K]`` (the dgrad of ``y = a @ w^T``): the contraction runs over w's ROWS, so the
    weight is first transposed into a k-contiguous operand (one HBM pass, ~1 ms for the 1.07 B-parameter mapper of cfg5) and the
    product takes the LDS-DMA kernels instead of the register-staged transposing one (3-4x slower at E = 4096)."""
    if x.dtype == torch.bfloat16 and w.shape[0] % 64 == 0 and w.shape[1] % 8 == 0 and w.is_contiguous():
        return ops.gemm(x, ops.transpose(w), **kw)
    return ops.gemm(x, w, b_kc=False, **kw)


class TransformerMapper(_MapperBase):
    """``TransformerMapper`` clipcap.py:213-237: ``linear`` (D -> clip_length*E), learned ``prefix_const``
    [L,E], then ``Transformer(dim, 8 heads, num_layers)`` (:141-210) of pre-LN layers (:107-138) with
    bias-free q / kv projections (:76-77), mlp_ratio 2.0 and ReLU; output rows ``[:, clip_length:]``.
    State-dict keys match the reference module tree."""

    HEADS = 8  # clipcap.py:233

    def __init__(self, dim_clip: int, dim_embedding: int, prefix_length: int, clip_length: int, num_layers: int = 8, *,
                 device="cuda", dtype=torch.bfloat16):
        super().__init__()
        E = dim_embedding
        self.clip_length, self.prefix_length, self.E, self.num_layers = clip_length, prefix_length, E, num_layers
        # build the same module tree as the reference so that default init + key names agree
        tr = nn.Module()
        tr.layers = nn.ModuleList()
        for _ in range(num_layers):
            lay = nn.Module()
            lay.norm1 = nn.LayerNorm(E)
            att = nn.Module()
            att.to_queries = nn.Linear(E, E, bias=False)
            att.to_keys_values = nn.Linear(E, 2 * E, bias=False)
            att.project = nn.Linear(E, E)
            lay.attn = att
            lay.norm2 = nn.LayerNorm(E)
            mlp = nn.Module()
            mlp.fc1 = nn.Linear(E, int(E * 2.0))
            mlp.fc2 = nn.Linear(int(E * 2.0), E)
            lay.mlp = mlp
            tr.layers.append(lay)
        self.transformer = tr
        self.linear = nn.Linear(dim_clip, clip_length * E)
        self.prefix_const = nn.Parameter(torch.randn(prefix_length, E), requires_grad=True)
        named = [(n, tuple(p.shape)) for n, p in self.named_parameters()]
        flat = FlatParams(named, device, dtype)
        self._adopt(flat, "", dict(self.named_parameters()))
        self.dtype = dtype

    def layer_range(self, i: int) -> Tuple[int, int]:
        """[lo, hi) of layer i's matrices in the flat buffers (contiguous: the 1-D parameters live in the leading small region)."""
        p = f"transformer.layers.{i}."
        return self.flat.range_of(p + "attn.to_queries.weight", p + "attn.to_keys_values.weight", p + "attn.project.weight",
                                  p + "mlp.fc1.weight", p + "mlp.fc2.weight")

    def update_chunks(self) -> List[Tuple[int, int]]:
        """Ranges of the flat buffers in the order the forward reads them: 1-D parameters, ``linear``, ``prefix_const``, layer 0 .. n - 1
        (in the flat layout ``prefix_const`` - a direct parameter - comes first and ``linear.weight`` last)."""
        fl = self.flat
        return _cover(fl, [(0, fl.small_numel), fl.range_of("linear.weight"), fl.range_of("prefix_const")] + [self.layer_range(i) for i in range(self.num_layers)])

    def forward(self, x: Tensor) -> Tensor:
        """``x``: [B, D] (or [B,1,1,D]) -> the mapper's whole residual stream cast to the compute dtype,
        ``[B, clip_length + L, E]``; the caller reads rows ``[:, clip_length:]`` (clipcap.py:220)."""
        B = x.shape[0]
        x2 = x.reshape(B, -1)
        self.flat.refresh_shadow()
        y = _TransformerMapperFunction.apply(x2, self, *[p for _, p in self._names])
        return y.view(B, self.clip_length + self.prefix_length, self.E)


class _TransformerMapperFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod: "TransformerMapper", *params):
        fl, T = mod.flat, mod.dtype
        B, CL, L, E, H = x.shape[0], mod.clip_length, mod.prefix_length, mod.E, mod.HEADS
        N, hd = CL + L, E // H
        xT = _to_compute(x, T)
        fl.wait_ready(0, fl.small_numel)
        fl.wait_ready(*fl.range_of("linear.weight"))
        fl.wait_ready(*fl.range_of("prefix_const"))
        lin = ops.gemm(xT, fl.w("linear.weight"), bias=fl.f("linear.bias"))          # [B, CL*E]
        # rows of the stream: clip rows come from `lin` viewed [B*CL, E], const rows from prefix_const
        idx = torch.arange(N, device=x.device, dtype=torch.int32)
        b_idx = torch.arange(B, device=x.device, dtype=torch.int32)[:, None]
        src = torch.where(idx[None] < CL, -(1 + b_idx * CL + idx[None]), (idx[None] - CL).expand(B, N)).contiguous()
        h = ops.embed_assemble(src, None, fl.w("prefix_const"), lin.view(B * CL, E), None)   # fp32 [B*N, E]
        tape = []
        for i in range(mod.num_layers):
            p = f"transformer.layers.{i}."
            fl.wait_ready(*mod.layer_range(i))                # a pipelined optimiser may still be updating the later layers
            a, m1, r1 = ops.layernorm_fwd(h, fl.f(p + "norm1.weight"), fl.f(p + "norm1.bias"), 1e-5, T, save_stats=True)
            q = ops.gemm(a, fl.w(p + "attn.to_queries.weight"))
            kv = ops.gemm(a, fl.w(p + "attn.to_keys_values.weight"))
            ctxv, lse = ops.attention_fwd(q, kv[:, :E], kv[:, E:], B, H, N, N, hd, causal=False, scale=hd ** -0.5, save_lse=True)
            h1 = ops.gemm(ctxv, fl.w(p + "attn.project.weight"), bias=fl.f(p + "attn.project.bias"), residual=h, out_f32=True)
            a2, m2, r2 = ops.layernorm_fwd(h1, fl.f(p + "norm2.weight"), fl.f(p + "norm2.bias"), 1e-5, T, save_stats=True)
            f1 = ops.gemm(a2, fl.w(p + "mlp.fc1.weight"), bias=fl.f(p + "mlp.fc1.bias"), act="relu")
            h2 = ops.gemm(f1, fl.w(p + "mlp.fc2.weight"), bias=fl.f(p + "mlp.fc2.bias"), residual=h1, out_f32=True)
            tape.append((h, m1, r1, a, q, kv, ctxv, lse, h1, m2, r2, a2, f1))
            h = h2
        fl.wait_ready()
        out = ops.cast_rows(h, T) if T != torch.float32 else h
        ctx.mod, ctx.tape, ctx.src, ctx.xT = mod, tape, src, xT
        ctx.dims = (B, CL, L, E, H, N, hd)
        return out

    @staticmethod
    def backward(ctx, dout):
        mod: TransformerMapper = ctx.mod
        fl, T = mod.flat, mod.dtype
        B, CL, L, E, H, N, hd = ctx.dims
        acc = fl.grad_live and not mod.grads_were_reset()
        lowp = T != torch.float32
        dout = dout.contiguous().view(B * N, E)
        # the stream gradient is kept in fp32; only rows [:, CL:] of the output were consumed, the
        # caller's scatter leaves the clip rows zero
        dh = dout.float() if lowp else dout.clone()

        def as_T(t):
            return ops.cast_rows(t, T) if lowp else t

        for i in reversed(range(mod.num_layers)):
            p = f"transformer.layers.{i}."
            h, m1, r1, a, q, kv, ctxv, lse, h1, m2, r2, a2, f1 = ctx.tape[i]
            dhT = as_T(dh)
            # h2 = h1 + fc2(relu(fc1(a2)))
            _wgrad(dhT, f1, fl.g(p + "mlp.fc2.weight"), acc)
            ops.colsum(dhT, fl.g(p + "mlp.fc2.bias"), acc)
            du = _dgrad(dhT, fl.w(p + "mlp.fc2.weight"), act="relu", aux_in=f1)   # relu'(u) == (f1 > 0)
            _wgrad(du, a2, fl.g(p + "mlp.fc1.weight"), acc)
            ops.colsum(du, fl.g(p + "mlp.fc1.bias"), acc)
            da2 = _dgrad(du, fl.w(p + "mlp.fc1.weight"))
            g2w, g2b = fl.g(p + "norm2.weight"), fl.g(p + "norm2.bias")
            if not acc:
                g2w.zero_(); g2b.zero_()
            dh1 = ops.layernorm_bwd(h1, da2, fl.f(p + "norm2.weight"), m2, r2, dres=dh, dgamma=g2w, dbeta=g2b, out=dh)
            dh1T = as_T(dh1)
            # h1 = h + project(attn(q, k, v))
            _wgrad(dh1T, ctxv, fl.g(p + "attn.project.weight"), acc)
            ops.colsum(dh1T, fl.g(p + "attn.project.bias"), acc)
            dctx = _dgrad(dh1T, fl.w(p + "attn.project.weight"))
            dq = torch.empty_like(q)
            dkv = torch.empty_like(kv)
            ops.attention_bwd(q, kv[:, :E], kv[:, E:], ctxv, dctx, lse, B, H, N, N, hd, causal=False, scale=hd ** -0.5,
                              dq=dq, dk=dkv[:, :E], dv=dkv[:, E:])
            _wgrad(dq, a, fl.g(p + "attn.to_queries.weight"), acc)
            _wgrad(dkv, a, fl.g(p + "attn.to_keys_values.weight"), acc)
            da = _dgrad(dq, fl.w(p + "attn.to_queries.weight"), out_f32=True)
            da = _dgrad(dkv, fl.w(p + "attn.to_keys_values.weight"), residual=da, out=da)
            g1w, g1b = fl.g(p + "norm1.weight"), fl.g(p + "norm1.bias")
            if not acc:
                g1w.zero_(); g1b.zero_()
            dh = ops.layernorm_bwd(h, as_T(da), fl.f(p + "norm1.weight"), m1, r1, dres=dh1, dgamma=g1w, dbeta=g1b, out=dh1)
            fl.notify_grad(*mod.layer_range(i))               # this layer's weight gradients are final: a sharded exchange may start on them
        # stream assembly: prefix_const rows (sum over batch) and the linear projection rows
        ops.colsum(dh.view(B, N * E)[:, CL * E:], fl.g("prefix_const").view(L * E), acc)
        dlin = ops.embed_assemble_bwd(ctx.src, dh, B * CL, T).view(B, CL * E)
        _wgrad(dlin, ctx.xT, fl.g("linear.weight"), acc)
        ops.colsum(dlin, fl.g("linear.bias"), acc)
        fl.notify_grad(0, fl.numel)
        fl.grad_live = True
        mod.attach_grads()
        return (None, None) + (None,) * len(mod._names)


# =========================================================================== the prefix LM function
class _PrefixLMLoss(torch.autograd.Function):
    """loss = CE(LM([prefix rows | text])) with the frozen LM's dgrad as backward."""

    @staticmethod
    def forward(ctx, prefix_rows, lm: FrozenCausalLM, src, pos, mask, labels, B, S, holder):
        out = lm.forward(prefix_rows, src, pos, mask, B, S, labels=labels, save=True, pack=holder.pop("pack", False),
                         lengths=holder.pop("lengths", None), n_scored=holder.pop("n_scored", None))
        ctx.lm, ctx.tape, ctx.n_rows = lm, out["tape"], prefix_rows.shape[0]
        holder.update(out)
        return out["loss"].view(())

    @staticmethod
    def backward(ctx, gloss):
        g = gloss.reshape(1).to(torch.float32).contiguous()
        d = ctx.lm.backward(ctx.tape, g, ctx.n_rows)
        ctx.tape = None
        return (d,) + (None,) * 8


class _Output:
    """``.loss`` and ``.logits`` like HF's ``CausalLMOutputWithCrossAttentions`` (clipcap.py:337-342).
    ``.logits`` [B, L+T, V] float32 is assembled on first access: after a packed forward the padded positions (whose
    reference logits are never used: no loss, never attended) read as zeros."""

    def __init__(self, loss, rows_logits, B, S, V, flat_index=None, scored_only=None):
        self.loss = loss
        self._lg, self._B, self._S, self._V, self._flat = rows_logits, B, S, V, flat_index
        self._scored_only = scored_only      # (lm, hidden): only the scored rows went through the lm_head during the step
        self._full = None

    @property
    def logits(self):
        if self._full is None:
            if self._scored_only is not None:
                lm, hidden = self._scored_only
                with torch.no_grad():
                    self._lg = lm._head(hidden)          # on demand: the logits of every packed row
                self._scored_only = None
            if self._flat is None:
                self._full = self._lg.view(self._B, self._S, -1)[:, :, :self._V]
            else:
                full = torch.zeros((self._B * self._S, self._V), device=self._lg.device, dtype=self._lg.dtype)
                full[self._flat.long()] = self._lg[:, :self._V]
                self._full = full.view(self._B, self._S, self._V)
        return self._full


# =========================================================================== models
def _resolve_lm(model_version: str, dtype, device, seed: int = 2021, weight_format: str = "native") -> FrozenCausalLM:
    """``GPT2LMHeadModel.from_pretrained(model_version)`` clipcap.py:252 without network access:
    a local HF directory is loaded; a known architecture name gets seeded random-init weights
    (synthetic benchmarking - say so wherever results are reported)."""
    root = os.environ.get("EAVQA_MODEL_DIR", "")
    for cand in (model_version, os.path.join(root, model_version) if root else ""):
        if cand and os.path.isdir(cand) and os.path.exists(os.path.join(cand, "config.json")):
            cfgd, sd = load_local_hf(cand)
            return FrozenCausalLM(LMConfig.from_hf_dict(cfgd), sd, dtype, device, weight_format)
    if model_version in KNOWN_CONFIGS:
        synthetic_weights_notice(model_version)
        cfg = LMConfig.from_hf_dict(KNOWN_CONFIGS[model_version])
        lm = FrozenCausalLM(cfg, random_init_state_dict(cfg, seed, device), dtype, device, weight_format)
        lm.synthetic_weights = True                   # executors / bench stamp their outputs with it
        return lm
    raise FileNotFoundError(f"{model_version!r}: not a local HF directory and not a known architecture name; "
                            "no network access is attempted")


class ClipCaptionModel(nn.Module):
    """``ClipCaptionModel`` clipcap.py:240-471."""

    def __init__(self, prefix_length: int, clip_length: Optional[int] = None, prefix_size: int = 512, num_layers: int = 8,
                 mapping_type: str = "mlp", model_version: str = "gpt2", *, lm: Optional[FrozenCausalLM] = None,
                 dtype: torch.dtype = torch.bfloat16, device="cuda", lm_weight_format: str = "native"):
        """``lm_weight_format="fp8"`` (an addition to the reference's ``model_args``, BASELINE configs[4]): hold the frozen LM's
        Linear weights in e4m3 and run them on the block-scaled MFMA (models/lm.py ``Fp8Weight``)."""
        super().__init__()
        self.prefix_length = prefix_length
        self.dtype, self.device_ = dtype, torch.device(device)
        self.gpt = lm if lm is not None else _resolve_lm(model_version, dtype, device, weight_format=lm_weight_format)
        self.gpt_embedding_size = self.gpt.cfg.n_embd
        E = self.gpt_embedding_size
        self.mapping_type = mapping_type
        if mapping_type == "mlp":
            self.clip_project = MLP((prefix_size, (E * prefix_length) // 2, E * prefix_length), device=device, dtype=dtype)
        else:
            self.clip_project = TransformerMapper(prefix_size, E, prefix_length, clip_length, num_layers, device=device, dtype=dtype)
        # with a host-side label count the packed training forward runs the lm_head on the scored rows only
        self.score_labeled_rows_only = True
        # training forward drops padded positions before the first GEMM (exact for loss / gradients / attended logits)
        self.pack_padding = True

    # -- helpers ----------------------------------------------------------------------------
    def get_dummy_token(self, batch_size: int, num_question_tokens: int, device) -> Tensor:
        return torch.full((batch_size, self.prefix_length + num_question_tokens), -100, dtype=torch.int64, device=device)

    def _project(self, prefix: Tensor) -> Tuple[Tensor, int, int]:
        """Mapper output as LM prefix rows + (row stride, row offset) of sample b's L slots."""
        L, E = self.prefix_length, self.gpt_embedding_size
        if self.mapping_type == "mlp":
            rows = self.clip_project(prefix).reshape(-1, E)                    # [B*n_img*L, E]   (:318-320)
            return rows, L, 0
        B = prefix.shape[0]
        if prefix.numel() // B != self.clip_project.linear.in_features:
            raise ValueError("TransformerMapper takes one CLIP embedding per sample (clipcap.py:215)")
        stream = self.clip_project(prefix)                                     # [B, CL+L, E]
        CL = self.clip_project.clip_length
        return stream.reshape(-1, E), CL + L, CL

    # -- training forward --------------------------------------------------------------------
    def forward(self, question_tokens: Tensor, prefix: Tensor, question_mask: Optional[Tensor] = None,
                labels: Optional[Tensor] = None, pad_token_id: Optional[int] = None, question_lengths=None,
                label_count: Optional[int] = None):
        """``forward`` clipcap.py:290-342 -> object with ``.loss`` (0-d, differentiable w.r.t. the mapper)
        and ``.logits`` [B, L+T, V] float32.  ``question_lengths`` (optional host ints: attended tokens per row,
        i.e. ``question_mask.sum(1)``) spares the packed path one device->host read.  ``label_count`` (optional host int,
        ``(labels != -100).sum()``; taken from ``labels`` itself when that is a host tensor) lets the packed training
        path run the lm_head on the scored positions only: loss and gradients are unchanged; ``.logits``, if read, is
        computed on demand for all packed rows."""
        dev = self.device_
        B, T = question_tokens.shape
        L = self.prefix_length
        S = L + T
        tok = question_tokens.to(dev)
        qm = question_mask.to(dev) if question_mask is not None else torch.ones_like(tok)
        rows, stride, off = self._project(prefix.to(dev))
        src, mask, pos = ops.build_prefix_rows(tok, qm, L, self.gpt.cfg.pos_mode, stride, off)
        if labels is not None:
            full = torch.cat((self.get_dummy_token(B, 0, dev), labels.to(dev)), dim=1).contiguous()   # :323-335
            pack = bool(self.pack_padding and self.training)
            lengths = None
            if pack:
                if question_lengths is None:
                    if question_mask is None:
                        question_lengths = [T] * B
                    elif not question_mask.is_cuda:
                        question_lengths = (question_mask != 0).sum(1).tolist()     # host tensor: free
                if question_lengths is not None:
                    lengths = [L + int(n) for n in question_lengths]
            if pack and label_count is None and not labels.is_cuda:
                label_count = int((labels != -100).sum())                           # host tensor: free
            holder: dict = {"pack": pack, "lengths": lengths, "n_scored": label_count if (pack and self.score_labeled_rows_only) else None}
            loss = _PrefixLMLoss.apply(rows, self.gpt, src, pos, mask, full, B, S, holder)
            scored_only = (self.gpt, holder["hidden"]) if holder.get("sel") is not None else None
            return _Output(loss, holder["logits"], B, S, self.gpt.vocab, holder["flat_index"] if pack else None, scored_only)
        lg = self.gpt.forward(rows, src, pos, mask, B, S, logits="all")["logits"]
        return _Output(None, lg, B, S, self.gpt.vocab)

    # -- generation --------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, question_tokens: Tensor, prefix: Tensor, question_mask: Optional[Tensor] = None,
                 **generation_kwargs) -> List[List[int]]:
        """``generate`` clipcap.py:344-385."""
        dev = self.device_
        tok = question_tokens.to(dev)
        qm = question_mask.to(dev) if question_mask is not None else torch.ones_like(tok)
        rows, stride, off = self._project(prefix.to(dev))
        return self._generate_from_rows(rows, tok, qm, stride, off, **generation_kwargs)

    def _generate_from_rows(self, rows, tok, qm, stride, off, max_length: Optional[int] = 10,
                            pad_token_id: Optional[int] = None, eos_token_id: Optional[int] = None,
                            use_cache: bool = True, output_scores: bool = False):
        """``_generate_from_embeddings`` clipcap.py:387-471 (greedy; finished rows emit pad; the embedding
        fed back is the RAW argmax :423; early stop :463)."""
        from .decode import greedy_decode
        lm = self.gpt
        pad_token_id = pad_token_id if pad_token_id is not None else lm.cfg.pad_token_id
        eos_token_id = eos_token_id if eos_token_id is not None else lm.cfg.eos_token_id
        if eos_token_id is not None and pad_token_id is None:
            raise ValueError("If `eos_token_id` is defined, make sure that `pad_token_id` is defined.")   # :426-430
        B, T = tok.shape
        L = self.prefix_length
        # masks / positions for the whole horizon at once: appended tokens are always attended (:444-453)
        tok_ext = torch.cat([tok, torch.zeros((B, max_length), dtype=tok.dtype, device=tok.device)], dim=1)
        qm_ext = torch.cat([qm.to(torch.int64), torch.ones((B, max_length), dtype=torch.int64, device=tok.device)], dim=1)
        src, mask, pos = ops.build_prefix_rows(tok_ext, qm_ext, L, lm.cfg.pos_mode, stride, off)
        return greedy_decode(lm, rows, src, mask, pos, B, L + T, max_length, pad_token_id, eos_token_id, use_cache, output_scores)


    @torch.no_grad()
    def generate_fewshot(self, question_tokens: Tensor, prefix: Tensor, question_mask: Optional[Tensor] = None,
                         num_shots: Optional[int] = None, special_token_id: int = 32099, max_length: Optional[int] = 10,
                         pad_token_id: Optional[int] = None, eos_token_id: Optional[int] = None,
                         use_cache: bool = True, output_scores: bool = False, marks: Optional[list] = None):
        """Few-shot prompt path: the causal-LM counterpart of ``VCT0Model.generate`` with
        ``insert_prefix_into_input`` (src/models/vct0.py:446-464,494-533).  ``prefix``: [B, n_img, D] (or
        [B, n_img, 1, D]) CLIP embeddings; the n-th sentinel token (ids ``special_token_id - i``) of each row
        expands into the L prefix vectors of image n."""
        from .decode import greedy_decode
        if self.mapping_type != "mlp":
            raise NotImplementedError("several images per row need the MLP mapper (as in the reference configs)")
        dev = self.device_
        lm = self.gpt
        tok = question_tokens.to(dev)
        qm = question_mask.to(dev) if question_mask is not None else torch.ones_like(tok)
        B, T = tok.shape
        prefix = prefix.to(dev).reshape(B, -1, prefix.shape[-1])
        n_img = prefix.shape[1]
        if num_shots is not None and num_shots + 1 != n_img:
            raise ValueError("num_shots + 1 must equal the number of images per row")
        pad_token_id = pad_token_id if pad_token_id is not None else lm.cfg.pad_token_id
        eos_token_id = eos_token_id if eos_token_id is not None else lm.cfg.eos_token_id
        if eos_token_id is not None and pad_token_id is None:
            raise ValueError("If `eos_token_id` is defined, make sure that `pad_token_id` is defined.")
        L = self.prefix_length
        rows = self.clip_project(prefix).reshape(-1, self.gpt_embedding_size)          # [(b, n, l), E]
        tok_ext = torch.cat([tok, torch.zeros((B, max_length), dtype=tok.dtype, device=dev)], dim=1)
        qm_ext = torch.cat([qm.to(torch.int64), torch.ones((B, max_length), dtype=torch.int64, device=dev)], dim=1)
        src, mask, pos, status = ops.build_fewshot_rows(tok_ext, qm_ext, L, n_img, special_token_id, lm.cfg.pos_mode)
        if not bool((status == n_img).all().item()):
            raise ValueError("every row must hold exactly one sentinel token per image")   # vct0.py:512 .view fails
        S0 = T + (L - 1) * n_img
        return greedy_decode(lm, rows, src, mask, pos, B, S0, max_length, pad_token_id, eos_token_id, use_cache, output_scores, marks)


class ClipCaptionPrefix(ClipCaptionModel):
    """``ClipCaptionPrefix`` clipcap.py:590-599: only the mapper trains; the LM stays frozen / eval."""

    def parameters(self, recurse: bool = True):
        return self.clip_project.parameters()

    def train(self, mode: bool = True):
        super().train(mode)
        return self   # the LM holds no torch parameters: it is frozen by construction
