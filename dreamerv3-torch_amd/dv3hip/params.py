"""Flat fp32 parameter / gradient buckets -- one per optimizer (model, actor, value).

Every nn.Parameter of a group becomes a view into one contiguous buffer, its .grad a view into a
second one.  That makes (a) the optimizer step three kernel launches regardless of the number of
tensors (tools.py:760-776 in the reference walks them one by one), and (b) the data-parallel
gradient exchange ONE all-reduce per optimizer on the flat gradient -- the RCCL-over-xGMI collective
SURVEY.md §5.8 asks for, inserted between backward (tools.py:765) and clipping (tools.py:768).
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist

from . import _dev, ops

# dev: issue the collective on a one-rank process group too (rehearses the stream topology of N > 1 on one GPU)
_FORCE_ALLREDUCE = _dev.flag("DV3_FORCE_ALLREDUCE", False)

_ALIGN = 4  # floats: keep every tensor 16-byte aligned inside the bucket


class ParamBucket:
    def __init__(self, name: str, params: Iterable[torch.nn.Parameter], allow_cpu: bool = False, extra: int = 0):
        """allow_cpu: layout + all-reduce logic on CPU tensors (multi-process gloo tests); step() still
        needs the GPU kernels.  extra: floats that ride behind the gradients through the same all-reduce (`tail`: the
        two return-normalisation EMA values travel with the critic's gradient instead of in a collective of their own);
        they are not parameters -- clipping and Adam never see them."""
        self.name = name
        self.params: List[torch.nn.Parameter] = [p for p in params]
        self.flat = None
        self._layout = None
        self._allow_cpu = allow_cpu
        self._extra = int(extra)
        self.tail = None

    # ------------------------------------------------------------------------------------------
    def _needs_build(self) -> bool:
        if self.flat is None:
            return True
        base, end = self.flat.data_ptr(), self.flat.data_ptr() + self.flat.numel() * 4
        for p, (off, n) in zip(self.params, self._layout):
            if p.data_ptr() != base + off * 4 or p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + off * 4:
                return True
        return False

    def settled(self) -> bool:
        """O(1) form of `not _needs_build()` for callers that REPLAY captured launches (hipGraphs hold raw pointers into
        `flat`): the bucket exists and its first and last parameter still live in it.  Whatever moves parameters moves
        all of them (Module.to, a rebuilt bucket)."""
        if self.flat is None:
            return False
        base = self.flat.data_ptr()
        (o0, _), (o1, _) = self._layout[0], self._layout[-1]
        return self.params[0].data_ptr() == base + 4 * o0 and self.params[-1].data_ptr() == base + 4 * o1

    def ensure(self):
        """(Re)flatten if any parameter no longer lives in the bucket (e.g. after Module.to())."""
        if not self._needs_build():
            return self
        dev = self.params[0].device
        if dev.type != "cuda" and not self._allow_cpu:
            raise RuntimeError(f"ParamBucket {self.name}: parameters must live on the GPU (got {dev}); "
                               "the MI355X hot path has no CPU implementation")
        layout, total = [], 0
        for p in self.params:
            n = p.numel()
            layout.append((total, n))
            total += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        extra = (self._extra + _ALIGN - 1) // _ALIGN * _ALIGN
        wire = torch.zeros(total + extra, dtype=torch.float32, device=dev)  # what crosses xGMI: gradients | tail
        grad = wire[:total]
        self._wire, self.tail = wire, (wire[total:total + self._extra] if self._extra else None)
        old_m = getattr(self, "exp_avg", None)
        for p, (off, n) in zip(self.params, layout):
            flat[off:off + n].copy_(p.data.reshape(-1).to(torch.float32))
            p.data = flat[off:off + n].view(p.shape)
            p.grad = grad[off:off + n].view(p.shape)
        self.flat, self.grad, self._layout = flat, grad, layout
        if old_m is None or old_m.numel() != total:
            self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
            self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
            self.state = torch.zeros(4, dtype=torch.float32, device=dev)  # step, sumsq, last norm, spare
            self._norm_partial = torch.zeros(1024, dtype=torch.float32, device=dev)  # ops.sumsq_ordered's scratch
        return self

    def numel(self) -> int:
        return sum(n for _, n in self._layout)

    def zero_grad(self):
        self.grad.zero_()

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def distributed() -> bool:
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_ALLREDUCE)

    def allreduce(self) -> float:
        """Sum gradients (and the tail) over ranks (RCCL when the process group is NCCL); returns the scale (1/world)
        the optimizer must apply.  No-op on a single rank."""
        if self.distributed():
            dist.all_reduce(self._wire, op=dist.ReduceOp.SUM)
            return 1.0 / dist.get_world_size()
        return 1.0

    def offset_of(self, param) -> int:
        """Start of `param` in the flat buffers (floats): where a caller may cut the bucket into two collectives."""
        for p, (off, _) in zip(self.params, self._layout):
            if p is param:
                return off
        raise ValueError(f"{self.name}: not a parameter of this bucket")

    def allreduce_range(self, lo: int, hi: int = None, async_op: bool = False):
        """The collective of allreduce() on floats [lo, hi) of the wire buffer only (hi=None: to its end, tail included).
        A sum over ranks is elementwise: two ranges reduce to exactly what one call over both would.  The world-model
        bucket is cut where its decoder / head gradients end, so that their half crosses xGMI while the encoder's
        backward still runs (graph.UpdateRunner).  -> the work handle (async_op) or None."""
        if not self.distributed():
            return None
        hi = self._wire.numel() if hi is None else hi
        if hi <= lo:
            return None
        return dist.all_reduce(self._wire[lo:hi], op=dist.ReduceOp.SUM, async_op=async_op)

    def step(self, *, lr, eps, clip, weight_decay=0.0, grad_scale=1.0):
        """clip_grad_norm_ + Adam on the whole bucket; the pre-clip norm lands in state[2]."""
        # (fixed summation order: replicas with the same all-reduced gradient clip and step bit-identically)
        ops.sumsq_ordered(self.grad, self.state[1:2], self._norm_partial)
        ops.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.state, lr=lr, eps=eps, clip=clip,
                      weight_decay=weight_decay, grad_scale=grad_scale)

    @property
    def grad_norm(self) -> torch.Tensor:
        return self.state[2]
