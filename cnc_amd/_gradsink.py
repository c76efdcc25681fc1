"""Per-step gradient sinks for the training step (cnc_amd.trainer).

The custom backward kernels of the path ACCUMULATE: the encoder's scatter adds into a table-sized buffer with atomics,
the context heads' weight gradients into replicated buffers.  Under plain autograd every call hands the engine a FRESH
zero-filled buffer of the parameter's size and the engine adds them up — per training step 23 table-sized fills, 16
table-sized adds, and a fill + a reduction + two adds per context-head call (tools/aten_by_range.py: 90 of the step's
~400 library launches).  With a sink active on the calling thread those kernels add straight into ONE buffer per
parameter that is zeroed once per step, return no gradient to the engine, and the trainer adds the buffers to `.grad`
once, after both backward passes have been joined (`flush`).

Scope: thread-local and explicit (`with activate(sink)` around the FORWARD pass: the autograd functions remember the
sink that was current when they ran forward — their backward runs on autograd's device thread, where the caller's
thread-local does not exist): outside a Trainer step nothing changes — a backward call returns its gradients to
autograd as before.  One sink per pass (render / context): within a pass every kernel is on one stream, and the
finest-level scatter adds its slabs with plain read-modify-writes, which must not meet another stream's atomics.  Atomic adds from two streams into one sink are fine; what a sink must
never be is a tensor that something else read-modify-writes non-atomically at the same time, which is why it is not
`.grad` itself.
"""
from __future__ import annotations

import threading
from contextlib import contextmanager
from typing import Iterable, List, Optional

import torch

_tls = threading.local()
REPLICAS = 16          # copies of the small-parameter buffer the ~1000 workgroups of a weight-gradient kernel spread over


class GradSink:
    def __init__(self, tables: Iterable[torch.nn.Parameter], small: Iterable[torch.nn.Parameter]):
        self.tables: List[torch.nn.Parameter] = list(tables)
        self.small: List[torch.nn.Parameter] = list(small)
        dev = (self.tables + self.small)[0].device
        n_t = sum(p.numel() for p in self.tables)
        self.arena = torch.zeros(n_t, dtype=torch.float32, device=dev)
        self.table_views, self._table_at, o = [], {}, 0
        for k, p in enumerate(self.tables):
            self.table_views.append(self.arena[o:o + p.numel()].view_as(p))
            self._table_at[p.data_ptr()] = k
            o += p.numel()
        self.n_small = sum(p.numel() for p in self.small)
        self.replicas = torch.zeros((REPLICAS, max(self.n_small, 1)), dtype=torch.float32, device=dev)
        self._small_at, o = {}, 0
        for p in self.small:
            self._small_at[p.data_ptr()] = (o, p.numel())
            o += p.numel()
        self._tables_used = [False] * len(self.tables)
        self._small_used = False

    def zero(self) -> None:
        """Once per step, on the stream both backward passes are ordered after."""
        self.arena.zero_()
        if self.n_small:
            self.replicas.zero_()
        self._tables_used = [False] * len(self.tables)
        self._small_used = False

    def table(self, t: torch.Tensor) -> Optional[torch.Tensor]:
        """The buffer an encoder backward on table `t` adds into, or None (not one of this sink's tables)."""
        k = self._table_at.get(t.data_ptr())
        if k is None or t.shape != self.tables[k].shape:
            return None
        self._tables_used[k] = True
        return self.table_views[k]

    def small_slot(self, ws) -> Optional[List[Optional[torch.Tensor]]]:
        """For a list of small parameters (None entries allowed): their slices of replica 0 (the kernel addresses the
        other replicas at a stride of `stride()` floats), or None if any of them is not in the sink."""
        out = []
        for w in ws:
            if w is None:
                out.append(None)
                continue
            at = self._small_at.get(w.data_ptr())
            if at is None or at[1] != w.numel():
                return None
            out.append(self.replicas[0, at[0]:at[0] + at[1]])
        self._small_used = True
        return out

    def stride(self) -> int:
        return self.replicas.shape[1]

    @torch.no_grad()
    def flush(self, grads_of=None, table_pieces=None) -> None:
        """Add what the sinks hold to the parameters' gradients (`.grad`, or `grads_of(p)` -> the tensor to add into —
        the data-parallel step keeps the gradient in a flat bucket).  Call after the backward passes that used the
        sink have been joined to the current stream.  `table_pieces` (a dict): the tables' buffers are not added anywhere
        but listed there, `id(parameter) -> [(buffer, None), ...]` — the tables' optimizer kernel sums the pieces itself
        (cnc_amd._table_adam); they stay valid until the next `zero()`."""
        def target(p):
            return p.grad if grads_of is None else grads_of(p)
        add_to, add_from = [], []
        for p, v, used in zip(self.tables, self.table_views, self._tables_used):
            if not used:
                continue
            if table_pieces is not None:
                table_pieces.setdefault(id(p), []).append((v, None))
                continue
            g = target(p)
            if g is None:
                p.grad = v.clone()          # the arena is zeroed again next step: the gradient must not alias it
            else:
                add_to.append(g)
                add_from.append(v)
        if self._small_used and self.n_small:
            flat = self.replicas.sum(0)
            for p in self.small:
                o, n = self._small_at[p.data_ptr()]
                v = flat[o:o + n].view_as(p)
                g = target(p)
                if g is None:
                    p.grad = v              # a view of `flat`, which nothing else holds
                else:
                    add_to.append(g)
                    add_from.append(v)
        if add_to:
            torch._foreach_add_(add_to, add_from)


def current() -> Optional[GradSink]:
    return getattr(_tls, "sink", None)


@contextmanager
def activate(sink: Optional[GradSink]):
    """Make `sink` the calling THREAD's sink for the duration (None: no sink)."""
    prev = getattr(_tls, "sink", None)
    _tls.sink = sink
    try:
        yield sink
    finally:
        _tls.sink = prev
