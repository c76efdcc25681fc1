"""The optimizer's update of the feature tables as ONE kernel that sums the step's gradient pieces on the way in
(cnc_table_adam, csrc/table_adam.hip) — for the single-process training step (cnc_amd.trainer).

The tables stay in the Trainer's torch.optim.Adam (their own parameter group: one learning-rate schedule for everything),
and the kernel works on THAT optimizer's state tensors (`exp_avg`, `exp_avg_sq`, the float32 device-side `step` of its
fused form): a step may go through either — `TableAdam.step(...)` leaves the tables' `.grad` None, which is how
`Optimizer.step()` skips a parameter, so the library's step that follows updates everything else; a step whose gradients
were flushed into `.grad` instead (data parallel, the tests that read `.grad`) goes through the library as before.

Reference: torch.optim.Adam(lr, eps=1e-15, weight_decay) over every parameter, stepped once per iteration
(examples/train_CNC_nerf_synthetic.py:254-259,363).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _caches, _lib

Piece = Tuple[torch.Tensor, Optional[Tuple[int, int]]]       # (gradient piece, the ROWS of the table it covers or None = all)


class TableAdam:
    def __init__(self, opt: torch.optim.Adam, tables: Sequence[torch.nn.Parameter], encoders: Optional[Sequence] = None):
        """`encoders` (optional, one GridEncoder per table, `encoder.params is table`): the kernel then also leaves each
        updated table's sign bit plane and clip counter in the encoder's cache buffers (what `GridEncoder._bit_plane` would
        make by reading the table again at the next forward); `mark_planes_current()` re-validates them after the library
        optimizers' post-step hook has dropped every cache."""
        self.opt = opt
        self.tables: List[torch.nn.Parameter] = list(tables)
        self.encoders = list(encoders) if encoders is not None else None
        if self.encoders is not None and (len(self.encoders) != len(self.tables)
                                          or any(e.params is not p for e, p in zip(self.encoders, self.tables))):
            raise ValueError("TableAdam: one encoder per table, in the tables' order")
        self._planes_written: List = []
        if not 1 <= len(self.tables) <= 4:
            raise ValueError("TableAdam: one to four tables")
        ids = {id(p) for p in self.tables}
        groups = [g for g in opt.param_groups if any(id(p) in ids for p in g["params"])]
        if len(groups) != 1 or {id(p) for p in groups[0]["params"]} != ids:
            raise ValueError("TableAdam: the tables must be one parameter group of the optimizer, and all of it")
        self.group = groups[0]
        g = self.group
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable") \
                or g.get("decoupled_weight_decay"):
            raise ValueError("TableAdam: plain Adam only (no amsgrad / maximize / capturable / decoupled decay)")
        for p in self.tables:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.numel() % 4:
                raise ValueError("TableAdam: contiguous float32 device tables of a multiple of 4 elements")
        self.steps_done = 0            # host mirror of the state's step count (the bias corrections are host scalars)

    def _state(self, p):
        st = self.opt.state[p]
        if len(st) == 0:               # as Adam._init_group lays it out for the fused form
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if not (torch.is_tensor(st["step"]) and st["step"].is_cuda and st["step"].dtype == torch.float32):
            raise RuntimeError("TableAdam: the optimizer keeps its step count on the host (not the fused form)")
        return st

    def resync(self) -> None:
        """Take the step count over from the optimizer's state (after steps that went through the library)."""
        st = self.opt.state.get(self.tables[0], {})
        self.steps_done = int(st["step"].item()) if len(st) else 0

    @torch.no_grad()
    def step(self, pieces: Dict[int, List[Piece]]) -> None:
        """One Adam update of every table from its gradient pieces (`pieces[id(p)]`, in the order they are to be summed; a
        table's own `.grad`, if autograd left one, goes first and is dropped).  On the current stream: every piece must
        be complete on it.  A table with no piece at all is updated with a zero gradient — like the library's step on a
        `.grad` of zeros (a table always has a gradient in a training step; moments and weight decay move it regardless)."""
        g = self.group
        a = _lib.AdamTables()
        a.n_tables = len(self.tables)
        keep = []
        for k, p in enumerate(self.tables):
            st = self._state(p)
            t = a.table[k]
            t.p, t.m, t.v, t.step, t.n = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), \
                st["step"].data_ptr(), p.numel()
            row = p.numel() // p.shape[0]
            src: List[Piece] = ([(p.grad, None)] if p.grad is not None else []) + list(pieces.get(id(p), ()))
            if len(src) > 4:
                raise RuntimeError("TableAdam: more than four gradient pieces for one table")
            for j, (gt, rows) in enumerate(src):
                lo, hi = (0, p.numel()) if rows is None else (rows[0] * row, rows[1] * row)
                if gt.dtype != torch.float32 or gt.device != p.device or not gt.is_contiguous() or gt.numel() != hi - lo:
                    raise RuntimeError("TableAdam: a gradient piece must be a contiguous float32 tensor of its range's size")
                t.g[j], t.g_lo[j], t.g_hi[j] = gt.data_ptr(), lo, hi
                keep.append(gt)
            p.grad = None
        # the updated tables' sign planes into the encoders' own cache buffers (kept at their addresses: a recorded graph
        # reads them there), when those exist in the size the table needs
        self._planes_written = []
        if self.encoders is not None:
            counters = []
            for k, (p, e) in enumerate(zip(self.tables, self.encoders)):
                bits, cc = getattr(e, "_bits", None), getattr(e, "_clip_count", None)
                if (p.numel() % 8 == 0 and bits is not None and cc is not None and bits.device == p.device
                        and bits.dtype == torch.uint8 and bits.numel() == p.numel() // 8 and cc.device == p.device
                        and cc.numel() == 1 and cc.dtype == torch.int32 and getattr(e, "bitplane", True)):
                    a.table[k].sign_bits, a.table[k].clip_count = bits.data_ptr(), cc.data_ptr()
                    counters.append(cc)
                    self._planes_written.append((e, p))
            if counters:
                torch._foreach_zero_(counters)
        b1, b2 = g["betas"]
        lr = g["lr"]
        self.steps_done += 1
        _lib.check(_lib.lib().cnc_table_adam(C.byref(a), float(lr), float(b1), float(b2), float(g["eps"]),
                                             float(g["weight_decay"]), float(self.steps_done),
                                             torch.cuda.current_stream(self.tables[0].device).cuda_stream), "cnc_table_adam")
        # the kernel writes the tables through their addresses: `Tensor._version` does not move, so the copies keyed on it
        # (the encoders' sign bit planes, packed weights) are dropped here as after any optimizer step (cnc_amd._caches)
        _caches.invalidate_all()
        self.mark_planes_current()

    def mark_planes_current(self) -> None:
        """The sign planes the last `step` wrote ARE the planes of the tables as they stand (nothing has written the tables
        since): re-validate the encoders' cache entries.  Called by `step` itself, and by the Trainer once more behind the
        library optimizers' steps — their post-step hook (cnc_amd._caches) drops every cache, it cannot know which
        parameters a step touched."""
        for e, p in self._planes_written:
            e._bits_key = (p.data_ptr(), p._version, tuple(p.shape))
            e._bits_src = (p,)
