"""The planes' half of the training step's entropy pass — forward AND backward — as one captured HIP graph.

Between two occupancy refreshes nothing about that half changes but the VALUES it reads: the vertex lists of the three
planes, their slot orders, the vote plan and every tensor shape are functions of the occupancy grid (rebuilt every
`step_update` steps, `CNC_context_models.structures_version`), the tables and the context heads are updated in place, the
sign bit planes are repacked in place (`GridEncoder._bit_plane`), and there is no random draw and no host round trip in
it (utils_bpp_acc.py:556-617: the planes' levels are coded whole; the random windows are the 3-D table's).  Launched op
by op it is ~50 kernels forward and ~60 backward from the entropy pass's host thread — ~1.2 ms of launches in the
launch-bound first half of the step, in front of the 3-D half, which is the step's critical chain.  Captured once per
refresh (stream capture on the planes' stream, thread-local mode: the render pass's thread keeps launching) it is ONE
graph launch per step.

What the graph computes, exactly as `CNC_context_models._bits_2D` + autograd do eagerly:
    bits_2D            the planes' bits (a static 0-dim tensor, read by the 3-D half for the totals)
    gradients          of  lmbda * (bits_2D / n_params) * loss_scale  (the planes' share of the joint root of
                       Trainer._context_pass: same factors in the same order)
                       - the encoders' scatters and the heads' weight gradients add into the step's gradient sink
                         (cnc_amd._gradsink), as in the eager pass;
                       - what autograd returns for the tables (through the STE: zero-order statistics, rate kernels,
                         the dimension-wise votes into the finest 3-D level) is kept in static tensors and added to
                         `.grad` by `flush()`, on the main stream, after the join.

Recording autograd (what it took on ROCm 7.2 / PyTorch 2.10):
    * the leaves are DETACHED aliases of the tables made inside the capture (their `.grad`, filled by the recorded
      backward, are the static tensors): a Parameter's AccumulateGrad node lives on the stream it was first used on, the
      engine then joins that stream to the capturing one, and ending the capture with a foreign stream forked into it
      crashes the runtime.  The context heads and the encoders' own tables are Parameters too — their gradients never
      reach autograd (they go into the sink), which `capture()` checks before it records anything;
    * the engine runs on the capturing thread (`set_multithreading_enabled(False)`);
    * no BLAS call (`torch.dot` creates its handle inside the capture and fails): `_plane_bits` uses a product and a sum;
    * the device's default random generator is in capture mode for the duration: no other thread may draw from it, so the
      Trainer captures BEFORE it forks the step's passes off;
    * no hipMemsetAsync in the recorded region: the memset NODE of cnc_level_stats_forward's few bytes was not reliably
      ordered in front of the kernel behind it when other streams were busy (level sums came out as garbage once in a few
      dozen replays, Pg and everything behind it NaN) — that zeroing is a kernel now (ctx_head.hip).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import _gradsink
from .gridencoder import STE_binary


class PlanesGraph:
    def __init__(self, trainer):
        self.tr = trainer
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.key = None
        self.bits: Optional[torch.Tensor] = None
        self.n_params = 0
        self.pairs: List[Tuple[torch.Tensor, torch.Tensor, Optional[Tuple[int, int]]]] = []   # (parameter, static gradient, its rows)
        self.step_bits: Optional[torch.Tensor] = None     # what THIS step's planes' half left (a replay's, or an eager run's)
        self.step_pairs: List[Tuple[torch.Tensor, torch.Tensor, Optional[Tuple[int, int]]]] = []
        self._tables_used: List[bool] = []
        self._small_used = False
        self.captures = 0
        self.replays = 0
        self._pool = None

    def _key(self):
        tr = self.tr
        b = tr.estimator.binaries
        return (tr.context.structures_version, b.data_ptr(), tuple(b.shape), id(tr.sink_ctx), float(tr.cfg.lmbda),
                float(tr.loss_scale))

    def ready(self) -> bool:
        return self.graph is not None and self.key == self._key()

    def drop(self) -> None:
        self.graph, self.key, self.bits, self.pairs = None, None, None, []

    def run_eager(self) -> None:
        """The same forward + backward, op by op, NOW (an occupancy-refresh step: the structures have just been rebuilt and
        no graph exists for them yet): the bits in `self.bits`, the tables' autograd gradients in `self.pairs` for
        `flush()`, everything else in the sink — exactly what a replay leaves behind."""
        self.capture(record=False)

    def capture(self, record: bool = True) -> None:
        """Record the planes' forward + backward on the planes' stream (the caller's current stream must be it, with the
        step's gradient sink active on this thread).  Nothing runs: `replay()` does.  (`record=False`: `run_eager`.)"""
        tr = self.tr
        ctxm, mb, c = tr.context, tr.field.mlp_base, tr.cfg
        encs = (mb.encoding_xy, mb.encoding_xz, mb.encoding_yz)
        exyz = mb.encoding_xyz
        o0, o1 = int(ctxm._off3_host[-2]), int(ctxm._off3_host[-1])
        n_total = sum(e.params.numel() for e in encs) + exyz.params.numel()
        sink = _gradsink.current()
        if record:
            self.drop()
        heads = [p for p in ctxm.parameters() if p.requires_grad]
        if sink is None or any(sink._small_at.get(p.data_ptr(), (0, -1))[1] != p.numel() for p in heads) \
                or any(sink.table(e.params) is None for e in encs):
            raise RuntimeError("the planes' graph needs the step's gradient sink to hold the context heads and the tables")
        # the tables as leaves of their own: aliases of the parameters' storage (the optimizer updates it in place); of the
        # 3-D table only its finest level, so that its gradient is level-sized, not table-sized
        leaves = [e.params.detach().requires_grad_() for e in encs]
        leaf_fin = exyz.params.detach()[o0:o1].requires_grad_()
        g = None
        # No garbage collection while the stream records: a collection that runs into ANOTHER PlanesGraph's dead cycle (a
        # Trainer a test has dropped) destroys a graph and releases its pool from this thread in the middle of the
        # recording, which the runtime refuses — and a destructor that fails ends the process (seen once the test files ran
        # in another order).  Reference counts still free what this method itself lets go of.
        import gc
        gc_was_on = record and gc.isenabled()
        if gc_was_on:
            gc.disable()
        try:
            if record:
                g = torch.cuda.CUDAGraph()
                # (not `with torch.cuda.graph(...)`: that synchronises the device and empties the allocator's cache on entry —
                # the render pass is running on its own thread and stream next to this)
                # ONE memory pool for every recording of this object: a graph records into a private pool, and a pool of its own
                # per recording left ~0.5 GB reserved behind every occupancy refresh (18 -> 31 GB over 24 refreshes; the
                # allocator only returns a dead graph's pool under memory pressure) and a hipMalloc of that size inside every
                # recording step — 25-35 ms for that step on some boxes.  The previous graph is dropped above, so its blocks are
                # free in the pool when this one records.
                # (A pool handle may only be passed again while a graph that records into it is alive: a one-kernel graph recorded
                # once holds it for this object's lifetime.)
                if self._pool is None:
                    pool, keeper = torch.cuda.graph_pool_handle(), torch.cuda.CUDAGraph()
                    keeper.capture_begin(pool=pool, capture_error_mode="thread_local")
                    try:
                        self._keeper_out = torch.zeros(1, device=exyz.params.device)
                    finally:
                        keeper.capture_end()
                    self._pool, self._keeper = pool, keeper
                g.capture_begin(pool=self._pool, capture_error_mode="thread_local")
            try:
                with torch.autograd.set_multithreading_enabled(False):
                    pq = [STE_binary.apply(t) for t in leaves]
                    fin = STE_binary.apply(leaf_fin)
                    bits, n2 = ctxm._bits_2D(*encs, *pq, fin, tr.estimator.binaries, ctxm._binary_2D, ctxm.idx_coords2_tmp, False)
                    root = c.lmbda * (bits / n_total) * tr.loss_scale
                    # the WHOLE graph (torch.autograd.grad with the leaves as inputs would skip every node that does not lead to
                    # them: the encoders' and the heads' backward, whose gradients go into the sink, not to autograd)
                    root.backward()
                    grads = [t.grad for t in leaves + [leaf_fin]]
                    bits = bits.detach()
            finally:
                if g is not None:
                    g.capture_end()
        finally:
            if gc_was_on:
                gc.enable()
        targets = [e.params for e in encs] + [exyz.params]
        pairs = []
        for k, (p, gr) in enumerate(zip(targets, grads)):
            if gr is not None:
                pairs.append((p, gr, (o0, o1) if k == 3 else None))
        self.n_params = n2
        if not record:                    # this step's results; a graph recorded behind this leaves them alone
            self.step_bits, self.step_pairs = bits, pairs
            return
        self.graph, self.bits, self.pairs = g, bits, pairs
        if record:
            # what the recorded kernels add into (`replay` marks it, `GradSink.flush` skips buffers nobody marked): the three
            # planes' scatters and the heads' weight gradients (the check above has marked them for this step already)
            plane_ptrs = {e.params.data_ptr() for e in encs}
            self._tables_used = [t.data_ptr() in plane_ptrs for t in sink.tables]
            self._small_used = True
            self.key = self._key()
            self.captures += 1

    def replay(self):
        """One graph launch on the current stream (the planes' stream, ordered after the step's fork) -> (bits, parameter
        count) of the planes."""
        self.graph.replay()
        sink = _gradsink.current()
        if sink is not None:          # what the captured kernels add into: `flush` skips buffers nobody marked
            for k, used in enumerate(self._tables_used):      # (in place, only ever to True: the 3-D half marks the same
                if used:                                      # list from its own thread meanwhile)
                    sink._tables_used[k] = True
            if self._small_used:
                sink._small_used = True
        self.replays += 1
        self.step_bits, self.step_pairs = self.bits, self.pairs
        return self.bits, self.n_params

    @torch.no_grad()
    def flush(self, table_pieces=None) -> None:
        """The gradients autograd returned inside the graph -> `.grad` (after the sinks' flush, on the stream every
        pass has been joined to) — or, with `table_pieces` (see GradSink.flush), listed there for the tables' optimizer
        kernel: `(gradient, rows of the table it covers or None)`."""
        if table_pieces is not None:
            for p, g, rows in self.step_pairs:
                table_pieces.setdefault(id(p), []).append((g, rows))
            return
        add_to, add_from = [], []
        for p, g, rows in self.step_pairs:
            if p.grad is None:
                p.grad = torch.zeros_like(p) if rows is not None else g.clone()
                if rows is None:
                    continue
            add_to.append(p.grad if rows is None else p.grad[rows[0]:rows[1]])
            add_from.append(g)
        if add_to:
            torch._foreach_add_(add_to, add_from)
