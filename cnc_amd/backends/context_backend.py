"""Host mirror of the context-model entry points of libcnc_hip.so (include/cnc_hip.h, "Context-model heads
and the Bernoulli rate"): autograd Functions over cnc_ctx_mlp_{forward,backward},
cnc_bernoulli_bits_{forward,backward} and cnc_segment_weighted_sum_backward.  They replace ATen op chains of
examples/utils_bpp_acc.py (no extension exists there), so they are named after what they compute."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import _lib
from .._lib import check, check_input, ptr, stream


def _f32c(t, name):
    check_input(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    return t


_REPLICAS = 16        # zeroed copies of the weight-gradient buffer per backward call


class ContextMLP(Function):
    """y = MLP([in_a | in_b | pg]) per row — `seq` is the reference's nn.Sequential: one Linear (the 2-D
    heads) or Linear-LeakyReLU-Linear-LeakyReLU-Linear (context_model_3D).  in_b / pg may be None."""

    @staticmethod
    def forward(ctx, in_a, in_b, pg, W1, b1, W2, b2, W3, b3, pg_index=None):
        ctx.set_materialize_grads(False)
        in_a = _f32c(in_a.contiguous(), "in_a")
        in_b = None if in_b is None else _f32c(in_b.contiguous(), "in_b")
        pgv = None if pg is None else _f32c(pg.reshape(-1).contiguous(), "pg")
        if pg_index is not None:
            check_input(pg_index, "pg_index")
            if pg_index.dtype != torch.int64 or pg_index.shape[0] != in_a.shape[0] or pgv is None:
                raise RuntimeError("pg_index must be int64 [N] and comes with a pg table")
        elif pgv is not None and pgv.numel() != 1:
            raise RuntimeError("pg must be a scalar unless pg_index is given")
        ws = [_f32c(w.contiguous(), "weight") if w is not None else None for w in (W1, b1, W2, b2, W3, b3)]
        n_layers = 1 if W2 is None else 3
        N, Ca = in_a.shape
        Cb = 0 if in_b is None else in_b.shape[1]
        F = (ws[0] if n_layers == 1 else ws[4]).shape[0]
        if ws[0].shape[1] != Ca + Cb + (pgv is not None):
            raise RuntimeError("context MLP: input width does not match the first layer")
        out = torch.empty((N, F), dtype=torch.float32, device=in_a.device)
        check(_lib.lib().cnc_ctx_mlp_forward(ptr(in_a), Ca, Ca, ptr(in_b), Cb, Cb, ptr(pgv), ptr(pg_index), N, n_layers, F,
                                             *[ptr(w) for w in ws], ptr(out), stream(in_a.device)), "ctx_mlp_forward")
        ctx.save_for_backward(in_a, in_b, pgv, pg_index, *ws)
        ctx.dims = (N, Ca, Cb, n_layers, F, None if pg is None else tuple(pg.shape))
        from .. import _gradsink
        ctx.sink = _gradsink.current()     # the caller thread's sink; the backward runs on autograd's own thread
        return out

    @staticmethod
    def backward(ctx, g):
        in_a, in_b, pgv, pg_index, *ws = ctx.saved_tensors
        N, Ca, Cb, n_layers, F, pg_shape = ctx.dims
        if g is None:
            return (None,) * 10
        g = _f32c(g.contiguous(), "grad_out")
        dev = in_a.device
        g_a = torch.empty_like(in_a)
        g_b = torch.empty_like(in_b) if (in_b is not None and ctx.needs_input_grad[1]) else None
        # every weight / bias gradient (and the Pg gradient behind them) in ONE zero-filled buffer (the kernel
        # accumulates with atomics; _REPLICAS copies: its ~1000 workgroups spread their atomics over them, summed below)
        total = sum(w.numel() for w in ws if w is not None)
        n_pg = 0 if pgv is None else pgv.numel()
        # With a gradient sink on this thread (the training step, cnc_amd._gradsink) the weight gradients of every head
        # call of the step accumulate in the sink's replicated buffer — zeroed once, reduced once — and autograd gets
        # none; only the (tiny) Pg gradient is still a fresh zero-filled vector per call.
        from .. import _gradsink
        sink = ctx.sink
        slots = None if sink is None else sink.small_slot(ws)
        if slots is not None:
            g_pg = torch.zeros(n_pg, dtype=torch.float32, device=dev) if pgv is not None else None
            check(_lib.lib().cnc_ctx_mlp_backward(ptr(in_a), Ca, Ca, ptr(in_b), Cb, Cb, ptr(pgv), ptr(pg_index), N, n_layers, F,
                                                  *[ptr(w) for w in ws], ptr(g), ptr(g_a), ptr(g_b), ptr(g_pg),
                                                  *[ptr(w) for w in slots], _gradsink.REPLICAS, sink.stride(), 0, 0, stream(dev)),
                  "ctx_mlp_backward")
            return (g_a, g_b, None if g_pg is None else g_pg.reshape(pg_shape)) + (None,) * 7
        zeroed = torch.zeros(_REPLICAS * total + n_pg, dtype=torch.float32, device=dev)
        copies = zeroed[:_REPLICAS * total].view(_REPLICAS, total)
        g_pg = zeroed[_REPLICAS * total:] if pgv is not None else None
        first, at = [], 0
        for w in ws:
            first.append(None if w is None else copies[0, at:at + w.numel()])
            at += 0 if w is None else w.numel()
        check(_lib.lib().cnc_ctx_mlp_backward(ptr(in_a), Ca, Ca, ptr(in_b), Cb, Cb, ptr(pgv), ptr(pg_index), N, n_layers, F,
                                              *[ptr(w) for w in ws], ptr(g), ptr(g_a), ptr(g_b), ptr(g_pg),
                                              *[ptr(w) for w in first], _REPLICAS, total, 0, 0, stream(dev)), "ctx_mlp_backward")
        flat = copies.sum(0) if _REPLICAS > 1 else copies[0]
        gws, at = [], 0
        for w in ws:
            gws.append(None if w is None else flat[at:at + w.numel()].view_as(w))
            at += 0 if w is None else w.numel()
        return (g_a, g_b, None if g_pg is None else g_pg.reshape(pg_shape), *gws, None)


class ContextHeads(Function):
    """Several one-layer heads on row ranges of ONE input matrix (the coded levels of a plane, utils_bpp_acc.py:556-566,
    evaluated together): rows [r0, r1) of segment i see the columns [c0, c0 + Ca) of in_a, all of in_b and pg[pg_i],
    through Linear(W_i, b_i).  One output matrix, one gradient matrix per input (columns outside a segment's window get
    zero), per-head weight gradients — no slicing nodes, no concatenation."""

    @staticmethod
    def forward(ctx, in_a, in_b, pg, segs, *wb):
        ctx.set_materialize_grads(False)
        in_a = _f32c(in_a.contiguous(), "in_a")
        in_b = None if in_b is None else _f32c(in_b.contiguous(), "in_b")
        pgv = None if pg is None else _f32c(pg.reshape(-1).contiguous(), "pg")
        ws = [_f32c(w.contiguous(), "weight") for w in wb]
        if len(ws) != 2 * len(segs):
            raise RuntimeError("ContextHeads: one (weight, bias) pair per segment")
        N, lda = in_a.shape
        Cb = 0 if in_b is None else in_b.shape[1]
        F = ws[0].shape[0]
        out = torch.empty((N, F), dtype=torch.float32, device=in_a.device)
        L, st = _lib.lib(), stream(in_a.device)
        at = 0
        for i, (r0, r1, c0, Ca, pg_i) in enumerate(segs):
            W, b = ws[2 * i], ws[2 * i + 1]
            if r0 != at or r1 < r0 or c0 + Ca > lda or W.shape != (F, Ca + Cb + (pgv is not None)) or b.shape != (F,):
                raise RuntimeError("ContextHeads: segments must tile the rows in order and fit their heads")
            at = r1
            if r1 == r0:
                continue
            check(L.cnc_ctx_mlp_forward(in_a.data_ptr() + 4 * (r0 * lda + c0), lda, Ca,
                                        None if in_b is None else in_b.data_ptr() + 4 * r0 * Cb, Cb, Cb,
                                        None if pgv is None else pgv.data_ptr() + 4 * pg_i, None, r1 - r0, 1, F,
                                        ptr(W), ptr(b), None, None, None, None, out.data_ptr() + 4 * r0 * F, st),
                  "ctx_mlp_forward")
        if at != N:
            raise RuntimeError("ContextHeads: segments must cover every row")
        ctx.save_for_backward(in_a, in_b, pgv, *ws)
        ctx.segs, ctx.pg_shape = tuple(segs), None if pg is None else tuple(pg.shape)
        from .. import _gradsink
        ctx.sink = _gradsink.current()
        return out

    @staticmethod
    def backward(ctx, g):
        in_a, in_b, pgv, *ws = ctx.saved_tensors
        none = (None,) * (4 + len(ws))
        if g is None:
            return none
        g = _f32c(g.contiguous(), "grad_out")
        dev = in_a.device
        N, lda = in_a.shape
        Cb = 0 if in_b is None else in_b.shape[1]
        F = ws[0].shape[0]
        g_a = torch.empty_like(in_a)
        for (r0, r1, c0, Ca, _) in ctx.segs:          # columns outside a segment's window: zero (only those are filled)
            if r1 > r0 and c0 > 0:
                g_a[r0:r1, :c0].zero_()
            if r1 > r0 and c0 + Ca < lda:
                g_a[r0:r1, c0 + Ca:].zero_()
        g_b = torch.empty_like(in_b) if (in_b is not None and ctx.needs_input_grad[1]) else None
        g_pg = torch.zeros_like(pgv) if pgv is not None else None
        from .. import _gradsink
        sink = ctx.sink
        slots = None if sink is None else sink.small_slot(ws)
        if slots is not None:
            firsts, reps, stride = slots, _gradsink.REPLICAS, sink.stride()
        else:
            total = sum(w.numel() for w in ws)
            copies = torch.zeros((_REPLICAS, total), dtype=torch.float32, device=dev)
            firsts, o = [], 0
            for w in ws:
                firsts.append(copies[0, o:o + w.numel()])
                o += w.numel()
            reps, stride = _REPLICAS, total
        L, st = _lib.lib(), stream(dev)
        for i, (r0, r1, c0, Ca, pg_i) in enumerate(ctx.segs):
            if r1 == r0:
                continue
            check(L.cnc_ctx_mlp_backward(in_a.data_ptr() + 4 * (r0 * lda + c0), lda, Ca,
                                         None if in_b is None else in_b.data_ptr() + 4 * r0 * Cb, Cb, Cb,
                                         None if pgv is None else pgv.data_ptr() + 4 * pg_i, None, r1 - r0, 1, F,
                                         ptr(ws[2 * i]), ptr(ws[2 * i + 1]), None, None, None, None,
                                         g.data_ptr() + 4 * r0 * F, g_a.data_ptr() + 4 * (r0 * lda + c0),
                                         None if g_b is None else g_b.data_ptr() + 4 * r0 * Cb,
                                         None if g_pg is None else g_pg.data_ptr() + 4 * pg_i,
                                         ptr(firsts[2 * i]), ptr(firsts[2 * i + 1]), None, None, None, None,
                                         reps, stride, lda, Cb, st), "ctx_mlp_backward")
        g_pg_out = None if g_pg is None else g_pg.reshape(ctx.pg_shape)
        if slots is not None:
            return (g_a, g_b, g_pg_out, None) + (None,) * len(ws)
        flat = copies.sum(0)
        gws, o = [], 0
        for w in ws:
            gws.append(flat[o:o + w.numel()].view_as(w))
            o += w.numel()
        return (g_a, g_b, g_pg_out, None, *gws)


def context_heads(heads, in_a, in_b, pg, segs):
    """`ContextHeads` over nn.Linear modules `heads` (one per segment); segs = [(row0, row1, col0, n_cols, pg_index)]."""
    wb = []
    for h in heads:
        lin = [m for m in h if isinstance(m, torch.nn.Linear)] if isinstance(h, torch.nn.Sequential) else [h]
        if len(lin) != 1:
            raise RuntimeError("context_heads: every head is one Linear")
        wb += [lin[0].weight, lin[0].bias]
    return ContextHeads.apply(in_a, in_b, pg, tuple(tuple(int(v) for v in s) for s in segs), *wb)


def context_mlp(seq, in_a, in_b=None, pg=None, pg_index=None):
    """Apply an nn.Sequential of Linear / LeakyReLU layers (1 or 3 Linear) through the fused kernel.  The input
    row is [in_a | in_b | pg]; with `pg_index` (int64 [N]) pg is a table and row i takes pg[pg_index[i]]."""
    lin = [m for m in seq if isinstance(m, torch.nn.Linear)] if isinstance(seq, torch.nn.Sequential) else [seq]
    if len(lin) == 1:
        return ContextMLP.apply(in_a, in_b, pg, lin[0].weight, lin[0].bias, None, None, None, None, pg_index)
    if len(lin) == 3 and lin[0].out_features == 32 and lin[1].out_features == 32:
        return ContextMLP.apply(in_a, in_b, pg, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias,
                                lin[2].weight, lin[2].bias, pg_index)
    raise RuntimeError("context_mlp: expected Linear(C,F) or Linear(C,32)-LeakyReLU-Linear(32,32)-LeakyReLU-Linear(32,F)")


class BernoulliBits(Function):
    """sum of Bernoulli_entropy(table[rows], mean) — the gather, the clamp / log2 / mask arithmetic and the
    reduction in one kernel; gradient w.r.t. the table (dense, zero off `rows`) and the means."""

    @staticmethod
    def forward(ctx, table, rows, mean):
        table, mean = _f32c(table.contiguous(), "table"), _f32c(mean.contiguous(), "mean")
        if rows is not None:
            check_input(rows, "rows")
            if rows.dtype != torch.int64:
                raise RuntimeError("rows must be int64")
        S, F = mean.shape
        L = _lib.lib()
        partial = torch.empty(int(L.cnc_bernoulli_bits_partials(S, F)), dtype=torch.float32, device=mean.device)
        check(L.cnc_bernoulli_bits_forward(ptr(table), ptr(rows), ptr(mean) if S else None, S, F, ptr(partial),
                                           stream(mean.device)), "bernoulli_bits_forward")
        ctx.save_for_backward(table, rows, mean)
        return partial.sum()

    @staticmethod
    def backward(ctx, g):
        table, rows, mean = ctx.saved_tensors
        S, F = mean.shape
        g = g.reshape(1).to(torch.float32).contiguous()
        need_t, need_m = ctx.needs_input_grad[0], ctx.needs_input_grad[2]
        g_mean = torch.empty_like(mean) if need_m else None
        g_x = torch.empty_like(mean) if need_t else None
        check(_lib.lib().cnc_bernoulli_bits_backward(ptr(table), ptr(rows), ptr(mean), ptr(g), S, F, ptr(g_mean),
                                                     ptr(g_x), stream(mean.device)), "bernoulli_bits_backward")
        g_table = None
        if need_t:
            if rows is None:
                g_table = g_x
            else:            # rows are distinct hash slots: a plain scatter into zeros (one small kernel: the library's
                g_table = torch.zeros_like(table)          # index_put took 0.28 ms for these 1.5e5 rows)
                check(_lib.lib().cnc_rows_scatter(ptr(g_x), ptr(rows), ptr(g_table), S, F, stream(mean.device)), "rows_scatter")
        return g_table, None, g_mean


def bernoulli_bits(table, rows, mean):
    return BernoulliBits.apply(table, rows, mean)


def segment_backward(g, cumsum, weights, wsum, T, mode, order=None):
    """d values of cnc_segment_weighted_sum{,_gathered} (pack_and_align.segment_weighted_sum)."""
    g = _f32c(g.contiguous(), "grad")
    S, F = g.shape
    out = torch.empty((T, F), dtype=torch.float32, device=g.device)
    check(_lib.lib().cnc_segment_weighted_sum_gathered_backward(ptr(g), ptr(order), ptr(cumsum), ptr(weights), ptr(wsum), S, T,
                                                                F, int(mode), ptr(out), stream(g.device)),
          "segment_weighted_sum_backward")
    return out


class LevelStats(Function):
    """(Pg [L], bits [L]) of every level of a binarised table in one pass (cnc_level_stats_*)."""

    @staticmethod
    def forward(ctx, table, off_host):
        import ctypes as C
        ctx.set_materialize_grads(False)
        table = _f32c(table.contiguous(), "table")
        L = len(off_host) - 1
        F, dev = table.shape[1], table.device
        offs = (C.c_int64 * (L + 1))(*[int(o) for o in off_host])
        sums = torch.empty(L, dtype=torch.float64, device=dev)
        Pg = torch.empty(L, dtype=torch.float32, device=dev)
        bits = torch.empty(L, dtype=torch.float32, device=dev)
        check(_lib.lib().cnc_level_stats_forward(ptr(table), C.cast(offs, C.c_void_p), L, F, ptr(sums), ptr(Pg), ptr(bits),
                                                 stream(dev)), "level_stats_forward")
        ctx.save_for_backward(sums)
        ctx.meta = (offs, L, F, table.shape[0])
        return Pg, bits

    @staticmethod
    def backward(ctx, g_Pg, g_bits):
        import ctypes as C
        (sums,) = ctx.saved_tensors
        offs, L, F, rows = ctx.meta
        if g_Pg is None and g_bits is None:
            return None, None
        g = torch.empty((rows, F), dtype=torch.float32, device=sums.device)
        gp = None if g_Pg is None else g_Pg.contiguous()
        gb = None if g_bits is None else g_bits.contiguous()
        check(_lib.lib().cnc_level_stats_backward(ptr(sums), C.cast(offs, C.c_void_p), L, F, ptr(gp), ptr(gb), rows,
                                                  ptr(g), stream(sums.device)), "level_stats_backward")
        return g, None


def level_stats(table, off_host):
    return LevelStats.apply(table, tuple(off_host))


def plane_ring_vertices(cells, T, resolution, hashmap_size):
    """cells i32 [M, 2] (occupied cells of a projected occupancy plane) -> (rows i32 [M (T+2)^2], points f32 [., 2]) of
    the 2-D level's vertices inside / one ring around them — `fetch_2D_batches` (utils_bpp_acc.py:431-456) as one
    kernel (cnc_plane_ring_vertices)."""
    if cells.dim() != 2 or cells.shape[1] != 2 or cells.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("plane_ring_vertices: cells must be int32 [M, 2]")
    cells = cells.to(torch.int32).contiguous()
    check_input(cells, "cells")
    n = cells.shape[0] * (int(T) + 2) ** 2
    rows = torch.empty(n, dtype=torch.int32, device=cells.device)
    points = torch.empty((n, 2), dtype=torch.float32, device=cells.device)
    check(_lib.lib().cnc_plane_ring_vertices(ptr(cells), cells.shape[0], int(T), int(resolution), int(hashmap_size),
                                             ptr(rows), ptr(points), stream(cells.device)), "plane_ring_vertices")
    return rows, points


def compact_masked(idx, pts_n, level_ids, overlap, L):
    """(pts_n[idx], level_ids[idx], (level_ids[idx] - L) as int32, clamp(overlap[idx], min = 1) as float32 or None) in one
    kernel (cnc_ctx_compact); idx int64 [M] ascending, overlap int32 [N] or None."""
    for name, t in (("idx", idx), ("pts_n", pts_n), ("level_ids", level_ids)):
        check_input(t, name)
    if idx.dtype != torch.int64 or level_ids.dtype != torch.int64 or pts_n.dtype != torch.float32 or pts_n.shape[-1] != 3 \
            or (overlap is not None and overlap.dtype != torch.int32):
        raise RuntimeError("compact_masked: idx / level_ids int64, pts_n float32 [N, 3], overlap int32")
    M, dev = idx.shape[0], idx.device
    pts_m = torch.empty((M, 3), dtype=torch.float32, device=dev)
    lvl_m = torch.empty(M, dtype=torch.int64, device=dev)
    min_l = torch.empty(M, dtype=torch.int32, device=dev)
    ow = torch.empty(M, dtype=torch.float32, device=dev) if overlap is not None else None
    check(_lib.lib().cnc_ctx_compact(ptr(idx), ptr(pts_n), ptr(level_ids), ptr(overlap), M, int(L), ptr(pts_m), ptr(lvl_m),
                                     ptr(min_l), ptr(ow), stream(dev)), "ctx_compact")
    return pts_m, lvl_m, min_l, ow


def window_gather(levels, device):
    """levels: list of dicts {pos (int16 [*,3] window slice), cnt, val (int64 window slices), level, res, row0}.
    Returns (pts i16 [P,3], pts_n f32 [P,3], level_ids i64 [P], resolutions i64 [P], slot_counts i64 [V],
    table_rows i64 [V]) concatenated over the levels — one kernel (cnc_ctx_window_gather)."""
    import ctypes as C
    w = _lib.CtxWindow()
    w.n_win = len(levels)
    P = V = 0
    for i, lv in enumerate(levels):
        for name in ("pos", "cnt", "val"):
            check_input(lv[name], name)
        w.pos[i], w.cnt[i], w.val[i] = lv["pos"].data_ptr(), lv["cnt"].data_ptr(), lv["val"].data_ptr()
        w.p_at[i], w.v_at[i] = P, V
        w.row0[i], w.level[i], w.res[i] = int(lv["row0"]), int(lv["level"]), int(lv["res"])
        P += lv["pos"].shape[0]
        V += lv["cnt"].shape[0]
    w.p_at[len(levels)], w.v_at[len(levels)] = P, V
    pts = torch.empty((P, 3), dtype=torch.int16, device=device)
    pts_n = torch.empty((P, 3), dtype=torch.float32, device=device)
    lvl = torch.empty(P, dtype=torch.int64, device=device)
    res = torch.empty(P, dtype=torch.int64, device=device)
    cnts = torch.empty(V, dtype=torch.int64, device=device)
    rows = torch.empty(V, dtype=torch.int64, device=device)
    check(_lib.lib().cnc_ctx_window_gather(C.byref(w), ptr(pts), ptr(pts_n), ptr(lvl), ptr(res), ptr(cnts), ptr(rows),
                                           stream(device)), "ctx_window_gather")
    return pts, pts_n, lvl, res, cnts, rows
