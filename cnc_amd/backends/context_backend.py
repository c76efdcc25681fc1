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


class ContextMLP(Function):
    """y = MLP([in_a | in_b | pg]) per row — `seq` is the reference's nn.Sequential: one Linear (the 2-D
    heads) or Linear-LeakyReLU-Linear-LeakyReLU-Linear (context_model_3D).  in_b / pg may be None."""

    @staticmethod
    def forward(ctx, in_a, in_b, pg, W1, b1, W2, b2, W3, b3):
        ctx.set_materialize_grads(False)
        in_a = _f32c(in_a.contiguous(), "in_a")
        in_b = None if in_b is None else _f32c(in_b.contiguous(), "in_b")
        pgv = None if pg is None else _f32c(pg.reshape(1).contiguous(), "pg")
        ws = [_f32c(w.contiguous(), "weight") if w is not None else None for w in (W1, b1, W2, b2, W3, b3)]
        n_layers = 1 if W2 is None else 3
        N, Ca = in_a.shape
        Cb = 0 if in_b is None else in_b.shape[1]
        F = (ws[0] if n_layers == 1 else ws[4]).shape[0]
        if ws[0].shape[1] != Ca + Cb + (pgv is not None):
            raise RuntimeError("context MLP: input width does not match the first layer")
        out = torch.empty((N, F), dtype=torch.float32, device=in_a.device)
        check(_lib.lib().cnc_ctx_mlp_forward(ptr(in_a), Ca, Ca, ptr(in_b), Cb, Cb, ptr(pgv), N, n_layers, F,
                                             *[ptr(w) for w in ws], ptr(out), stream(in_a.device)), "ctx_mlp_forward")
        ctx.save_for_backward(in_a, in_b, pgv, *ws)
        ctx.dims = (N, Ca, Cb, n_layers, F, None if pg is None else tuple(pg.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        in_a, in_b, pgv, *ws = ctx.saved_tensors
        N, Ca, Cb, n_layers, F, pg_shape = ctx.dims
        if g is None:
            return (None,) * 9
        g = _f32c(g.contiguous(), "grad_out")
        dev = in_a.device
        g_a = torch.empty_like(in_a)
        g_b = torch.empty_like(in_b) if (in_b is not None and ctx.needs_input_grad[1]) else None
        g_pg = torch.zeros(1, dtype=torch.float32, device=dev) if pgv is not None else None
        # every weight / bias gradient in ONE zero-filled buffer (the kernel accumulates with atomics)
        flat = torch.zeros(sum(w.numel() for w in ws if w is not None), dtype=torch.float32, device=dev)
        gws, at = [], 0
        for w in ws:
            gws.append(None if w is None else flat[at:at + w.numel()].view_as(w))
            at += 0 if w is None else w.numel()
        check(_lib.lib().cnc_ctx_mlp_backward(ptr(in_a), Ca, Ca, ptr(in_b), Cb, Cb, ptr(pgv), N, n_layers, F,
                                              *[ptr(w) for w in ws], ptr(g), ptr(g_a), ptr(g_b), ptr(g_pg),
                                              *[ptr(w) for w in gws], stream(dev)), "ctx_mlp_backward")
        return (g_a, g_b, None if g_pg is None else g_pg.reshape(pg_shape), *gws)


def context_mlp(seq, in_a, in_b=None, pg=None):
    """Apply an nn.Sequential of Linear / LeakyReLU layers (1 or 3 Linear) through the fused kernel."""
    lin = [m for m in seq if isinstance(m, torch.nn.Linear)] if isinstance(seq, torch.nn.Sequential) else [seq]
    if len(lin) == 1:
        return ContextMLP.apply(in_a, in_b, pg, lin[0].weight, lin[0].bias, None, None, None, None)
    if len(lin) == 3 and lin[0].out_features == 32 and lin[1].out_features == 32:
        return ContextMLP.apply(in_a, in_b, pg, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias,
                                lin[2].weight, lin[2].bias)
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
            else:            # rows are distinct hash slots: a plain scatter into zeros
                g_table = torch.zeros_like(table)
                g_table[rows] = g_x
        return g_table, None, g_mean


def bernoulli_bits(table, rows, mean):
    return BernoulliBits.apply(table, rows, mean)


def segment_backward(g, cumsum, weights, wsum, T, mode):
    """d values of cnc_segment_weighted_sum (pack_and_align.segment_weighted_sum)."""
    g = _f32c(g.contiguous(), "grad")
    S, F = g.shape
    out = torch.empty((T, F), dtype=torch.float32, device=g.device)
    check(_lib.lib().cnc_segment_weighted_sum_backward(ptr(g), ptr(cumsum), ptr(weights), ptr(wsum), S, T, F, int(mode),
                                                       ptr(out), stream(g.device)), "segment_weighted_sum_backward")
    return out
