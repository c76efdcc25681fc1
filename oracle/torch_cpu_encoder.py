"""TEST INFRASTRUCTURE / CPU BASELINE ONLY — never imported by the product (cnc_amd/).

A pure-PyTorch-CPU `GridEncoder` forward + backward: index math on int64 tensors, `index_select`
gathers, a weighted sum over the 2^D corners, and autograd's `index_add_` for the embedding gradient.
This is what BASELINE.json calls the "PyTorch-CPU gridencoder fallback" of config 1 (16 levels,
log2T=19, F=2, 4096 rays per batch).  The reference itself has no CPU path (gridencoder.cu:15,764-768
assert CUDA tensors), so the fallback is restated here from the same lines as oracle/cnc_oracle.c:

  * corner set-up, validity and weight renormalisation: gridencoder.cu:160-291
      pos = x * (R - 2) + 0.5 (the 0.5 is a double literal), corners clamped to R - 1, corners on the
      border ring (0 or R - 1) dropped, weights divided by the sum of the surviving ones;
  * row index: gridencoder.cu:45-87 (dense stride walk or xor of prime products, then % rows);
  * STE binarisation: ngp.py:22-39 (forward sign with 0 -> +1, backward mask |p| <= 1).

No occupancy mask (`binary_vxl=None`) and no per-point level window: the micro-bench configuration.
It is pinned against the C restatement in tests/test_oracle_pins.py (bit-exact forward with the
binarised table — the products by +-1 are exact, so the missing fmaf cannot show).
"""
from __future__ import annotations

import torch

_PRIMES = (1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737)
_U32 = 0xFFFFFFFF


def _rows(q, hashmap_size: int, resolution: int):
    """q: list of D int64 tensors (corner coordinates) -> int64 row index, gridencoder.cu:45-87."""
    D = len(q)
    stride, dense = 1, True
    index = torch.zeros_like(q[0])
    for d in range(D):                       # :72-77, uint32 arithmetic
        if stride > hashmap_size:
            break
        index = (index + q[d] * stride) & _U32
        stride = (stride * resolution) & _U32
    if stride > hashmap_size:                # :80-82
        dense = False
    if not dense:
        index = torch.zeros_like(q[0])
        for d in range(D):
            index = index ^ ((q[d] * _PRIMES[d]) & _U32)
    return index % hashmap_size


class _STE(torch.autograd.Function):
    """ngp.py:22-39"""

    @staticmethod
    def forward(ctx, p):
        ctx.save_for_backward(p)
        return torch.where(p >= 0, torch.ones_like(p), -torch.ones_like(p))

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        return g * ((p >= -1) & (p <= 1)).to(g.dtype)


def grid_encode(inputs: torch.Tensor, embeddings: torch.Tensor, offsets, resolutions,
                ste_binary: bool = True) -> torch.Tensor:
    """inputs [N, D] float32 in [0, 1]; embeddings [rows, F]; returns [L, N, F] (kernel_grid layout).
    Differentiable with respect to `embeddings`."""
    N, D = inputs.shape
    table = _STE.apply(embeddings) if ste_binary else embeddings
    # one view per level: each gather's gradient is then a level-sized index_add_, not a table-sized one
    sizes = [int(offsets[l + 1]) - int(offsets[l]) for l in range(len(resolutions))]
    rest = table.shape[0] - sum(sizes)
    levels = torch.split(table, sizes + ([rest] if rest else []), dim=0)
    inside = ((inputs >= 0) & (inputs <= 1)).all(dim=1)                      # :134-140
    outs = []
    for level, R in enumerate(int(r) for r in resolutions):
        off = int(offsets[level])
        hs = int(offsets[level + 1]) - off
        prod = inputs * float(R - 2)                                          # float * float, :173
        pos = (prod.double() + 0.5).float()
        g = torch.floor(pos)
        frac = pos - g
        g = g.long()
        ws, rows, valids = [], [], []
        wn = torch.zeros(N, dtype=torch.float32)
        for i in range(1 << D):
            w = torch.ones(N, dtype=torch.float32)
            q = []
            for d in range(D):                                                # :200-208
                if (i >> d) & 1:
                    w = w * frac[:, d]
                    q.append(torch.clamp(g[:, d] + 1, max=R - 1))
                else:
                    w = w * (1 - frac[:, d])
                    q.append(g[:, d])
            border = torch.zeros(N, dtype=torch.bool)
            for d in range(D):                                                # :212-219
                border |= (q[d] == 0) | (q[d] == R - 1)
            valid = ~border
            row = torch.where(valid, _rows(q, hs, R), torch.zeros_like(q[0]))
            wn = torch.where(valid, wn + w, wn)                               # :281-285, corner order
            ws.append(w)
            rows.append(row)
            valids.append(valid)
        wn = torch.where(wn == 0, (wn.double() + 1e-9).float(), wn)           # :288-290
        wn_re = (1.0 / wn.double()).float()                                   # :291
        out = torch.zeros((N, table.shape[1]), dtype=torch.float32)
        for w, row, valid in zip(ws, rows, valids):                           # :294-303
            t = torch.where(valid & inside, w * wn_re, torch.zeros_like(w))
            out = out + t[:, None] * torch.index_select(levels[level], 0, row)
        outs.append(out)
    return torch.stack(outs, 0)


def forward_backward(inputs, embeddings, offsets, resolutions, grad_out, ste_binary=True):
    """One forward + embedding-gradient pass; returns (outputs [L, N, F], grad_embeddings)."""
    emb = embeddings.detach().clone().requires_grad_(True)
    out = grid_encode(inputs, emb, offsets, resolutions, ste_binary)
    out.backward(grad_out)
    return out.detach(), emb.grad
