"""Prefix sums / products along rays (reference: nerfacc/scan.py:12-288): `inclusive_sum`, `exclusive_sum`,
`inclusive_prod`, `exclusive_prod`.

Batched input (no `packed_info`) is scanned along its last axis with plain torch ops on any device.  Flattened
input (1-D, with `packed_info` (n_rays, 2) = (start, count) per ray) goes through the segmented-scan kernel
(cnc_amd/csrc/scan.hip, same 32-wide tile tree and therefore the same float association as the reference's CUDA
kernel); its autograd rule is the same kernel run right-to-left over each ray.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import cuda as _C


def _shift_in(x: Tensor, fill: float) -> Tensor:
    """x moved one place to the right along the last axis, `fill` entering on the left."""
    return torch.nn.functional.pad(x[..., :-1], (1, 0), value=fill)


def _ray_chunks(inputs: Tensor, packed_info: Tensor):
    if inputs.dim() != 1:
        raise AssertionError("inputs must be flattened.")
    if packed_info.dim() != 2 or packed_info.shape[-1] != 2:
        raise AssertionError("packed_info must be 2-D with shape (B, 2).")
    return packed_info[:, 0].contiguous(), packed_info[:, 1].contiguous()


class _SegmentedSum(torch.autograd.Function):
    """Sum scan over ray chunks; d/dx of a prefix sum is the suffix sum of the incoming gradient."""

    @staticmethod
    def forward(ctx, starts, counts, x, exclusive):
        ctx.exclusive = exclusive
        ctx.save_for_backward(starts, counts)
        kernel = _C.exclusive_sum if exclusive else _C.inclusive_sum
        return kernel(starts, counts, x.contiguous(), False, False)

    @staticmethod
    def backward(ctx, g):
        starts, counts = ctx.saved_tensors
        kernel = _C.exclusive_sum if ctx.exclusive else _C.inclusive_sum
        return None, None, kernel(starts, counts, g.contiguous(), False, True), None


class _SegmentedProd(torch.autograd.Function):
    """Product scan over ray chunks; the backward kernel is reverse-scan-sum(g * y) / max(x, 1e-10)."""

    @staticmethod
    def forward(ctx, starts, counts, x, exclusive):
        x = x.contiguous()
        y = (_C.exclusive_prod_forward if exclusive else _C.inclusive_prod_forward)(starts, counts, x)
        ctx.exclusive = exclusive
        ctx.save_for_backward(starts, counts, x, y)
        return y

    @staticmethod
    def backward(ctx, g):
        starts, counts, x, y = ctx.saved_tensors
        kernel = _C.exclusive_prod_backward if ctx.exclusive else _C.inclusive_prod_backward
        return None, None, kernel(starts, counts, x, y, g.contiguous()), None


def inclusive_sum(inputs: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    """[1..9] in chunks (0,2),(2,3),(5,4) -> [1,3, 3,7,12, 6,13,21,30]."""
    if packed_info is None:
        return inputs.cumsum(dim=-1)
    return _SegmentedSum.apply(*_ray_chunks(inputs, packed_info), inputs, False)


def exclusive_sum(inputs: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    """[1..9] in chunks (0,2),(2,3),(5,4) -> [0,1, 0,3,7, 0,6,13,21]."""
    if packed_info is None:
        return _shift_in(inputs, 0.0).cumsum(dim=-1)
    return _SegmentedSum.apply(*_ray_chunks(inputs, packed_info), inputs, True)


def inclusive_prod(inputs: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    """[1..9] in chunks (0,2),(2,3),(5,4) -> [1,2, 3,12,60, 6,42,336,3024]."""
    if packed_info is None:
        return inputs.cumprod(dim=-1)
    return _SegmentedProd.apply(*_ray_chunks(inputs, packed_info), inputs, False)


def exclusive_prod(inputs: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    """[1..9] in chunks (0,2),(2,3),(5,4) -> [1,1, 1,3,12, 1,6,42,336]."""
    if packed_info is None:
        return _shift_in(inputs, 1.0).cumprod(dim=-1)
    return _SegmentedProd.apply(*_ray_chunks(inputs, packed_info), inputs, True)
