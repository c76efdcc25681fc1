"""Segmented scans over flattened per-ray samples (reference: nerfacc/scan.py:12-288).

`packed_info=None` means a batched tensor scanned along its last dimension with plain torch ops
(same as the reference); with `packed_info` (n_rays, 2) the input is a flattened 1-D tensor and
the HIP kernel (cnc_amd/csrc/scan.hip) runs, with an autograd rule that is the reverse scan.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import cuda as _C


def _unpack(inputs, packed_info):
    assert inputs.dim() == 1, "inputs must be flattened."
    assert packed_info.dim() == 2 and packed_info.shape[-1] == 2, \
        "packed_info must be 2-D with shape (B, 2)."
    return packed_info.unbind(dim=-1)


def inclusive_sum(inputs: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    """[1..9] with chunks (0,2),(2,3),(5,4) -> [1,3, 3,7,12, 6,13,21,30]."""
    if packed_info is None:
        return torch.cumsum(inputs, dim=-1)
    starts, cnts = _unpack(inputs, packed_info)
    return _InclusiveSum.apply(starts, cnts, inputs, False)


def exclusive_sum(inputs: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    """[1..9] with chunks (0,2),(2,3),(5,4) -> [0,1, 0,3,7, 0,6,13,21]."""
    if packed_info is None:
        shifted = torch.cat([torch.zeros_like(inputs[..., :1]), inputs[..., :-1]], dim=-1)
        return torch.cumsum(shifted, dim=-1)
    starts, cnts = _unpack(inputs, packed_info)
    return _ExclusiveSum.apply(starts, cnts, inputs, False)


def inclusive_prod(inputs: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    """[1..9] with chunks (0,2),(2,3),(5,4) -> [1,2, 3,12,60, 6,42,336,3024]."""
    if packed_info is None:
        return torch.cumprod(inputs, dim=-1)
    starts, cnts = _unpack(inputs, packed_info)
    return _InclusiveProd.apply(starts, cnts, inputs)


def exclusive_prod(inputs: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    """[1..9] with chunks (0,2),(2,3),(5,4) -> [1,1, 1,3,12, 1,6,42,336]."""
    if packed_info is None:
        shifted = torch.cat([torch.ones_like(inputs[..., :1]), inputs[..., :-1]], dim=-1)
        return torch.cumprod(shifted, dim=-1)
    starts, cnts = packed_info.unbind(dim=-1)
    return _ExclusiveProd.apply(starts, cnts, inputs)


def _make_sum(name, kernel):
    class _Sum(torch.autograd.Function):
        @staticmethod
        def forward(ctx, chunk_starts, chunk_cnts, inputs, normalize: bool = False):
            chunk_starts, chunk_cnts, inputs = (
                t.contiguous() for t in (chunk_starts, chunk_cnts, inputs))
            outputs = kernel(chunk_starts, chunk_cnts, inputs, normalize, False)
            if ctx.needs_input_grad[2]:
                ctx.normalize = normalize
                ctx.save_for_backward(chunk_starts, chunk_cnts)
            return outputs

        @staticmethod
        def backward(ctx, grad_outputs):
            chunk_starts, chunk_cnts = ctx.saved_tensors
            assert ctx.normalize is False, "Only support backward for normalize==False."
            # d/dx of a prefix sum is the suffix sum of the incoming gradient: same kernel,
            # chunks scanned right-to-left
            grad_inputs = kernel(chunk_starts, chunk_cnts, grad_outputs.contiguous(), False, True)
            return None, None, grad_inputs, None

    _Sum.__name__ = _Sum.__qualname__ = name
    return _Sum


def _make_prod(name, fwd, bwd):
    class _Prod(torch.autograd.Function):
        @staticmethod
        def forward(ctx, chunk_starts, chunk_cnts, inputs):
            chunk_starts, chunk_cnts, inputs = (
                t.contiguous() for t in (chunk_starts, chunk_cnts, inputs))
            outputs = fwd(chunk_starts, chunk_cnts, inputs)
            if ctx.needs_input_grad[2]:
                ctx.save_for_backward(chunk_starts, chunk_cnts, inputs, outputs)
            return outputs

        @staticmethod
        def backward(ctx, grad_outputs):
            chunk_starts, chunk_cnts, inputs, outputs = ctx.saved_tensors
            grad_inputs = bwd(chunk_starts, chunk_cnts, inputs, outputs, grad_outputs.contiguous())
            return None, None, grad_inputs

    _Prod.__name__ = _Prod.__qualname__ = name
    return _Prod


_InclusiveSum = _make_sum("_InclusiveSum", _C.inclusive_sum)
_ExclusiveSum = _make_sum("_ExclusiveSum", _C.exclusive_sum)
_InclusiveProd = _make_prod("_InclusiveProd", _C.inclusive_prod_forward, _C.inclusive_prod_backward)
_ExclusiveProd = _make_prod("_ExclusiveProd", _C.exclusive_prod_forward, _C.exclusive_prod_backward)
