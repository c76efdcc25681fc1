"""pack_info (reference: nerfacc/pack.py:11-49)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor


@torch.no_grad()
def pack_info(ray_indices: Tensor, n_rays: Optional[int] = None) -> Tensor:
    """Sorted per-sample ray indices -> (n_rays, 2) LongTensor of (start, count) per ray.

    >>> pack_info(torch.tensor([0, 0, 1, 1, 1, 2, 2, 2, 2]), n_rays=3)
    tensor([[0, 2], [2, 3], [5, 4]])

    The reference only accepts CUDA tensors (pack.py:38-48); the same torch ops run on any
    device, so no such restriction is imposed here.
    """
    assert ray_indices.dim() == 1, "ray_indices must be a 1D tensor with shape (n_samples)."
    if n_rays is None:
        n_rays = int(ray_indices.max().item()) + 1 if ray_indices.numel() else 0
    cnts = torch.zeros((n_rays,), device=ray_indices.device, dtype=ray_indices.dtype)
    cnts.index_add_(0, ray_indices, torch.ones_like(ray_indices))
    starts = cnts.cumsum(dim=0, dtype=ray_indices.dtype) - cnts
    return torch.stack([starts, cnts], dim=-1)
