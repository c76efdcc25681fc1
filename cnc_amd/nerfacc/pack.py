"""pack_info (reference: nerfacc/pack.py:11-49): per-ray (start, count) of flattened, ray-sorted samples."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor


@torch.no_grad()
def pack_info(ray_indices: Tensor, n_rays: Optional[int] = None) -> Tensor:
    """Sorted per-sample ray ids -> LongTensor (n_rays, 2) holding (first sample, number of samples).

    >>> pack_info(torch.tensor([0, 0, 1, 1, 1, 2, 2, 2, 2]), n_rays=3)
    tensor([[0, 2], [2, 3], [5, 4]])

    On the GPU the run boundaries are found by one kernel (cnc_pack_bounds); elsewhere a bincount does it
    (the reference insists on CUDA tensors, pack.py:38-48 — no need to here)."""
    if ray_indices.dim() != 1:
        raise AssertionError("ray_indices must be a 1D tensor with shape (n_samples).")
    if n_rays is None:
        n_rays = int(ray_indices.max().item()) + 1 if ray_indices.numel() else 0
    if ray_indices.is_cuda and ray_indices.dtype == torch.int64:
        from ..backends.volrend_backend import pack_bounds
        starts, counts = pack_bounds(ray_indices.contiguous(), int(n_rays))
    else:
        counts = torch.bincount(ray_indices, minlength=n_rays).to(ray_indices.dtype)
        starts = torch.cumsum(counts, 0) - counts
    return torch.stack([starts, counts], dim=-1)
