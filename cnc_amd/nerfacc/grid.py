"""Ray/AABB intersection and occupancy-grid traversal (reference: nerfacc/grid.py:13-240)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import cuda as _C
from .data_specs import RayIntervals, RaySamples


@torch.no_grad()
def ray_aabb_intersect(rays_o: Tensor, rays_d: Tensor, aabbs: Tensor,
                       near_plane: float = -float("inf"), far_plane: float = float("inf"),
                       miss_value: float = float("inf")) -> Tuple[Tensor, Tensor, Tensor]:
    """(n_rays,3) x (m,6) -> t_mins (n_rays,m), t_maxs (n_rays,m), hits (n_rays,m) bool.
    Misses carry `miss_value`; hits are clipped to [near_plane, far_plane]."""
    assert rays_o.ndim == 2 and rays_o.shape[-1] == 3
    assert rays_d.ndim == 2 and rays_d.shape[-1] == 3
    assert aabbs.ndim == 2 and aabbs.shape[-1] == 6
    t_mins, t_maxs, hits = _C.ray_aabb_intersect(rays_o.contiguous(), rays_d.contiguous(),
                                                 aabbs.contiguous(), near_plane, far_plane,
                                                 miss_value)
    return t_mins, t_maxs, hits


def _ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane=-float("inf"), far_plane=float("inf"),
                        miss_value=float("inf")):
    """Pure-torch slab test, runs anywhere (reference twin: nerfacc/grid.py:55-91)."""
    lo, hi = aabbs[None, :, :3], aabbs[None, :, 3:]
    ta = (lo - rays_o[:, None, :]) / rays_d[:, None, :]
    tb = (hi - rays_o[:, None, :]) / rays_d[:, None, :]
    t_mins = torch.minimum(ta, tb).amax(dim=-1)
    t_maxs = torch.maximum(ta, tb).amin(dim=-1)
    hits = (t_maxs > t_mins) & (t_maxs > 0)
    t_mins = torch.where(hits, t_mins.clamp(near_plane, far_plane), miss_value)
    t_maxs = torch.where(hits, t_maxs.clamp(near_plane, far_plane), miss_value)
    return t_mins, t_maxs, hits


@torch.no_grad()
def traverse_grids(rays_o: Tensor, rays_d: Tensor, binaries: Tensor, aabbs: Tensor,
                   near_planes: Optional[Tensor] = None, far_planes: Optional[Tensor] = None,
                   step_size: Optional[float] = 1e-3, cone_angle: Optional[float] = 0.0,
                   traverse_steps_limit: Optional[int] = None,
                   over_allocate: Optional[bool] = False, rays_mask: Optional[Tensor] = None,
                   t_sorted: Optional[Tensor] = None, t_indices: Optional[Tensor] = None,
                   hits: Optional[Tensor] = None) -> Tuple[RayIntervals, RaySamples, Tensor]:
    """March rays through m nested occupancy grids `binaries` (m,rx,ry,rz) with boxes `aabbs`
    (m,6).  Returns the interval edges, the sample mid-points and the per-ray termination t.
    `traverse_steps_limit` bounds the samples per ray; with `over_allocate` the march is a single
    pass into an upper-bound allocation (used by the iterative evaluation loop)."""
    if near_planes is None:
        near_planes = torch.zeros_like(rays_o[:, 0])
    if far_planes is None:
        far_planes = torch.full_like(rays_o[:, 0], float("inf"))
    if rays_mask is None:
        rays_mask = torch.ones_like(rays_o[:, 0], dtype=torch.bool)
    if traverse_steps_limit is None:
        traverse_steps_limit = -1
    if over_allocate:
        assert traverse_steps_limit > 0, "traverse_steps_limit must be set if over_allocate is True."
    if t_sorted is None or t_indices is None or hits is None:
        t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
        t_sorted, t_indices = torch.sort(torch.cat([t_mins, t_maxs], dim=-1), dim=-1)
    intervals, samples, termination_planes = _C.traverse_grids(
        rays_o.contiguous(), rays_d.contiguous(), rays_mask.contiguous(), binaries.contiguous(),
        aabbs.contiguous(), t_sorted.contiguous(), t_indices.contiguous(), hits.contiguous(),
        near_planes.contiguous(), far_planes.contiguous(), step_size, cone_angle,
        True, True, True, traverse_steps_limit, over_allocate)
    return RayIntervals._from_cpp(intervals), RaySamples._from_cpp(samples), termination_planes


def _enlarge_aabb(aabb, factor: float) -> Tensor:
    center = (aabb[:3] + aabb[3:]) / 2
    extent = (aabb[3:] - aabb[:3]) / 2
    return torch.cat([center - extent * factor, center + extent * factor])


def _query(x: Tensor, data: Tensor, base_aabb: Tensor):
    """Look up (m,rx,ry,rz) mip-grid values at points x, assuming each level doubles the box."""
    aabb_min, aabb_max = torch.split(base_aabb, 3, dim=0)
    x_norm = (x - aabb_min) / (aabb_max - aabb_min)
    maxval = (x_norm - 0.5).abs().max(dim=-1).values.clamp(min=0.1)
    mip = (torch.frexp(maxval)[1].long() + 1).clamp(min=0)
    selector = mip < data.shape[0]
    x_unit = (x_norm - 0.5) / (2 ** mip)[:, None] + 0.5
    resolution = torch.tensor(data.shape[1:], device=x.device)
    ix = torch.clamp((x_unit * resolution).long(), max=resolution - 1)
    mip = mip.clamp(max=data.shape[0] - 1)
    return data[mip, ix[:, 0], ix[:, 1], ix[:, 2]] * selector, selector
