"""Ray / box intersection and occupancy-grid marching — the two public functions of the reference's
nerfacc/grid.py (`ray_aabb_intersect` :13-52, `traverse_grids` :94-190) over the HIP kernels
(cnc_amd/csrc/march.hip).  The multi-level lookup helper of the reference (`_query`, :204-240) is not part
of the CNC path and is not provided."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import cuda as _C
from .data_specs import RayIntervals, RaySamples

_INF = float("inf")


def _need(t: Tensor, last: int, what: str) -> Tensor:
    if t.dim() != 2 or t.shape[-1] != last:
        raise AssertionError(f"{what} must have shape (n, {last}), got {tuple(t.shape)}")
    return t.contiguous()


@torch.no_grad()
def ray_aabb_intersect(rays_o: Tensor, rays_d: Tensor, aabbs: Tensor, near_plane: float = -_INF,
                       far_plane: float = _INF, miss_value: float = _INF) -> Tuple[Tensor, Tensor, Tensor]:
    """Entry / exit distance of every ray (n_rays, 3) with every box (m, 6) -> t_mins, t_maxs (n_rays, m),
    clipped to [near_plane, far_plane], `miss_value` where the ray misses; hits (n_rays, m) bool."""
    t_lo, t_hi, hit = _C.ray_aabb_intersect(_need(rays_o, 3, "rays_o"), _need(rays_d, 3, "rays_d"),
                                            _need(aabbs, 6, "aabbs"), near_plane, far_plane, miss_value)
    return t_lo, t_hi, hit


def _ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane=-_INF, far_plane=_INF, miss_value=_INF):
    """The same slab test in plain torch, any device (the reference keeps one too, grid.py:55-91)."""
    o, d = rays_o[:, None, :], rays_d[:, None, :]
    through_lo = (aabbs[None, :, :3] - o) / d
    through_hi = (aabbs[None, :, 3:] - o) / d
    t_in = torch.minimum(through_lo, through_hi).amax(dim=-1)
    t_out = torch.maximum(through_lo, through_hi).amin(dim=-1)
    hit = (t_out > t_in) & (t_out > 0)
    clip = lambda t: torch.where(hit, t.clamp(near_plane, far_plane), miss_value)
    return clip(t_in), clip(t_out), hit


@torch.no_grad()
def traverse_grids(rays_o: Tensor, rays_d: Tensor, binaries: Tensor, aabbs: Tensor,
                   near_planes: Optional[Tensor] = None, far_planes: Optional[Tensor] = None,
                   step_size: Optional[float] = 1e-3, cone_angle: Optional[float] = 0.0,
                   traverse_steps_limit: Optional[int] = None, over_allocate: Optional[bool] = False,
                   rays_mask: Optional[Tensor] = None, t_sorted: Optional[Tensor] = None,
                   t_indices: Optional[Tensor] = None,
                   hits: Optional[Tensor] = None) -> Tuple[RayIntervals, RaySamples, Tensor]:
    """March rays through m nested occupancy grids (`binaries` (m, rx, ry, rz) over boxes `aabbs` (m, 6)).

    Returns the interval edges, the sample mid points and the distance at which each ray stopped.  Defaults:
    near 0, far inf, every ray active, no cap on samples per ray (`traverse_steps_limit`).  With
    `over_allocate` (needs a cap) the march is one pass into cap-sized slots per ray instead of count + fill.
    `t_sorted` / `t_indices` / `hits` (the sorted box crossings of every ray) are computed when not given."""
    n = rays_o.shape[0]
    dev = rays_o.device
    near = rays_o.new_zeros(n) if near_planes is None else near_planes
    far = rays_o.new_full((n,), _INF) if far_planes is None else far_planes
    active = torch.ones(n, dtype=torch.bool, device=dev) if rays_mask is None else rays_mask
    cap = -1 if traverse_steps_limit is None else traverse_steps_limit
    if over_allocate and cap <= 0:
        raise AssertionError("traverse_steps_limit must be set if over_allocate is True.")
    if t_sorted is None or t_indices is None or hits is None:
        t_lo, t_hi, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
        t_sorted, t_indices = torch.sort(torch.cat([t_lo, t_hi], dim=-1), dim=-1)
    iv, sm, stopped = _C.traverse_grids(
        rays_o.contiguous(), rays_d.contiguous(), active.contiguous(), binaries.contiguous(), aabbs.contiguous(),
        t_sorted.contiguous(), t_indices.contiguous(), hits.contiguous(), near.contiguous(), far.contiguous(),
        step_size, cone_angle, True, True, True, cap, over_allocate)
    return RayIntervals._from_cpp(iv), RaySamples._from_cpp(sm), stopped
