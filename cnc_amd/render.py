"""Render harness: occupancy-grid sampling + volume rendering of a radiance field — host mirror of
examples/utils.py (`set_random_seed` :77-80, `render_image_with_occgrid` :83-216,
`render_image_with_occgrid_test` :316-489) and examples/datasets/utils.py (`Rays`, `namedtuple_map`).
Same signatures and return tuples; every kernel underneath is HIP (cnc_amd.nerfacc, GridEncoder)."""
from __future__ import annotations

import collections
import random
from typing import Optional

import numpy as np
import torch

from .nerfacc import OccGridEstimator
from .nerfacc.grid import ray_aabb_intersect, traverse_grids
from .nerfacc.volrend import accumulate_along_rays_, render_weight_from_density, rendering


def _sample_positions(rays_o, rays_d, ray_indices, t_starts, t_ends):
    """(positions, directions) of the samples; the fused kernel on the GPU, the reference's
    expression elsewhere (rays are never differentiated)."""
    if rays_o.is_cuda and ray_indices.dtype == torch.int64:
        from .backends import nerfacc_cuda as _C
        return _C.sample_positions(rays_o.contiguous(), rays_d.contiguous(), ray_indices.contiguous(),
                                   t_starts.contiguous(), t_ends.contiguous(), want_dirs=True)
    o, d = rays_o[ray_indices], rays_d[ray_indices]
    return o + d * (t_starts + t_ends)[:, None] / 2.0, d

Rays = collections.namedtuple("Rays", ("origins", "viewdirs"))

NERF_SYNTHETIC_SCENES = ["chair", "drums", "ficus", "hotdog", "lego", "materials", "mic", "ship"]
TANKS_SCENES = ["Barn", "Caterpillar", "Family", "Ignatius", "Truck"]


def namedtuple_map(fn, tup):
    return type(tup)(*(None if x is None else fn(x) for x in tup))


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def _flatten(rays):
    shape = rays.origins.shape
    if len(shape) == 3:
        n = shape[0] * shape[1]
        rays = namedtuple_map(lambda r: r.reshape([n] + list(r.shape[2:])), rays)
    else:
        n = shape[0]
    return rays, shape, n


def render_image_with_occgrid(radiance_field: torch.nn.Module, estimator: OccGridEstimator, rays: Rays,
                              near_plane: float = 0.0, far_plane: float = 1e10,
                              render_step_size: float = 1e-3, render_bkgd: Optional[torch.Tensor] = None,
                              cone_angle: float = 0.0, alpha_thre: float = 0.0,
                              test_chunk_size: int = 8192, timestamps: Optional[torch.Tensor] = None,
                              return_extra=False, tmp=None):
    """Training / chunked-eval render: sample with the estimator (density pre-pass for visibility),
    then query the field with gradients and composite.  Returns (rgb, opacity, depth, n_samples[, extras])."""
    rays, rays_shape, num_rays = _flatten(rays)

    def positions_of(t_starts, t_ends, ray_indices):
        # o + d * (t_starts + t_ends) / 2 and d, one kernel (examples/utils.py:251-262)
        return _sample_positions(chunk_rays.origins, chunk_rays.viewdirs, ray_indices, t_starts, t_ends)

    def sigma_fn(t_starts, t_ends, ray_indices):
        positions, _ = positions_of(t_starts, t_ends, ray_indices)
        return radiance_field.query_density(positions).squeeze(-1)

    def rgb_sigma_fn(t_starts, t_ends, ray_indices):
        positions, dirs = positions_of(t_starts, t_ends, ray_indices)
        rgbs, sigmas = radiance_field(positions, dirs)
        return rgbs, sigmas.squeeze(-1), positions

    chunk = torch.iinfo(torch.int32).max if radiance_field.training else test_chunk_size
    results = []
    extras = None
    for i in range(0, num_rays, chunk):
        chunk_rays = namedtuple_map(lambda r: r[i:i + chunk], rays)
        ray_indices, t_starts, t_ends = estimator.sampling(
            chunk_rays.origins, chunk_rays.viewdirs, sigma_fn=sigma_fn, near_plane=near_plane,
            far_plane=far_plane, render_step_size=render_step_size, stratified=radiance_field.training,
            cone_angle=cone_angle, alpha_thre=alpha_thre)
        rgb, opacity, depth, extras = rendering(t_starts, t_ends, ray_indices,
                                                n_rays=chunk_rays.origins.shape[0],
                                                rgb_sigma_fn=rgb_sigma_fn, render_bkgd=render_bkgd)
        results.append((rgb, opacity, depth, len(t_starts)))
    colors = torch.cat([r[0] for r in results], dim=0)
    opacities = torch.cat([r[1] for r in results], dim=0)
    depths = torch.cat([r[2] for r in results], dim=0)
    n_samples = sum(r[3] for r in results)
    out = (colors.view((*rays_shape[:-1], -1)), opacities.view((*rays_shape[:-1], -1)),
           depths.view((*rays_shape[:-1], -1)), n_samples)
    return out + (extras,) if return_extra else out


@torch.no_grad()
def render_image_with_occgrid_test(max_samples: int, radiance_field: torch.nn.Module,
                                   estimator: OccGridEstimator, rays: Rays, near_plane: float = 0.0,
                                   far_plane: float = 1e10, render_step_size: float = 1e-3,
                                   render_bkgd: Optional[torch.Tensor] = None, cone_angle: float = 0.0,
                                   alpha_thre: float = 0.0, early_stop_eps: float = 1e-4,
                                   timestamps: Optional[torch.Tensor] = None):
    """Whole-image evaluation render: march all rays a bounded number of steps at a time (more steps
    per round as rays die), accumulate in place, restart the survivors from their termination
    planes (examples/utils.py:395-478)."""
    rays, rays_shape, num_rays = _flatten(rays)
    rays_o, rays_d = rays.origins, rays.viewdirs
    device = rays_o.device

    def rgb_sigma_fn(t_starts, t_ends, ray_indices):
        positions, d = _sample_positions(rays_o, rays_d, ray_indices, t_starts, t_ends)
        rgbs, sigmas = radiance_field(positions, d)
        return rgbs, sigmas.squeeze(-1)

    opacity = torch.zeros(num_rays, 1, device=device)
    depth = torch.zeros(num_rays, 1, device=device)
    rgb = torch.zeros(num_rays, 3, device=device)
    ray_mask = torch.ones(num_rays, device=device).bool()
    min_samples = 1 if cone_angle == 0 else 4
    iter_samples = total_samples = 0
    near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
    far_planes = torch.full_like(rays_o[..., 0], fill_value=far_plane)

    t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, estimator.aabbs)
    n_grids = estimator.binaries.size(0)
    if n_grids > 1:
        t_sorted, t_indices = torch.sort(torch.cat([t_mins, t_maxs], -1), -1)
    else:
        t_sorted = torch.cat([t_mins, t_maxs], -1)
        t_indices = torch.arange(0, n_grids * 2, device=device, dtype=torch.int64).expand(num_rays, n_grids * 2)
    opc_thre = 1 - early_stop_eps
    rgbs = rgb

    while iter_samples < max_samples:
        n_alive = ray_mask.sum().item()
        if n_alive == 0:
            break
        n_samples = max(min(num_rays // n_alive, 64), min_samples)
        iter_samples += n_samples
        intervals, samples, termination_planes = traverse_grids(
            rays_o, rays_d, estimator.binaries, estimator.aabbs, near_planes, far_planes,
            render_step_size, cone_angle, n_samples, True, ray_mask, t_sorted, t_indices, hits)
        t_starts = intervals.vals[intervals.is_left]
        t_ends = intervals.vals[intervals.is_right]
        ray_indices = samples.ray_indices[samples.is_valid]
        packed_info = samples.packed_info
        rgbs, sigmas = rgb_sigma_fn(t_starts, t_ends, ray_indices)
        weights, _, alphas = render_weight_from_density(
            t_starts, t_ends, sigmas, ray_indices=ray_indices, n_rays=num_rays,
            prefix_trans=1 - opacity[ray_indices].squeeze(-1))
        if alpha_thre > 0:
            vis = alphas >= alpha_thre
            ray_indices, rgbs, weights, t_starts, t_ends = (ray_indices[vis], rgbs[vis], weights[vis],
                                                            t_starts[vis], t_ends[vis])
        accumulate_along_rays_(weights, values=rgbs, ray_indices=ray_indices, outputs=rgb)
        accumulate_along_rays_(weights, values=None, ray_indices=ray_indices, outputs=opacity)
        accumulate_along_rays_(weights, values=(t_starts + t_ends)[..., None] / 2.0,
                               ray_indices=ray_indices, outputs=depth)
        near_planes = termination_planes
        ray_mask = torch.logical_and(opacity.view(-1) <= opc_thre, packed_info[:, 1] == n_samples)
        total_samples += ray_indices.shape[0]

    rgb = rgb + render_bkgd * (1.0 - opacity)
    depth = depth / opacity.clamp_min(torch.finfo(rgbs.dtype).eps)
    return (rgb.view((*rays_shape[:-1], -1)), opacity.view((*rays_shape[:-1], -1)),
            depth.view((*rays_shape[:-1], -1)), total_samples)
