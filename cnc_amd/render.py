"""Image rendering with an occupancy-grid sampler: the two functions the CNC drivers call,
`render_image_with_occgrid` (training batches, chunked evaluation) and `render_image_with_occgrid_test`
(whole-image evaluation that marches all rays a few steps at a time), plus `Rays`, `namedtuple_map`,
`set_random_seed` and the scene lists — the names of the reference's examples/utils.py (:77-80, :83-216,
:316-489) and examples/datasets/utils.py, with the same arguments and return tuples.

What differs is underneath: sample positions, visibility filtering, compositing and the in-place
accumulation of the iterative render are single HIP kernels (cnc_amd/csrc/march.hip, volrend.hip), and the
(start, count) table of the samples is handed from the sampler to the compositor instead of being rebuilt
from `ray_indices`.
"""
from __future__ import annotations

import collections
import random
from typing import Optional

import numpy as np
import torch

from .nerfacc import OccGridEstimator
from .nerfacc.volrend import rendering

Rays = collections.namedtuple("Rays", ("origins", "viewdirs"))

NERF_SYNTHETIC_SCENES = ["chair", "drums", "ficus", "hotdog", "lego", "materials", "mic", "ship"]
TANKS_SCENES = ["Barn", "Caterpillar", "Family", "Ignatius", "Truck"]


def namedtuple_map(fn, tup):
    """fn over the fields of a namedtuple, keeping None fields."""
    return type(tup)(*(None if x is None else fn(x) for x in tup))


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def _as_ray_list(rays):
    """Rays of an image (H, W, 3) or a batch (N, 3) -> flat (N, 3) origins / directions + the leading shape."""
    lead = tuple(rays.origins.shape[:-1])
    if len(lead) == 2:
        rays = namedtuple_map(lambda r: r.reshape(lead[0] * lead[1], *r.shape[2:]), rays)
    return rays, lead


def _mid_points(origins, dirs, ray_indices, t_starts, t_ends):
    """o + d (t_start + t_end) / 2 and d for every sample: one kernel on the GPU."""
    if origins.is_cuda and ray_indices.dtype == torch.int64:
        from .backends import nerfacc_cuda as _C
        return _C.sample_positions(origins.contiguous(), dirs.contiguous(), ray_indices.contiguous(),
                                   t_starts.contiguous(), t_ends.contiguous(), want_dirs=True)
    d = dirs[ray_indices]
    return origins[ray_indices] + d * (t_starts + t_ends)[:, None] / 2.0, d


class _FieldOnRays:
    """The two callbacks the sampler / compositor want, bound to one set of rays."""

    def __init__(self, field, origins, dirs, with_positions):
        self.field, self.o, self.d, self.with_positions = field, origins, dirs, with_positions

    def density(self, t_starts, t_ends, ray_indices):
        x, _ = _mid_points(self.o, self.d, ray_indices, t_starts, t_ends)
        return self.field.query_density(x).squeeze(-1)

    def density_windows(self):
        """(extension, for OccGridEstimator's front-to-back sampler) What it takes to evaluate depth windows whose sample
        count stays on the device: (origins, directions, evaluate(positions [capacity, 3], n_rows_dev) -> density
        [capacity]) — or None when the field has no evaluator that accepts a device-side row count (then the sampler calls
        `density` on exactly sized batches, one host round trip per window)."""
        probe = getattr(self.field, "_fused_forward", None)
        if probe is None or not self.o.is_cuda or self.o.dtype != torch.float32:
            return None
        fused = probe(self.o)
        if fused is None:
            return None
        return self.o.contiguous(), self.d.contiguous(), lambda x, n: fused(x, n_rows_dev=n).view(-1)

    def colour_and_density(self, t_starts, t_ends, ray_indices):
        x, v = _mid_points(self.o, self.d, ray_indices, t_starts, t_ends)
        return self.colour_and_density_at(x, v)

    def colour_and_density_at(self, x, v):
        """The same for sample positions / directions the caller already has (cnc_march_samples' extras)."""
        rgb, sigma = self.field(x, v)
        sigma = sigma.squeeze(-1)
        return (rgb, sigma, x) if self.with_positions else (rgb, sigma)


def render_image_with_occgrid(radiance_field: torch.nn.Module, estimator: OccGridEstimator, rays: Rays,
                              near_plane: float = 0.0, far_plane: float = 1e10, render_step_size: float = 1e-3,
                              render_bkgd: Optional[torch.Tensor] = None, cone_angle: float = 0.0,
                              alpha_thre: float = 0.0, test_chunk_size: int = 8192,
                              timestamps: Optional[torch.Tensor] = None, return_extra=False, tmp=None):
    """(rgb, opacity, depth, n_samples[, extras]) for a batch or an image of rays.  Training renders every
    ray in one go with jittered sample offsets; evaluation goes `test_chunk_size` rays at a time.  For each
    chunk the estimator picks the samples (a gradient-free density pass decides visibility), then the field is
    queried with gradients and composited."""
    if timestamps is not None:
        raise NotImplementedError("time-dependent fields (dnerf) are outside the CNC path")
    rays, lead = _as_ray_list(rays)
    total = rays.origins.shape[0]
    training = radiance_field.training
    per_chunk = total if training else test_chunk_size
    parts, n_samples, extras = [], 0, None
    for first in range(0, total, max(per_chunk, 1)):
        o, d = rays.origins[first:first + per_chunk], rays.viewdirs[first:first + per_chunk]
        fn = _FieldOnRays(radiance_field, o, d, with_positions=True)
        ray_indices, t_starts, t_ends = estimator.sampling(
            o, d, sigma_fn=fn.density, near_plane=near_plane, far_plane=far_plane,
            render_step_size=render_step_size, stratified=training, cone_angle=cone_angle, alpha_thre=alpha_thre)
        rgb, opacity, depth, extras = rendering(t_starts, t_ends, ray_indices, n_rays=o.shape[0],
                                                rgb_sigma_fn=fn.colour_and_density, render_bkgd=render_bkgd,
                                                packed_info=getattr(estimator, "last_packed_info", None))
        parts.append((rgb, opacity, depth))
        n_samples += t_starts.shape[0]
    rgb, opacity, depth = (torch.cat(p, dim=0).view(*lead, -1) for p in zip(*parts))
    return (rgb, opacity, depth, n_samples, extras) if return_extra else (rgb, opacity, depth, n_samples)


@torch.no_grad()
def render_image_with_occgrid_test(max_samples: int, radiance_field: torch.nn.Module,
                                   estimator: OccGridEstimator, rays: Rays, near_plane: float = 0.0,
                                   far_plane: float = 1e10, render_step_size: float = 1e-3,
                                   render_bkgd: Optional[torch.Tensor] = None, cone_angle: float = 0.0,
                                   alpha_thre: float = 0.0, early_stop_eps: float = 1e-4,
                                   timestamps: Optional[torch.Tensor] = None):
    """Evaluation render of all rays together, a bounded number of march steps per round: rays that became
    opaque (opacity > 1 - early_stop_eps) or left the grid drop out, the survivors continue from where they
    stopped, and the fewer are alive the more steps each gets (num_rays // alive, at most 64).  Returns
    (rgb, opacity, depth, n_samples)."""
    if timestamps is not None:
        raise NotImplementedError("time-dependent fields (dnerf) are outside the CNC path")
    from .backends import nerfacc_cuda as _C
    from .backends import volrend_backend as _K
    rays, lead = _as_ray_list(rays)
    o, d = rays.origins.contiguous(), rays.viewdirs.contiguous()
    n, dev = o.shape[0], o.device
    fn = _FieldOnRays(radiance_field, o, d, with_positions=False)
    rgb = torch.zeros(n, 3, device=dev)
    opacity = torch.zeros(n, 1, device=dev)
    depth = torch.zeros(n, 1, device=dev)

    # where each ray enters / leaves the grid boxes, sorted along the ray (one box: already in order)
    boxes = estimator.aabbs.contiguous()
    t_in, t_out, hit = _C.ray_aabb_intersect(o, d, boxes, -float("inf"), float("inf"), float("inf"))
    crossings = torch.cat([t_in, t_out], dim=-1)
    if boxes.shape[0] > 1:
        crossings, order = torch.sort(crossings, dim=-1)
    else:
        order = torch.arange(2, device=dev, dtype=torch.int64).expand(n, 2)
    crossings, order, hit = crossings.contiguous(), order.contiguous(), hit.contiguous()
    grids = estimator.binaries.contiguous()

    alive = torch.ones(n, dtype=torch.bool, device=dev)
    resume_at = torch.full((n,), float(near_plane), device=dev)
    far = torch.full((n,), float(far_plane), device=dev)
    fewest = 1 if cone_angle == 0 else 4
    opaque = 1.0 - early_stop_eps
    marched = shaded = 0
    while marched < max_samples:
        n_alive = int(alive.sum().item())
        if n_alive == 0:
            break
        steps = max(min(n // n_alive, 64), fewest)
        marched += steps
        # the fill pass emits each sample's position and direction itself (the marching lane has o and d in registers:
        # bit-equal to `_mid_points`, without the pass that re-reads (ray, t0, t1) per sample); the int64 ray ids are
        # only written when something below indexes by them
        ex = {"positions": True, "dirs": True, "ray_indices": "int64" if alpha_thre > 0 else None}
        ray_indices, t_starts, t_ends, starts, counts, resume_at = _C.march_samples(
            o, d, alive, grids, boxes, crossings, order, hit, resume_at, far, render_step_size, cone_angle,
            traverse_steps_limit=steps, want_terminate_planes=True, extras=ex)
        if t_starts.shape[0]:
            rgbs, sigmas = fn.colour_and_density_at(ex["positions"], ex["dirs"])
            rgbs, sigmas = rgbs.float().contiguous(), sigmas.float().contiguous()
            if alpha_thre > 0:
                # transparent samples are left out of the sums (but still attenuate what lies behind them)
                w, _, a, _, _, _ = _K.volrend_forward(starts, counts, t_starts, t_ends, sigmas,
                                                      opacity_in=opacity.view(-1), want_rays=False)
                w = torch.where(a >= alpha_thre, w, torch.zeros_like(w))[:, None]
                shaded += int((a >= alpha_thre).sum().item())
                rgb.index_add_(0, ray_indices, w * rgbs)
                opacity.index_add_(0, ray_indices, w)
                depth.index_add_(0, ray_indices, w * ((t_starts + t_ends)[:, None] / 2.0))
            else:
                # transmittance continues from what the earlier rounds left: prefix = 1 - opacity so far
                _K.volrend_forward(starts, counts, t_starts, t_ends, sigmas, rgbs, opacity_in=opacity.view(-1),
                                   want_samples=False, accumulate_into=(rgb, opacity, depth))
                shaded += t_starts.shape[0]
        # a ray goes on if it is not opaque yet and used its whole step budget (else it left the grid)
        alive = (opacity.view(-1) <= opaque) & (counts == steps)
    if render_bkgd is not None:
        rgb = rgb + render_bkgd * (1.0 - opacity)
    depth = depth / opacity.clamp_min(torch.finfo(rgb.dtype).eps)
    return rgb.view(*lead, -1), opacity.view(*lead, -1), depth.view(*lead, -1), shaded
