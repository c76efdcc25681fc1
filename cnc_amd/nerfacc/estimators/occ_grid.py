"""Occupancy-grid transmittance estimator (reference: nerfacc/estimators/occ_grid.py:14-443).

Same constructor, buffers (`resolution`, `aabbs`, `occs`, `binaries`, `grid_coords`,
`grid_indices`) and methods (`sampling`, `update_every_n_steps`, `_update`,
`mark_invisible_cells`) as the reference, so the CNC drivers and context models can use it
unchanged.  The march itself is the HIP `traverse_grids` kernel.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple, Union

import torch
from torch import Tensor

from ..grid import _enlarge_aabb, traverse_grids
from ..volrend import render_visibility_from_alpha, render_visibility_from_density
from .base import AbstractEstimator


class OccGridEstimator(AbstractEstimator):
    DIM: int = 3

    def __init__(self, roi_aabb: Union[List[int], Tensor],
                 resolution: Union[int, List[int], Tensor] = 128, levels: int = 1, **kwargs) -> None:
        super().__init__()
        if "contraction_type" in kwargs:
            raise ValueError("`contraction_type` is not supported anymore for nerfacc >= 0.4.0.")
        if isinstance(resolution, int):
            resolution = [resolution] * self.DIM
        if isinstance(resolution, (list, tuple)):
            resolution = torch.tensor(resolution, dtype=torch.int32)
        assert isinstance(resolution, Tensor), f"Invalid type: {resolution}!"
        assert resolution.shape[0] == self.DIM, f"Invalid shape: {resolution}!"
        if isinstance(roi_aabb, (list, tuple)):
            roi_aabb = torch.tensor(roi_aabb, dtype=torch.float32)
        assert isinstance(roi_aabb, Tensor), f"Invalid type: {roi_aabb}!"
        assert roi_aabb.shape[0] == self.DIM * 2, f"Invalid shape: {roi_aabb}!"

        # level i covers the roi scaled by 2^i about its centre
        aabbs = torch.stack([_enlarge_aabb(roi_aabb, 2 ** i) for i in range(levels)], dim=0)
        self.cells_per_lvl = int(resolution.prod().item())
        self.levels = levels
        self.register_buffer("resolution", resolution)
        self.register_buffer("aabbs", aabbs)
        self.register_buffer("occs", torch.zeros(self.levels * self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros([levels] + resolution.tolist(), dtype=torch.bool))
        grid_coords = _meshgrid3d(resolution).reshape(self.cells_per_lvl, self.DIM)
        self.register_buffer("grid_coords", grid_coords, persistent=False)
        self.register_buffer("grid_indices", torch.arange(self.cells_per_lvl), persistent=False)

    @torch.no_grad()
    def sampling(self, rays_o: Tensor, rays_d: Tensor, sigma_fn: Optional[Callable] = None,
                 alpha_fn: Optional[Callable] = None, near_plane: float = 0.0,
                 far_plane: float = 1e10, t_min: Optional[Tensor] = None,
                 t_max: Optional[Tensor] = None, render_step_size: float = 1e-3,
                 early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, stratified: bool = False,
                 cone_angle: float = 0.0) -> Tuple[Tensor, Tensor, Tensor]:
        """Returns (ray_indices, t_starts, t_ends) of the samples that survive occupancy skipping
        and, if `sigma_fn` / `alpha_fn` is given, the transmittance / alpha visibility test."""
        near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
        far_planes = torch.full_like(rays_o[..., 0], fill_value=far_plane)
        if t_min is not None:
            near_planes = torch.clamp(near_planes, min=t_min)
        if t_max is not None:
            far_planes = torch.clamp(far_planes, max=t_max)
        if stratified:
            near_planes += torch.rand_like(near_planes) * render_step_size
        intervals, samples, _ = traverse_grids(rays_o, rays_d, self.binaries, self.aabbs,
                                               near_planes=near_planes, far_planes=far_planes,
                                               step_size=render_step_size, cone_angle=cone_angle)
        t_starts = intervals.vals[intervals.is_left]
        t_ends = intervals.vals[intervals.is_right]
        ray_indices = samples.ray_indices
        packed_info = samples.packed_info

        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and (sigma_fn is not None or alpha_fn is not None):
            alpha_thre = min(alpha_thre, self.occs.mean().item())
            if sigma_fn is not None:
                sigmas = (sigma_fn(t_starts, t_ends, ray_indices) if t_starts.shape[0] != 0
                          else torch.empty((0,), device=t_starts.device))
                assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(sigmas.shape)
                masks = render_visibility_from_density(t_starts=t_starts, t_ends=t_ends, sigmas=sigmas,
                                                       packed_info=packed_info,
                                                       early_stop_eps=early_stop_eps,
                                                       alpha_thre=alpha_thre)
            else:
                alphas = (alpha_fn(t_starts, t_ends, ray_indices) if t_starts.shape[0] != 0
                          else torch.empty((0,), device=t_starts.device))
                assert alphas.shape == t_starts.shape, "alphas must have shape of (N,)! Got {}".format(alphas.shape)
                masks = render_visibility_from_alpha(alphas=alphas, packed_info=packed_info,
                                                     early_stop_eps=early_stop_eps,
                                                     alpha_thre=alpha_thre)
            ray_indices, t_starts, t_ends = ray_indices[masks], t_starts[masks], t_ends[masks]
        return ray_indices, t_starts, t_ends

    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2,
                             ema_decay: float = 0.95, warmup_steps: int = 256, n: int = 16) -> None:
        if not self.training:
            raise RuntimeError("You should only call this function only during training. "
                               "Please call _update() directly if you want to update the "
                               "field during inference.")
        if step % n == 0 and self.training:
            self._update(step=step, occ_eval_fn=occ_eval_fn, occ_thre=occ_thre, ema_decay=ema_decay,
                         warmup_steps=warmup_steps)

    @torch.no_grad()
    def mark_invisible_cells(self, K: Tensor, c2w: Tensor, width: int, height: int,
                             near_plane: float = 0.0, chunk: int = 32 ** 3) -> None:
        """Set occs = -1 for cells no camera sees (or that sit in front of a camera's near plane)."""
        assert K.dim() == 3 and K.shape[1:] == (3, 3)
        assert c2w.dim() == 3 and (c2w.shape[1:] == (3, 4) or c2w.shape[1:] == (4, 4))
        assert K.shape[0] == c2w.shape[0] or K.shape[0] == 1
        n_cams = c2w.shape[0]
        w2c_R = c2w[:, :3, :3].transpose(2, 1)
        w2c_T = -w2c_R @ c2w[:, :3, 3:]
        for lvl, indices in enumerate(self._get_all_cells()):
            coords = self.grid_coords[indices]
            lo, hi = self.aabbs[lvl, :3], self.aabbs[lvl, 3:]
            for i in range(0, len(indices), chunk):
                x = coords[i:i + chunk] / (self.resolution - 1)
                idx = indices[i:i + chunk]
                xyz_w = (lo + x * (hi - lo)).T
                uvd = K @ (w2c_R @ xyz_w + w2c_T)
                uv = uvd[:, :2] / uvd[:, 2:]
                in_image = ((uvd[:, 2] >= 0) & (uv[:, 0] >= 0) & (uv[:, 0] < width)
                            & (uv[:, 1] >= 0) & (uv[:, 1] < height))
                seen = ((uvd[:, 2] >= near_plane) & in_image).sum(0) / n_cams
                too_near = ((uvd[:, 2] < near_plane) & in_image).any(0)
                valid = (seen > 0) & (~too_near)
                self.occs[lvl * self.cells_per_lvl + idx] = torch.where(valid, 0.0, -1.0)

    @torch.no_grad()
    def _get_all_cells(self) -> List[Tensor]:
        out = []
        for lvl in range(self.levels):
            cell_ids = lvl * self.cells_per_lvl + self.grid_indices
            out.append(self.grid_indices[self.occs[cell_ids] >= 0.0])
        return out

    @torch.no_grad()
    def _sample_uniform_and_occupied_cells(self, n: int) -> List[Tensor]:
        out = []
        for lvl in range(self.levels):
            uniform = torch.randint(self.cells_per_lvl, (n,), device=self.device)
            uniform = uniform[self.occs[lvl * self.cells_per_lvl + uniform] >= 0.0]
            occupied = torch.nonzero(self.binaries[lvl].flatten())[:, 0]
            if n < len(occupied):
                occupied = occupied[torch.randint(len(occupied), (n,), device=self.device)]
            out.append(torch.cat([uniform, occupied], dim=0))
        return out

    @torch.no_grad()
    def _update(self, step: int, occ_eval_fn: Callable, occ_thre: float = 0.01,
                ema_decay: float = 0.95, warmup_steps: int = 256) -> None:
        """EMA update of `occs` from `occ_eval_fn` at jittered cell positions, then re-threshold."""
        if step < warmup_steps:
            lvl_indices = self._get_all_cells()
        else:
            lvl_indices = self._sample_uniform_and_occupied_cells(self.cells_per_lvl // 4)
        for lvl, indices in enumerate(lvl_indices):
            coords = self.grid_coords[indices]
            x = (coords + torch.rand_like(coords, dtype=torch.float32)) / self.resolution
            x = self.aabbs[lvl, :3] + x * (self.aabbs[lvl, 3:] - self.aabbs[lvl, :3])
            occ = occ_eval_fn(x).squeeze(-1)
            cell_ids = lvl * self.cells_per_lvl + indices
            self.occs[cell_ids] = torch.maximum(self.occs[cell_ids] * ema_decay, occ)
        thre = torch.clamp(self.occs[self.occs >= 0].mean(), max=occ_thre)
        self.binaries = (self.occs > thre).view(self.binaries.shape)


def _meshgrid3d(res: Tensor, device: Union[torch.device, str] = "cpu") -> Tensor:
    assert len(res) == 3
    rx, ry, rz = res.tolist()
    axes = [torch.arange(r, dtype=torch.long) for r in (rx, ry, rz)]
    return torch.stack(torch.meshgrid(axes, indexing="ij"), dim=-1).to(device)
