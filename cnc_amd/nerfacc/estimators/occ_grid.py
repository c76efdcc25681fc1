"""Occupancy-grid estimator: which stretches of a ray are worth sampling.

Interface of the reference's `nerfacc.estimators.occ_grid.OccGridEstimator` (occ_grid.py:14-443) — same
constructor, buffers (`resolution`, `aabbs`, `occs`, `binaries`; non-persistent `grid_coords`,
`grid_indices`), methods and RNG call order (so a seeded `_update` reproduces the reference's grid,
tests/golden/occ_grid.npz) — on top of the HIP marcher and the fused visibility / compaction kernels:

    sampling = march (count + fill of (ray, t_start, t_end))   cnc_march_samples, cnc_amd/csrc/march.hip
             -> sigma_fn / alpha_fn (the field)
             -> transmittance test + survivor counts     k_visibility
             -> stable compaction                         k_compact            (one host sync)

`sampling` also remembers the (start, count) table of what it returned (`last_packed_info`), so the renderer
does not have to rebuild it from `ray_indices`.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple, Union

import os

import torch
from torch import Tensor

from .base import AbstractEstimator


def _box_scaled_about_centre(box: Tensor, factor: float) -> Tensor:
    lo, hi = box[:3], box[3:]
    mid, half = (lo + hi) * 0.5, (hi - lo) * 0.5
    return torch.cat([mid - half * factor, mid + half * factor])


def _cell_lattice(res: Tensor) -> Tensor:
    """Integer (x, y, z) of every cell, x slowest — the flattening order of `binaries`."""
    nx, ny, nz = (int(v) for v in res.tolist())
    xs, ys, zs = torch.arange(nx), torch.arange(ny), torch.arange(nz)
    return torch.stack(torch.meshgrid(xs, ys, zs, indexing="ij"), dim=-1).reshape(-1, 3)


class OccGridEstimator(AbstractEstimator):
    DIM: int = 3

    def __init__(self, roi_aabb: Union[List[int], Tensor], resolution: Union[int, List[int], Tensor] = 128,
                 levels: int = 1, **kwargs) -> None:
        super().__init__()
        if "contraction_type" in kwargs:
            raise ValueError("`contraction_type` is not supported anymore for nerfacc >= 0.4.0.")
        res = resolution
        if isinstance(res, int):
            res = [res] * self.DIM
        if not isinstance(res, Tensor):
            res = torch.tensor(list(res), dtype=torch.int32)
        box = roi_aabb if isinstance(roi_aabb, Tensor) else torch.tensor(list(roi_aabb), dtype=torch.float32)
        if res.shape[0] != self.DIM:
            raise AssertionError(f"Invalid shape: {res}!")
        if box.shape[0] != 2 * self.DIM:
            raise AssertionError(f"Invalid shape: {box}!")

        self.levels = levels
        self.cells_per_lvl = int(res.prod().item())
        # level k covers the region of interest scaled by 2^k about its centre
        self.register_buffer("resolution", res)
        self.register_buffer("aabbs", torch.stack([_box_scaled_about_centre(box, 2.0 ** k) for k in range(levels)]))
        self.register_buffer("occs", torch.zeros(levels * self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros([levels] + res.tolist(), dtype=torch.bool))
        self.register_buffer("grid_coords", _cell_lattice(res), persistent=False)
        self.register_buffer("grid_indices", torch.arange(self.cells_per_lvl), persistent=False)
        self.last_packed_info: Optional[Tensor] = None

    # ----------------------------------------------------------------------------------- sampling
    def _march(self, rays_o, rays_d, near_planes, far_planes, step, cone_angle):
        """(ray_indices, t_starts, t_ends, starts, counts) of every sample inside occupied cells."""
        from ...backends import nerfacc_cuda as _C
        rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
        boxes = self.aabbs.contiguous()
        t_lo, t_hi, hit = _C.ray_aabb_intersect(rays_o, rays_d, boxes, -float("inf"), float("inf"), float("inf"))
        crossings = torch.cat([t_lo, t_hi], dim=-1)
        if boxes.shape[0] > 1:          # nested grids: the box crossings of a ray in order along it
            crossings, order = torch.sort(crossings, dim=-1)
        else:                           # one box: (entry, exit) is already sorted
            order = torch.arange(2, device=rays_o.device, dtype=torch.int64).expand(rays_o.shape[0], 2)
        first = {"at": self._WINDOWS[0]} if self._WINDOWS else None
        ray_indices, t_starts, t_ends, starts, counts, _ = _C.march_samples(
            rays_o, rays_d, None, self.binaries.contiguous(), boxes, crossings.contiguous(), order.contiguous(),
            hit.contiguous(), near_planes.contiguous(), far_planes.contiguous(), step, cone_angle, clamped_total=first)
        self._first_window_total = None if first is None else first.get("total")
        return ray_indices, t_starts, t_ends, starts, counts

    def _march_key(self, rays_o, rays_d, near_plane, far_plane, render_step_size, stratified, cone_angle):
        b = self.binaries
        return (rays_o.data_ptr(), rays_d.data_ptr(), tuple(rays_o.shape), str(rays_o.device), b.data_ptr(), b._version,
                float(near_plane), float(far_plane), float(render_step_size), bool(stratified), float(cone_angle))

    @torch.no_grad()
    def premarch(self, rays_o: Tensor, rays_d: Tensor, near_plane: float = 0.0, far_plane: float = 1e10,
                 render_step_size: float = 1e-3, stratified: bool = False, cone_angle: float = 0.0) -> None:
        """(extension) The march of a LATER `sampling(rays_o, rays_d, ...)` call with the same options, made now, on the
        caller's current stream.  The march reads the rays and the occupancy grid and nothing else — not the field — so a
        training loop can run the next batch's march (two serial traversal passes and a host round trip for the sample count:
        ~1.1 ms at the head of the render pass, which is the step's critical chain) while this step's backward keeps the
        GPU busy.  `sampling` takes the result when it is called with these very tensors and options and the grid has not
        been replaced or written since; otherwise it is dropped and the call marches as usual.  The jitter of a stratified
        call is drawn here, from the same generator, in the same way."""
        n_rays = rays_o.shape[0]
        near = torch.full((n_rays,), float(near_plane), dtype=rays_o.dtype, device=rays_o.device)
        far = torch.full((n_rays,), float(far_plane), dtype=rays_o.dtype, device=rays_o.device)
        if stratified:
            near = near + torch.rand_like(near) * render_step_size
        result = self._march(rays_o, rays_d, near, far, render_step_size, cone_angle)
        event = torch.cuda.current_stream(rays_o.device).record_event() if rays_o.is_cuda else None
        self._premarched = {"key": self._march_key(rays_o, rays_d, near_plane, far_plane, render_step_size, stratified,
                                                   cone_angle),
                            "result": result, "first_total": self._first_window_total, "event": event,
                            "keep": (rays_o, rays_d, self.binaries)}      # the addresses in the key stay unique meanwhile

    @torch.no_grad()
    def sampling(self, rays_o: Tensor, rays_d: Tensor, sigma_fn: Optional[Callable] = None,
                 alpha_fn: Optional[Callable] = None, near_plane: float = 0.0, far_plane: float = 1e10,
                 t_min: Optional[Tensor] = None, t_max: Optional[Tensor] = None, render_step_size: float = 1e-3,
                 early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, stratified: bool = False,
                 cone_angle: float = 0.0, front_to_back: Optional[bool] = None) -> Tuple[Tensor, Tensor, Tensor]:
        """Samples (ray_indices, t_starts, t_ends) along the rays: marched with `render_step_size` (growing
        with `cone_angle`) through occupied cells between the near / far planes (optionally per-ray `t_min` /
        `t_max`), jittered by up to one step when `stratified`; if a density (`sigma_fn`) or opacity
        (`alpha_fn`) callback is given, samples behind transmittance < `early_stop_eps` — and, for
        `alpha_thre` > 0, samples more transparent than min(alpha_thre, mean occupancy) — are dropped.

        `front_to_back` (extension): evaluate `sigma_fn` in depth windows and stop a ray once what is left of it
        is below `early_stop_eps` (see `_density_front_to_back`).  The callback is then called up to
        len(CNC_SAMPLER_WINDOWS) + 1 times on SUBSETS of the samples (never on all of them at once) and samples that
        are never evaluated count as sigma = 0, so it must be a stateless per-sample function returning non-negative
        densities — true of the radiance field's `query_density`; the survivors are then the same as with one call.
        None (default): on when the previous call marched at least 4x what it kept and this one marches >= 2^17
        samples on a GPU; True / False force it."""
        from ...backends import volrend_backend as _K
        n_rays = rays_o.shape[0]
        near = torch.full((n_rays,), float(near_plane), dtype=rays_o.dtype, device=rays_o.device)
        far = torch.full((n_rays,), float(far_plane), dtype=rays_o.dtype, device=rays_o.device)
        if t_min is not None:
            near = torch.maximum(near, t_min)
        if t_max is not None:
            far = torch.minimum(far, t_max)
        pre, self._premarched = getattr(self, "_premarched", None), None
        if pre is not None and t_min is None and t_max is None and pre["key"] == self._march_key(
                rays_o, rays_d, near_plane, far_plane, render_step_size, stratified, cone_angle):
            # the march of exactly this call was made ahead of time (`premarch`): join its stream and take it
            if pre["event"] is not None:
                cur = torch.cuda.current_stream(rays_o.device)
                cur.wait_event(pre["event"])
                for t in pre["result"]:
                    t.record_stream(cur)
            ray_indices, t_starts, t_ends, starts, counts = pre["result"]
            self._first_window_total = pre["first_total"]
        else:
            if stratified:
                near = near + torch.rand_like(near) * render_step_size
            ray_indices, t_starts, t_ends, starts, counts = self._march(rays_o, rays_d, near, far, render_step_size,
                                                                        cone_angle)
        field = sigma_fn if sigma_fn is not None else alpha_fn
        if field is not None and (alpha_thre > 0.0 or early_stop_eps > 0.0):
            n = t_starts.shape[0]
            windows = self._front_to_back_pays(n) if front_to_back is None else (bool(front_to_back) and bool(self._WINDOWS))
            if n and sigma_fn is not None and early_stop_eps > 0.0 and windows:
                owner = getattr(sigma_fn, "__self__", None)
                counted = owner.density_windows() if self._COUNTED_WINDOWS and hasattr(owner, "density_windows") else None
                if counted is not None:
                    values = self._density_front_to_back_counted(counted, starts, counts, t_starts, t_ends, early_stop_eps)
                else:
                    values = self._density_front_to_back(sigma_fn, starts, counts, t_starts, t_ends, early_stop_eps)
            else:
                values = field(t_starts, t_ends, ray_indices) if n else torch.empty((0,), device=t_starts.device)
            if values.shape != t_starts.shape:
                raise AssertionError("{} must have shape of (N,)! Got {}".format(
                    "sigmas" if sigma_fn is not None else "alphas", values.shape))
            # the threshold is capped by the grid's mean occupancy ON THE DEVICE (the reference reads it back)
            cap = self.occs.mean().reshape(1) if alpha_thre > 0.0 else None
            mask, kept = _K.render_visibility(starts, counts, values.float().contiguous(), t_starts, t_ends,
                                              from_alpha=sigma_fn is None, early_stop_eps=early_stop_eps,
                                              alpha_thre=alpha_thre, alpha_thre_cap=cap)
            ray_indices, t_starts, t_ends, starts, counts = _K.compact_samples(starts, counts, mask, kept,
                                                                               t_starts, t_ends)
            self._marched_per_kept = n / max(int(t_starts.shape[0]), 1)
        self.last_packed_info = torch.stack([starts, counts], dim=-1)
        return ray_indices, t_starts, t_ends

    # (extension) Once the field has formed opaque surfaces most marched samples lie BEHIND them: the visibility test
    # drops every sample whose transmittance is below early_stop_eps whatever its density (volrend.py:425-475), yet
    # the one-shot density callback evaluates them all — 1.5-2.0 M evaluations for 2^18 survivors on the full-size
    # training step.  Front to back in depth windows, a ray leaves as soon as what is left of it is below the
    # threshold: the same survivors (the densities of the samples never evaluated do not enter the test), for one
    # device->host read per window.
    _WINDOWS = tuple(int(v) for v in os.environ.get("CNC_SAMPLER_WINDOWS", "16,32").split(",") if v.strip())

    def _front_to_back_pays(self, n_samples: int) -> bool:
        return (bool(self._WINDOWS) and self.binaries.is_cuda and n_samples >= (1 << 17)
                and getattr(self, "_marched_per_kept", 1.0) >= 4.0)

    def _density_front_to_back(self, sigma_fn, starts, counts, t_starts, t_ends, early_stop_eps):
        from ...backends import volrend_backend as _K
        sigmas = torch.zeros_like(t_starts)
        done = torch.zeros_like(counts)
        take = torch.empty_like(counts)
        # a margin below the threshold: the window test's sum and the visibility test's prefix scan associate differently
        threshold = early_stop_eps * (1.0 - 1e-3)
        for i, w in enumerate(self._WINDOWS + (None,)):
            # the window just evaluated joins the ray's prefix, what is left of the ray decides whether it goes on, the
            # next window is sized: one kernel, in place on (done, take)
            _K.ray_window_next(starts, counts, t_starts, t_ends, sigmas, done, take, w, threshold, first=i == 0)
            ends = torch.cumsum(take, 0)
            if i == 0 and self._first_window_total is not None:
                total = int(self._first_window_total)          # came back with the march's own sample total
            else:
                total = int(ends[-1].item())
            if total == 0:
                break
            ri_w, ts_w, te_w, src = _K.window_samples(starts, done, take, t_starts, t_ends, total, ends=ends)
            values = sigma_fn(ts_w, te_w, ri_w)
            if values.shape != ts_w.shape:
                raise AssertionError("sigmas must have shape of (N,)! Got {}".format(values.shape))
            sigmas.index_copy_(0, src, values.to(sigmas.dtype))
        return sigmas

    # (round 6) The same windows with their sample counts left on the device: a window's size is the last element of a
    # running sum the host never reads; positions, density and the scatter back work on buffers of a bound's size and stop
    # at the count (cnc_ray_window_positions, cnc_fused_field_t.n_rows_dev, cnc_scatter_counted).  Same densities for the
    # same samples — what changes is that the host issues the three windows back to back instead of waiting for each
    # (two round trips, ~0.3 ms of the step's critical chain).  Needs a field that takes a device-side row count
    # (`_FieldOnRays.density_windows`); CNC_SAMPLER_COUNTED_WINDOWS=0: off.
    _COUNTED_WINDOWS = os.environ.get("CNC_SAMPLER_COUNTED_WINDOWS", "1") == "1"

    def _density_front_to_back_counted(self, counted, starts, counts, t_starts, t_ends, early_stop_eps):
        from ...backends import volrend_backend as _K
        rays_o, rays_d, evaluate = counted
        n, n_rays = int(t_starts.shape[0]), int(starts.shape[0])
        sigmas = torch.zeros_like(t_starts)
        done = torch.zeros_like(counts)
        take = torch.empty_like(counts)
        threshold = early_stop_eps * (1.0 - 1e-3)
        known = 0                      # samples the host KNOWS to be evaluated already (a lower bound)
        for i, w in enumerate(self._WINDOWS + (None,)):
            _K.ray_window_next(starts, counts, t_starts, t_ends, sigmas, done, take, w, threshold, first=i == 0)
            ends = torch.cumsum(take, 0)
            if i == 0 and self._first_window_total is not None:
                capacity = known = int(self._first_window_total)     # came back with the march's own sample total
            else:
                capacity = n - known if w is None else min(n_rays * int(w), n - known)
            if capacity <= 0:
                break
            pos, src = _K.window_positions(starts, done, take, ends, t_starts, t_ends, rays_o, rays_d, capacity)
            n_dev = ends[-1:]
            values = evaluate(pos, n_dev)
            if values.shape[0] != capacity:
                raise AssertionError("sigmas must have shape of (N,)! Got {}".format(tuple(values.shape)))
            _K.scatter_counted(sigmas, src, values.to(sigmas.dtype), n_dev)
        return sigmas

    # ------------------------------------------------------------------------------------- upkeep
    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2,
                             ema_decay: float = 0.95, warmup_steps: int = 256, n: int = 16) -> None:
        """Refresh the grid on every n-th training step (a no-op on the others)."""
        if not self.training:
            raise RuntimeError("You should only call this function only during training. Please call _update() "
                               "directly if you want to update the field during inference.")
        if step % n:
            return
        self._update(step=step, occ_eval_fn=occ_eval_fn, occ_thre=occ_thre, ema_decay=ema_decay,
                     warmup_steps=warmup_steps)

    def _all_visible(self) -> bool:
        """No cell is marked invisible (occs >= 0 everywhere) — the state of every run that never calls
        `mark_invisible_cells`.  Then the three boolean-index round trips of an update (each a `nonzero` with a host
        sync) select everything and are skipped: same cells, same mean.  The answer is cached against the buffer's
        identity and version counter: this class's own writes keep it (they never make a cell negative), anybody else's
        write to `occs` is seen and costs one check."""
        key = (id(self.occs), self.occs._version)
        if getattr(self, "_vis_key", None) != key:
            self._vis_flag = bool((self.occs >= 0).all())
            self._vis_key = key
        return self._vis_flag

    def _stamp_visible(self) -> None:
        """After a write of this class that cannot have changed which cells are visible."""
        if getattr(self, "_vis_key", None) is not None:
            self._vis_key = (id(self.occs), self.occs._version)

    def _vis_flag_after_update(self) -> bool:
        """`_all_visible()` behind the update's own writes to `occs` (which keep every cell's sign): the cached answer,
        re-stamped with the buffer's new version."""
        flag = getattr(self, "_vis_flag", None)
        if flag is None or getattr(self, "_vis_key", (None,))[0] != id(self.occs):
            return self._all_visible()
        self._stamp_visible()
        return flag

    @torch.no_grad()
    def _get_all_cells(self) -> List[Tensor]:
        """Per level: indices of the cells that are not marked invisible (occs >= 0)."""
        if self._all_visible():
            return [self.grid_indices for _ in range(self.levels)]
        per_level = self.occs.view(self.levels, self.cells_per_lvl)
        return [self.grid_indices[per_level[k] >= 0.0] for k in range(self.levels)]

    @torch.no_grad()
    def _sample_uniform_and_occupied_cells(self, n: int) -> List[Tensor]:
        """Per level: n uniformly drawn (visible) cells followed by up to n of the occupied ones."""
        picks = []
        per_level = self.occs.view(self.levels, self.cells_per_lvl)
        all_visible = self._all_visible()
        for k in range(self.levels):
            drawn = torch.randint(self.cells_per_lvl, (n,), device=self.device)
            if not all_visible:
                drawn = drawn[per_level[k][drawn] >= 0.0]
            occupied = self.binaries[k].reshape(-1).nonzero()[:, 0]
            if occupied.shape[0] > n:
                occupied = occupied[torch.randint(occupied.shape[0], (n,), device=self.device)]
            picks.append(torch.cat([drawn, occupied]))
        return picks

    @torch.no_grad()
    def _update(self, step: int, occ_eval_fn: Callable, occ_thre: float = 0.01, ema_decay: float = 0.95,
                warmup_steps: int = 256) -> None:
        """occs <- max(decay * occs, occ_eval_fn(jittered cell position)) on every cell while warming up, on a
        quarter uniformly drawn + up to a quarter occupied cells afterwards; binaries <- occs above
        min(mean of the visible cells, occ_thre)."""
        cells = (self._get_all_cells() if step < warmup_steps
                 else self._sample_uniform_and_occupied_cells(self.cells_per_lvl // 4))
        for k, idx in enumerate(cells):
            corner = self.grid_coords[idx]
            unit = (corner + torch.rand_like(corner, dtype=torch.float32)) / self.resolution
            lo, hi = self.aabbs[k, :3], self.aabbs[k, 3:]
            seen = occ_eval_fn(lo + unit * (hi - lo)).squeeze(-1)
            slot = idx + k * self.cells_per_lvl
            self.occs[slot] = torch.maximum(self.occs[slot] * ema_decay, seen)
        # (a visible cell stays visible: max(decay * occs, .) of a non-negative value)
        visible = self.occs if self._vis_flag_after_update() else self.occs[self.occs >= 0]
        level = torch.clamp(visible.mean(), max=occ_thre)
        self.binaries = (self.occs > level).view(self.binaries.shape)

    @torch.no_grad()
    def mark_invisible_cells(self, K: Tensor, c2w: Tensor, width: int, height: int, near_plane: float = 0.0,
                             chunk: int = 32 ** 3) -> None:
        """occs <- -1 for cells that no camera sees, or that lie between some camera and its near plane while
        projecting into its image; 0 for the others.  K (n,3,3) or (1,3,3); c2w (n,3,4) or (n,4,4)."""
        if K.dim() != 3 or tuple(K.shape[1:]) != (3, 3):
            raise AssertionError("K must be (N, 3, 3)")
        if c2w.dim() != 3 or tuple(c2w.shape[1:]) not in ((3, 4), (4, 4)):
            raise AssertionError("c2w must be (N, 3, 4) or (N, 4, 4)")
        if K.shape[0] not in (1, c2w.shape[0]):
            raise AssertionError("K must hold one matrix, or one per camera")
        rot_t = c2w[:, :3, :3].transpose(1, 2)                   # world -> camera rotation
        shift = -(rot_t @ c2w[:, :3, 3:])                        # (n, 3, 1)
        steps = (self.resolution - 1).to(torch.float32)
        for k, idx in enumerate(self._get_all_cells()):
            lo, hi = self.aabbs[k, :3], self.aabbs[k, 3:]
            for a in range(0, idx.shape[0], chunk):
                part = idx[a:a + chunk]
                world = lo + (self.grid_coords[part] / steps) * (hi - lo)           # cell corners, (m, 3)
                pix = K @ (rot_t @ world.T + shift)                                  # (n, 3, m): u*d, v*d, d
                depth = pix[:, 2]
                u, v = pix[:, 0] / depth, pix[:, 1] / depth
                framed = (depth >= 0) & (u >= 0) & (u < width) & (v >= 0) & (v < height)
                seen = (framed & (depth >= near_plane)).any(dim=0)
                too_close = (framed & (depth < near_plane)).any(dim=0)
                self.occs[part + k * self.cells_per_lvl] = torch.where(seen & ~too_close, 0.0, -1.0)
