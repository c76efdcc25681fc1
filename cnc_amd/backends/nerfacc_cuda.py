"""Drop-in for the reference's `nerfacc.csrc` extension (nerfacc/cuda/csrc/nerfacc.cpp:100-129):
ray_aabb_intersect, traverse_grids, the six segmented scans and the RaySegmentsSpec record.

The callee allocates and returns tensors on the inputs' device, like the reference's host
functions; the allocation / cumsum logic of grid.cu:356-510 and data_spec.hpp:53-106 lives here
(torch ops), the kernels behind the C ABI.
"""
from __future__ import annotations

import ctypes as C

import os

import torch

from .. import _lib
from .._lib import RaySegments, check, check_input, ptr, stream


class RaySegmentsSpec:
    """Mirror of RaySegmentsSpec (data_spec.hpp:6-13 / nerfacc.cpp:120-128): undefined tensors
    read as None from Python."""

    # alloc_starts (extension): where each ray's slots begin in `vals` when the march over-allocated
    # (chunk_starts is then recomputed from the true counts, data_spec.hpp:99-106, and no longer says)
    __slots__ = ("vals", "is_left", "is_right", "is_valid", "chunk_starts", "chunk_cnts",
                 "ray_indices", "alloc_starts")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)

    # data_spec.hpp:62-84
    def memalloc_data(self, size, alloc_masks=True, zero_init=True, alloc_valid=False):
        assert self.chunk_cnts is not None and self.vals is None
        dev = self.chunk_cnts.device
        mk = torch.zeros if zero_init else torch.empty
        self.vals = mk(size, dtype=torch.float32, device=dev)
        self.ray_indices = mk(size, dtype=torch.int64, device=dev)
        if alloc_masks:
            self.is_left = mk(size, dtype=torch.bool, device=dev)
            self.is_right = mk(size, dtype=torch.bool, device=dev)
        if alloc_valid:
            self.is_valid = torch.zeros(size, dtype=torch.bool, device=dev)

    # data_spec.hpp:86-96 (the .item() is the reference's host sync too)
    def memalloc_data_from_chunk(self, alloc_masks=True, zero_init=True, alloc_valid=False):
        assert self.chunk_cnts is not None and self.chunk_starts is None
        cumsum = torch.cumsum(self.chunk_cnts, 0, dtype=self.chunk_cnts.dtype)
        n_edges = int(cumsum[-1].item()) if cumsum.numel() else 0
        self.chunk_starts = cumsum - self.chunk_cnts
        self.memalloc_data(n_edges, alloc_masks, zero_init, alloc_valid)

    # data_spec.hpp:99-106
    def compute_chunk_start(self):
        if self.chunk_cnts is None:
            return
        cumsum = torch.cumsum(self.chunk_cnts, 0, dtype=self.chunk_cnts.dtype)
        self.chunk_starts = cumsum - self.chunk_cnts

    def _view(self) -> RaySegments:
        return RaySegments(ptr(self.vals), ptr(self.chunk_starts), ptr(self.chunk_cnts),
                           ptr(self.ray_indices), ptr(self.is_left), ptr(self.is_right),
                           ptr(self.is_valid))


def ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane, far_plane, miss_value):
    n_rays, n_aabbs = rays_o.shape[0], aabbs.shape[0]
    for name, t in (("rays_o", rays_o), ("rays_d", rays_d), ("aabbs", aabbs)):
        check_input(t, name)
    t_mins = torch.empty((n_rays, n_aabbs), dtype=rays_o.dtype, device=rays_o.device)
    t_maxs = torch.empty((n_rays, n_aabbs), dtype=rays_o.dtype, device=rays_o.device)
    hits = torch.empty((n_rays, n_aabbs), dtype=torch.bool, device=rays_o.device)
    rc = _lib.lib().cnc_ray_aabb_intersect(ptr(rays_o), ptr(rays_d), ptr(aabbs), n_rays, n_aabbs,
                                           float(near_plane), float(far_plane), float(miss_value),
                                           ptr(t_mins), ptr(t_maxs), ptr(hits), stream(rays_o.device))
    check(rc, "ray_aabb_intersect")
    return [t_mins, t_maxs, hits]


def traverse_grids(rays_o, rays_d, rays_mask, binaries, aabbs, t_sorted, t_indices, hits,
                   near_planes, far_planes, step_size, cone_angle, compute_intervals,
                   compute_samples, compute_terminate_planes, traverse_steps_limit, over_allocate):
    """grid.cu:356-510."""
    if over_allocate and not traverse_steps_limit > 0:
        raise RuntimeError("traverse_steps_limit must be > 0 when over_allocate is true")
    for name, t in (("rays_o", rays_o), ("rays_d", rays_d), ("rays_mask", rays_mask),
                    ("binaries", binaries), ("aabbs", aabbs), ("t_sorted", t_sorted),
                    ("t_indices", t_indices), ("hits", hits), ("near_planes", near_planes),
                    ("far_planes", far_planes)):
        check_input(t, name)
    n_rays = rays_o.shape[0]
    n_grids = binaries.shape[0]
    dev = rays_o.device
    L = _lib.lib()
    intervals, samples = RaySegmentsSpec(), RaySegmentsSpec()
    terminate_planes = (torch.empty(n_rays, dtype=rays_o.dtype, device=dev)
                        if compute_terminate_planes else None)

    def launch(mask, first_pass, term):
        iv, sm = intervals._view(), samples._view()
        rc = L.cnc_traverse_grids(ptr(rays_o), ptr(rays_d), ptr(mask), n_rays, ptr(binaries),
                                  n_grids, binaries.shape[1], binaries.shape[2], binaries.shape[3],
                                  ptr(aabbs), ptr(hits), ptr(t_sorted), ptr(t_indices),
                                  ptr(near_planes), ptr(far_planes), float(step_size),
                                  float(cone_angle), int(traverse_steps_limit), int(first_pass),
                                  C.byref(iv), C.byref(sm), ptr(term), stream(rays_o.device))
        check(rc, "traverse_grids")

    if over_allocate:
        # single pass into an upper-bound allocation (grid.cu:400-440)
        if compute_intervals:
            intervals.chunk_cnts = torch.full((n_rays,), traverse_steps_limit * 2, dtype=torch.int64,
                                              device=dev) * rays_mask
            intervals.memalloc_data_from_chunk(True, True)
        if compute_samples:
            samples.chunk_cnts = torch.full((n_rays,), traverse_steps_limit, dtype=torch.int64,
                                            device=dev) * rays_mask
            samples.memalloc_data_from_chunk(False, True, True)
        n_out = (intervals.vals.numel() if compute_intervals else 0) + \
                (samples.vals.numel() if compute_samples else 0)
        if n_out > 0:   # every ray masked out: nothing to march
            launch(rays_mask, 0, terminate_planes)
        intervals.alloc_starts, samples.alloc_starts = intervals.chunk_starts, samples.chunk_starts
        intervals.compute_chunk_start()
        samples.compute_chunk_start()
    else:
        # count, allocate exactly, fill (grid.cu:441-507); note: rays_mask is NOT applied here
        if compute_intervals:
            intervals.chunk_cnts = torch.empty(n_rays, dtype=torch.int64, device=dev)
        if compute_samples:
            samples.chunk_cnts = torch.empty(n_rays, dtype=torch.int64, device=dev)
        launch(None, 1, None)
        if compute_intervals:
            intervals.memalloc_data_from_chunk(True, True)
        if compute_samples:
            samples.memalloc_data_from_chunk(False, False, True)
        n_out = (intervals.vals.numel() if compute_intervals else 0) + \
                (samples.vals.numel() if compute_samples else 0)
        if n_out > 0:   # with nothing to fill every ray is skipped (grid.cu:103-106): no launch needed
            launch(None, 0, terminate_planes)
    return intervals, samples, terminate_planes


_COARSE = {}
_COARSE_ON = os.environ.get("CNC_MARCH_COARSE", "1") == "1"      # measurement / test switch


def occupancy_coarse_bits(binaries):
    """(extension) the coarse occupancy of `binaries` [n_grids, rx, ry, rz] for the march (cnc_occupancy_coarse_bits: one
    bit per block of 4 x 4 x 4 cells), or None when the kernels take none for this shape.  Cached on the grid tensor
    (address, version, shape): the estimator replaces its grid at every occupancy update."""
    if not _COARSE_ON or binaries.dim() != 4 or binaries.dtype not in (torch.bool, torch.uint8):
        return None
    key = (binaries.data_ptr(), binaries._version, tuple(binaries.shape), str(binaries.device))
    hit = _COARSE.get(key)
    if hit is not None:
        return hit[1]
    L = _lib.lib()
    nw = int(L.cnc_occupancy_coarse_words(*[int(v) for v in binaries.shape]))
    words = None
    if nw:
        words = torch.empty(nw, dtype=torch.int32, device=binaries.device)
        check(L.cnc_occupancy_coarse_bits(ptr(binaries), *[int(v) for v in binaries.shape], ptr(words), stream(binaries.device)),
              "occupancy_coarse_bits")
    if len(_COARSE) >= 8:
        _COARSE.clear()
    _COARSE[key] = (binaries, words)        # (the grid is kept: its address stays unique while the entry lives)
    return words


def march_samples(rays_o, rays_d, rays_mask, binaries, aabbs, t_sorted, t_indices, hits, near_planes, far_planes,
                  step_size, cone_angle, traverse_steps_limit=-1, want_terminate_planes=False, clamped_total=None,
                  extras=None):
    """(extension) The march as the renderer consumes it — see cnc_march_samples in include/cnc_hip.h.
    Returns (ray_indices i64 [S], t_starts [S], t_ends [S], chunk_starts [n_rays], chunk_cnts [n_rays],
    terminate_planes or None).  Count pass, exclusive cumsum + ONE host sync (the sample total sizes the
    result, as in data_spec.hpp:86-96), fill pass.  `clamped_total` = {"at": w}: the same sync also brings
    sum(min(count, w)) back, under the key "total" (a caller that is about to take the first w samples of every ray).
    `extras` (dict, filled in place): what the fill pass emits besides — "positions": True -> [S,3] sample positions
    (o + d (t0 + t1) / 2, in the unit cube of extras["aabb"] (6 floats on the device) when given), "dirs": True ->
    [S,3] ray directions, "ray_indices": "int32" | None -> the ray ids as int32 / not at all (the first return value
    is then that tensor / None)."""
    for name, t in (("rays_o", rays_o), ("rays_d", rays_d), ("binaries", binaries), ("aabbs", aabbs),
                    ("t_sorted", t_sorted), ("t_indices", t_indices), ("hits", hits),
                    ("near_planes", near_planes), ("far_planes", far_planes)):
        check_input(t, name)
    if rays_mask is not None:
        check_input(rays_mask, "rays_mask")
    n_rays, dev = rays_o.shape[0], rays_o.device
    counts = torch.zeros(n_rays, dtype=torch.int64, device=dev)       # masked-out rays stay at 0
    term = torch.empty(n_rays, dtype=rays_o.dtype, device=dev) if want_terminate_planes else None
    L = _lib.lib()
    # where each ray's first sample was produced: left by the count pass, picked up by the fill pass, which then marches
    # only the span between a ray's first and last sample (CNC_MARCH_RESUME=0: both passes march the whole ray)
    resume = torch.empty((n_rays, 8), dtype=torch.int32, device=dev) if os.environ.get("CNC_MARCH_RESUME", "1") == "1" \
        else None
    ex = extras if extras is not None else {}
    ri_kind = ex.get("ray_indices", "int64")
    if ri_kind not in ("int64", "int32", None):
        raise RuntimeError("march_samples: extras['ray_indices'] must be 'int64', 'int32' or None")
    if ri_kind is None and not ex.get("positions"):
        raise RuntimeError("march_samples: without ray indices the fill pass must at least emit positions")
    box = ex.get("aabb")
    if box is not None:
        check_input(box, "extras['aabb']")
        if box.dtype != torch.float32 or box.numel() != 6:
            raise RuntimeError("march_samples: extras['aabb'] must be 6 float32 values on the device")

    coarse = occupancy_coarse_bits(binaries)      # empty blocks of 4^3 cells are skipped from LDS (same samples)

    def launch(starts, t0, t1, ri, tp, pos=None, dirs=None, ri32=None):
        rc = L.cnc_march_samples_coarse(ptr(rays_o), ptr(rays_d), ptr(rays_mask), n_rays, ptr(binaries), binaries.shape[0],
                                        binaries.shape[1], binaries.shape[2], binaries.shape[3], ptr(aabbs), ptr(hits),
                                        ptr(t_sorted), ptr(t_indices), ptr(near_planes), ptr(far_planes), float(step_size),
                                        float(cone_angle), int(traverse_steps_limit), ptr(counts), ptr(starts), ptr(t0),
                                        ptr(t1), ptr(ri), ptr(tp), ptr(resume), ptr(pos), ptr(dirs), ptr(ri32), ptr(box),
                                        ptr(coarse), stream(dev))
        check(rc, "march_samples")

    launch(None, None, None, None, term)
    ends = torch.cumsum(counts, 0)
    if clamped_total is not None and n_rays:
        total, clamped_total["total"] = torch.stack([ends[-1], counts.clamp(max=int(clamped_total["at"])).sum()]).tolist()
    else:
        total = int(ends[-1].item()) if n_rays else 0
    starts = ends - counts
    t_starts = torch.empty(total, dtype=torch.float32, device=dev)
    t_ends = torch.empty(total, dtype=torch.float32, device=dev)
    ray_indices = torch.empty(total, dtype=torch.int64, device=dev) if ri_kind == "int64" else None
    ri32 = torch.empty(total, dtype=torch.int32, device=dev) if ri_kind == "int32" else None
    pos = torch.empty((total, 3), dtype=torch.float32, device=dev) if ex.get("positions") else None
    dirs = torch.empty((total, 3), dtype=torch.float32, device=dev) if ex.get("dirs") else None
    if total:
        launch(starts, t_starts, t_ends, ray_indices, None, pos, dirs, ri32)
    if extras is not None:
        if ex.get("positions"):
            extras["positions"] = pos
        if ex.get("dirs"):
            extras["dirs"] = dirs
    return (ri32 if ri_kind == "int32" else ray_indices), t_starts, t_ends, starts, counts, term


def sample_positions(rays_o, rays_d, ray_indices, t_a, t_b=None, aabb=None, want_dirs=False):
    """(extension) positions [S,3] (and directions) of ray samples in one kernel; see
    cnc_sample_positions in include/cnc_hip.h."""
    for name, t in (("rays_o", rays_o), ("rays_d", rays_d), ("ray_indices", ray_indices), ("t_a", t_a)):
        check_input(t, name)
    if ray_indices.dtype != torch.int64 or rays_o.dtype != torch.float32 or t_a.dtype != torch.float32:
        raise RuntimeError("sample_positions: ray_indices must be int64, rays and t float32")
    S = ray_indices.shape[0]
    pos = torch.empty((S, 3), dtype=torch.float32, device=rays_o.device)
    dirs = torch.empty((S, 3), dtype=torch.float32, device=rays_o.device) if want_dirs else None
    if t_b is not None:
        check_input(t_b, "t_b")
    if aabb is not None:
        aabb = aabb.reshape(-1).to(torch.float32).contiguous()
        if aabb.numel() != 6:
            raise RuntimeError("sample_positions: aabb must hold 6 values")
    rc = _lib.lib().cnc_sample_positions(ptr(rays_o), ptr(rays_d), ptr(ray_indices), ptr(t_a), ptr(t_b),
                                         ptr(aabb), S, ptr(pos), ptr(dirs), stream(rays_o.device))
    check(rc, "sample_positions")
    return (pos, dirs) if want_dirs else pos


def _scan_checks(chunk_starts, chunk_cnts, inputs):
    check_input(chunk_starts, "chunk_starts")
    check_input(chunk_cnts, "chunk_cnts")
    check_input(inputs, "inputs")
    if chunk_starts.dim() != 1 or chunk_cnts.dim() != 1 or inputs.dim() != 1 \
            or chunk_starts.shape[0] != chunk_cnts.shape[0]:
        raise RuntimeError("Expected 1-D chunk_starts/chunk_cnts of equal length and 1-D inputs")


def _sum(fn_name, chunk_starts, chunk_cnts, inputs, normalize, backward):
    _scan_checks(chunk_starts, chunk_cnts, inputs)
    outputs = torch.empty_like(inputs)
    if inputs.shape[0] == 0:
        return outputs
    rc = getattr(_lib.lib(), fn_name)(ptr(chunk_starts), ptr(chunk_cnts), ptr(inputs), ptr(outputs),
                                      chunk_cnts.shape[0], inputs.shape[0], int(bool(normalize)),
                                      int(bool(backward)), stream(inputs.device))
    check(rc, fn_name)
    return outputs


def inclusive_sum(chunk_starts, chunk_cnts, inputs, normalize, backward):
    return _sum("cnc_inclusive_sum", chunk_starts, chunk_cnts, inputs, normalize, backward)


def exclusive_sum(chunk_starts, chunk_cnts, inputs, normalize, backward):
    return _sum("cnc_exclusive_sum", chunk_starts, chunk_cnts, inputs, normalize, backward)


def _prod_fwd(fn_name, chunk_starts, chunk_cnts, inputs):
    _scan_checks(chunk_starts, chunk_cnts, inputs)
    outputs = torch.empty_like(inputs)
    if inputs.shape[0] == 0:
        return outputs
    rc = getattr(_lib.lib(), fn_name)(ptr(chunk_starts), ptr(chunk_cnts), ptr(inputs), ptr(outputs),
                                      chunk_cnts.shape[0], inputs.shape[0], stream(inputs.device))
    check(rc, fn_name)
    return outputs


def inclusive_prod_forward(chunk_starts, chunk_cnts, inputs):
    return _prod_fwd("cnc_inclusive_prod_forward", chunk_starts, chunk_cnts, inputs)


def exclusive_prod_forward(chunk_starts, chunk_cnts, inputs):
    return _prod_fwd("cnc_exclusive_prod_forward", chunk_starts, chunk_cnts, inputs)


def _prod_bwd(fn_name, chunk_starts, chunk_cnts, inputs, outputs, grad_outputs):
    _scan_checks(chunk_starts, chunk_cnts, inputs)
    check_input(grad_outputs, "grad_outputs")
    grad_inputs = torch.empty_like(grad_outputs)
    if inputs.shape[0] == 0:
        return grad_inputs
    rc = getattr(_lib.lib(), fn_name)(ptr(chunk_starts), ptr(chunk_cnts), ptr(inputs),
                                      ptr(outputs.contiguous()), ptr(grad_outputs),
                                      ptr(grad_inputs), chunk_cnts.shape[0], inputs.shape[0],
                                      stream(inputs.device))
    check(rc, fn_name)
    return grad_inputs


def inclusive_prod_backward(chunk_starts, chunk_cnts, inputs, outputs, grad_outputs):
    return _prod_bwd("cnc_inclusive_prod_backward", chunk_starts, chunk_cnts, inputs, outputs,
                     grad_outputs)


def exclusive_prod_backward(chunk_starts, chunk_cnts, inputs, outputs, grad_outputs):
    return _prod_bwd("cnc_exclusive_prod_backward", chunk_starts, chunk_cnts, inputs, outputs,
                     grad_outputs)


def _not_built(name):
    def f(*a, **k):
        raise NotImplementedError(
            f"{name}: outside the CNC hot path (never called by examples/train_CNC_*.py); "
            "not built in cnc_amd")
    return f


# bound by the reference module but unused by CNC
importance_sampling = _not_built("importance_sampling")
searchsorted = _not_built("searchsorted")
opencv_lens_undistortion = _not_built("opencv_lens_undistortion")
opencv_lens_undistortion_fisheye = _not_built("opencv_lens_undistortion_fisheye")
