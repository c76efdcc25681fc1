"""Host mirror of the fused volume-rendering entry points of libcnc_hip.so (include/cnc_hip.h, "Fused
per-ray volume rendering").  The reference has no extension here — these replace ATen op chains in
nerfacc/volrend.py, nerfacc/pack.py and nerfacc/estimators/occ_grid.py — so the functions are named after
what they compute.  Same conventions as the other mirrors: CUDA + contiguous tensors or RuntimeError,
callee allocates results on the inputs' device, kernels on torch's current stream of that device.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import _lib
from .._lib import CNC_VOLREND_ACCUMULATE, CNC_VOLREND_FINALIZE, check, check_input, ptr, stream


def _f32(t, name):
    check_input(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    return t


def _i64(t, name):
    check_input(t, name)
    if t.dtype != torch.int64:
        raise RuntimeError(f"{name} must be int64")
    return t


def _opt(t, name):
    return None if t is None else _f32(t, name)


def _p(t):
    """Device pointer; an EMPTY tensor (no storage, data_ptr 0) is handed down as a valid dummy address —
    the kernels never touch per-sample arrays of rays that have no samples."""
    if t is None:
        return None
    if t.numel() == 0:
        return torch.empty(4, dtype=torch.float32, device=t.device).data_ptr()
    return t.data_ptr()


def split_packed(packed_info):
    """(n_rays, 2) -> contiguous (starts, counts)."""
    return packed_info[:, 0].contiguous(), packed_info[:, 1].contiguous()


def pack_bounds(ray_indices, n_rays: int):
    """Sorted per-sample ray ids -> (starts, counts) per ray, the content of pack_info's two columns.
    starts of rays without samples are the running offset, as a cumsum gives them."""
    _i64(ray_indices, "ray_indices")
    first = torch.zeros(n_rays, dtype=torch.int64, device=ray_indices.device)
    last = torch.zeros(n_rays, dtype=torch.int64, device=ray_indices.device)
    check(_lib.lib().cnc_pack_bounds(ptr(ray_indices), ray_indices.shape[0], ptr(first), ptr(last), n_rays,
                                     stream(ray_indices.device)), "pack_bounds")
    counts = last - first
    return torch.cumsum(counts, 0) - counts, counts


def volrend_forward(starts, counts, t_starts, t_ends, sigmas, rgbs=None, *, opacity_in=None, prefix_trans=None,
                    render_bkgd=None, want_samples=True, want_rays=True, accumulate_into=None, finalize=False):
    """One pass over every ray.  Returns (weights, trans, alphas, colors, opacity, depth); per-sample
    outputs are None unless `want_samples`, per-ray ones unless `want_rays`.  `accumulate_into` =
    (colors, opacity, depth) tensors to add to in place (iterative evaluation render)."""
    n_rays, S, dev = starts.shape[0], t_starts.shape[0], t_starts.device
    _i64(starts, "chunk_starts"), _i64(counts, "chunk_cnts")
    _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends"), _f32(sigmas, "sigmas")
    if sigmas.shape != t_starts.shape or t_ends.shape != t_starts.shape:
        raise RuntimeError("t_starts, t_ends and sigmas must have the same shape (N,)")
    if rgbs is not None and tuple(_f32(rgbs, "rgbs").shape) != (S, 3):
        raise RuntimeError("rgbs must have shape (N, 3)")
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    weights = trans = alphas = colors = opacity = depth = None
    if want_samples:
        weights, trans, alphas = new(S), new(S), new(S)
    flags = 0
    if accumulate_into is not None:
        colors, opacity, depth = accumulate_into
        for t, nm in ((colors, "colors"), (opacity, "opacity"), (depth, "depth")):
            _f32(t, nm)
        flags = CNC_VOLREND_ACCUMULATE
    elif want_rays:
        # rays without samples are written too (zeros, or the background when finalising)
        colors = new(n_rays, 3) if rgbs is not None else None
        opacity, depth = new(n_rays, 1), new(n_rays, 1)
        flags = CNC_VOLREND_FINALIZE if finalize else 0
    check(_lib.lib().cnc_volrend_forward(
        ptr(starts), ptr(counts), _p(t_starts), _p(t_ends), _p(sigmas), _p(rgbs), ptr(_opt(opacity_in, "opacity_in")),
        _p(_opt(prefix_trans, "prefix_trans")), ptr(_opt(render_bkgd, "render_bkgd")), _p(weights), _p(trans),
        _p(alphas), ptr(colors), ptr(opacity), ptr(depth), n_rays, flags, stream(dev)), "volrend_forward")
    return weights, trans, alphas, colors, opacity, depth


def volrend_backward(starts, counts, t_starts, t_ends, rgbs, weights, trans, alphas, *, opacity=None, depth=None,
                     render_bkgd=None, grad_colors=None, grad_opacity=None, grad_depth=None, grad_weights=None,
                     grad_trans=None, grad_alphas=None, want_grad_rgbs=True, finalize=False):
    """(grad_sigmas [N], grad_rgbs [N,3] or None)."""
    n_rays, S, dev = starts.shape[0], t_starts.shape[0], t_starts.device
    g_sig = torch.empty(S, dtype=torch.float32, device=dev)
    g_rgb = torch.empty((S, 3), dtype=torch.float32, device=dev) if (want_grad_rgbs and rgbs is not None) else None
    c = lambda t, nm: None if t is None else _f32(t.contiguous(), nm)
    check(_lib.lib().cnc_volrend_backward(
        ptr(starts), ptr(counts), _p(t_starts), _p(t_ends), _p(rgbs), _p(weights), _p(trans), _p(alphas),
        ptr(opacity), ptr(depth), ptr(_opt(render_bkgd, "render_bkgd")), ptr(c(grad_colors, "grad_colors")),
        ptr(c(grad_opacity, "grad_opacity")), ptr(c(grad_depth, "grad_depth")), _p(c(grad_weights, "grad_weights")),
        _p(c(grad_trans, "grad_trans")), _p(c(grad_alphas, "grad_alphas")), _p(g_sig), _p(g_rgb), n_rays,
        CNC_VOLREND_FINALIZE if finalize else 0, stream(dev)), "volrend_backward")
    return g_sig, g_rgb


def render_visibility(starts, counts, values, t_starts=None, t_ends=None, *, from_alpha=False,
                      early_stop_eps=1e-4, alpha_thre=0.0, alpha_thre_cap=None, want_kept=True):
    """(mask uint8 [N], kept int64 [n_rays] or None)."""
    n_rays, S, dev = starts.shape[0], values.shape[0], values.device
    _i64(starts, "chunk_starts"), _i64(counts, "chunk_cnts"), _f32(values, "sigmas/alphas")
    if not from_alpha:
        _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends")
    mask = torch.empty(S, dtype=torch.uint8, device=dev)
    kept = torch.empty(n_rays, dtype=torch.int64, device=dev) if want_kept else None
    check(_lib.lib().cnc_render_visibility(
        ptr(starts), ptr(counts), _p(t_starts), _p(t_ends), _p(values), int(bool(from_alpha)), float(early_stop_eps),
        float(alpha_thre), ptr(_opt(alpha_thre_cap, "alpha_thre_cap")), _p(mask), ptr(kept), n_rays, stream(dev)),
        "render_visibility")
    return mask, kept


def compact_samples(starts, counts, mask, kept, t_starts, t_ends) -> Tuple[torch.Tensor, ...]:
    """Survivors of `mask`, in order: (ray_indices, t_starts, t_ends, new_starts, kept).  ONE host sync (the
    survivor total sizes the result); the reference's three boolean-index gathers each have one."""
    n_rays, dev = starts.shape[0], t_starts.device
    ends = torch.cumsum(kept, 0)
    total = int(ends[-1].item()) if n_rays else 0
    new_starts = ends - kept
    o_s = torch.empty(total, dtype=torch.float32, device=dev)
    o_e = torch.empty(total, dtype=torch.float32, device=dev)
    o_r = torch.empty(total, dtype=torch.int64, device=dev)
    if total:
        check(_lib.lib().cnc_compact_samples(ptr(starts), ptr(counts), ptr(new_starts), ptr(mask), ptr(t_starts),
                                             ptr(t_ends), ptr(o_s), ptr(o_e), ptr(o_r), n_rays, stream(dev)),
              "compact_samples")
    return o_r, o_s, o_e, new_starts, kept


def window_samples(starts, first, cnts, t_starts, t_ends, total, ends=None):
    """(ray_indices, t_starts, t_ends, source_index) of samples [first[r], first[r] + cnts[r]) of every ray; `total` =
    cnts.sum() (the caller read it back together with whatever else it needed); `ends` = cumsum(cnts) if the caller
    has it."""
    n_rays, dev = starts.shape[0], t_starts.device
    out_starts = (torch.cumsum(cnts, 0) if ends is None else ends) - cnts
    o_s = torch.empty(total, dtype=torch.float32, device=dev)
    o_e = torch.empty(total, dtype=torch.float32, device=dev)
    o_r = torch.empty(total, dtype=torch.int64, device=dev)
    o_i = torch.empty(total, dtype=torch.int64, device=dev)
    if total:
        check(_lib.lib().cnc_ray_window_samples(ptr(starts), ptr(first), ptr(cnts), ptr(out_starts), ptr(t_starts),
                                                ptr(t_ends), ptr(o_s), ptr(o_e), ptr(o_r), ptr(o_i), n_rays, stream(dev)),
              "ray_window_samples")
    return o_r, o_s, o_e, o_i


def window_positions(starts, first, cnts, ends, t_starts, t_ends, rays_o, rays_d, capacity):
    """(positions [capacity, 3], source_index [capacity]) of samples [first[r], first[r] + cnts[r]) of every ray, packed in
    ray order; `ends` = cumsum(cnts) — its last element is the number of rows that are written, and it stays on the
    device (cnc_ray_window_positions).  `capacity` >= that number (the caller's bound); rows behind it are left as they
    are."""
    n_rays, dev = starts.shape[0], t_starts.device
    pos = torch.empty((capacity, 3), dtype=torch.float32, device=dev)
    src = torch.empty(capacity, dtype=torch.int64, device=dev)
    if capacity and n_rays:
        check(_lib.lib().cnc_ray_window_positions(ptr(starts), ptr(first), ptr(cnts), ptr(ends), _p(t_starts), _p(t_ends),
                                                  _p(rays_o), _p(rays_d), ptr(pos), ptr(src), n_rays, stream(dev)),
              "ray_window_positions")
    return pos, src


def scatter_counted(out, src, values, n_dev):
    """out[src[i]] = values[i] for i < min(n_dev, len(src)); `n_dev`: one int64 on the device."""
    if src.shape[0]:
        check(_lib.lib().cnc_scatter_counted(_p(values), ptr(src), ptr(out), ptr(n_dev), int(src.shape[0]), stream(out.device)),
              "scatter_counted")


def ray_transmittance(starts, cnts, t_starts, t_ends, sigmas):
    """exp(-sum sigma dt) over the first cnts[r] samples of every ray: float32 [n_rays]."""
    n_rays = starts.shape[0]
    out = torch.empty(n_rays, dtype=torch.float32, device=sigmas.device)
    check(_lib.lib().cnc_ray_transmittance(ptr(starts), ptr(cnts), _p(t_starts), _p(t_ends), _p(sigmas), ptr(out), n_rays,
                                           stream(sigmas.device)), "ray_transmittance")
    return out


def ray_window_next(starts, counts, t_starts, t_ends, sigmas, done, take, window, threshold, first):
    """One step of the front-to-back sampler, in place on `done` / `take` (int64 [n_rays]): see cnc_ray_window_next."""
    n_rays = starts.shape[0]
    check(_lib.lib().cnc_ray_window_next(ptr(starts), ptr(counts), _p(t_starts), _p(t_ends), _p(sigmas), ptr(done), ptr(take),
                                         -1 if window is None else int(window), float(threshold), int(bool(first)), n_rays,
                                         stream(sigmas.device)), "ray_window_next")


def samples_from_intervals(intervals, sample_counts, total: Optional[int] = None):
    """(ray_indices, t_starts, t_ends, starts) of the samples a traverse_grids call produced, from its
    interval edges — `intervals` is the RaySegmentsSpec the extension returned (two-pass or over-allocated
    layout), `sample_counts` the samples' chunk_cnts.  `total` = number of samples when the caller already
    knows it (two-pass mode: samples.vals.numel()); otherwise it is read back (one sync)."""
    n_rays, dev = sample_counts.shape[0], sample_counts.device
    ends = torch.cumsum(sample_counts, 0)
    if total is None:
        total = int(ends[-1].item()) if n_rays else 0
    starts = ends - sample_counts
    o_s = torch.empty(total, dtype=torch.float32, device=dev)
    o_e = torch.empty(total, dtype=torch.float32, device=dev)
    o_r = torch.empty(total, dtype=torch.int64, device=dev)
    if total:
        iv_starts = getattr(intervals, "alloc_starts", None)
        if iv_starts is None:
            iv_starts = intervals.chunk_starts
        check(_lib.lib().cnc_interval_edges_to_samples(
            ptr(iv_starts), ptr(intervals.chunk_cnts), ptr(intervals.vals), ptr(intervals.is_left),
            ptr(intervals.is_right), ptr(starts), ptr(o_s), ptr(o_e), ptr(o_r), n_rays, stream(dev)),
            "interval_edges_to_samples")
    return o_r, o_s, o_e, starts
