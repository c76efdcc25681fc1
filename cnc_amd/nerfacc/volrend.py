"""Volume rendering over ray samples — the public functions of the reference's nerfacc/volrend.py
(`rendering` :14-157, `render_*_from_density/alpha` :160-475, `accumulate_along_rays(_)` :478-575), same
names, arguments and return tuples, built around ONE fused HIP kernel per direction
(cnc_amd/csrc/volrend.hip) instead of the reference's ATen chain
`pack_info -> exclusive_sum -> exp -> 1-exp -> mul -> 3 x index_add_`.

Two sample layouts, as in the reference: *flattened* (1-D tensors + `ray_indices` / `packed_info`; the GPU
path, fused) and *batched* ((n_rays, n_samples) tensors scanned along the last axis; plain torch, any
device).  CNC's fork of `rendering` calls `rgb_sigma_fn` for THREE values (rgbs, sigmas, positions) and
reports them in `extras` (reference volrend.py:89,108-115).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
from torch import Tensor

from ..backends import volrend_backend as _K
from .pack import pack_info


# ---------------------------------------------------------------------------------------------------
# ray layout of flattened samples
# ---------------------------------------------------------------------------------------------------
def _ray_chunks(n_samples_like: Tensor, packed_info, ray_indices, n_rays):
    """(starts, counts) int64 per ray for flattened samples, or None for the batched layout."""
    if packed_info is not None:
        return _K.split_packed(packed_info)
    if ray_indices is None:
        return None
    if n_rays is None:
        raise ValueError("n_rays is required with ray_indices")
    if ray_indices.is_cuda:
        return _K.pack_bounds(ray_indices.contiguous(), int(n_rays))
    return _K.split_packed(pack_info(ray_indices, n_rays))


def _fused(chunks, *tensors) -> bool:
    return chunks is not None and all(t is None or t.is_cuda for t in tensors)


def _loop_chunks(chunks):
    starts, counts = (c.tolist() for c in chunks)
    return [(s, s + n) for s, n in zip(starts, counts)]


class _WeightsFromDensity(torch.autograd.Function):
    """(weights, trans, alphas) of flattened samples; differentiable w.r.t. sigmas through all three."""

    @staticmethod
    def forward(ctx, sigmas, t_starts, t_ends, starts, counts, prefix_trans):
        ctx.set_materialize_grads(False)           # unused outputs arrive as None, not as zero tensors
        sigmas = sigmas.contiguous()
        weights, trans, alphas, _, _, _ = _K.volrend_forward(starts, counts, t_starts, t_ends, sigmas,
                                                             prefix_trans=prefix_trans, want_rays=False)
        ctx.save_for_backward(starts, counts, t_starts, t_ends, weights, trans, alphas)
        return weights, trans, alphas

    @staticmethod
    def backward(ctx, g_w, g_t, g_a):
        starts, counts, t_starts, t_ends, weights, trans, alphas = ctx.saved_tensors
        g_sig, _ = _K.volrend_backward(starts, counts, t_starts, t_ends, None, weights, trans, alphas,
                                       grad_weights=g_w, grad_trans=g_t, grad_alphas=g_a)
        return g_sig, None, None, None, None, None


class _Composite(torch.autograd.Function):
    """sigmas, rgbs -> colours (+ background), opacities, depths (normalised) per ray, and the per-sample
    weights / transmittance / alphas, in one kernel; the backward is one kernel too."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, t_starts, t_ends, starts, counts, render_bkgd):
        ctx.set_materialize_grads(False)
        sigmas, rgbs = sigmas.contiguous(), rgbs.contiguous()
        bk = None if render_bkgd is None else render_bkgd.to(torch.float32).reshape(-1).contiguous()
        weights, trans, alphas, colors, opacity, depth = _K.volrend_forward(
            starts, counts, t_starts, t_ends, sigmas, rgbs, render_bkgd=bk, finalize=True)
        ctx.save_for_backward(starts, counts, t_starts, t_ends, rgbs, weights, trans, alphas, opacity, depth, bk)
        ctx.mark_non_differentiable(trans, alphas)
        return colors, opacity, depth, weights, trans, alphas

    @staticmethod
    def backward(ctx, g_c, g_o, g_d, g_w, _g_t, _g_a):
        starts, counts, t_starts, t_ends, rgbs, weights, trans, alphas, opacity, depth, bk = ctx.saved_tensors
        g_sig, g_rgb = _K.volrend_backward(
            starts, counts, t_starts, t_ends, rgbs, weights, trans, alphas, opacity=opacity, depth=depth,
            render_bkgd=bk, grad_colors=g_c, grad_opacity=g_o, grad_depth=g_d, grad_weights=g_w, finalize=True)
        return g_sig, g_rgb, None, None, None, None, None


# ---------------------------------------------------------------------------------------------------
# the entry point the drivers use
# ---------------------------------------------------------------------------------------------------
def rendering(t_starts: Tensor, t_ends: Tensor, ray_indices: Optional[Tensor] = None,
              n_rays: Optional[int] = None, rgb_sigma_fn: Optional[Callable] = None,
              rgb_alpha_fn: Optional[Callable] = None, render_bkgd: Optional[Tensor] = None,
              packed_info: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor, Dict]:
    """Query the field at the samples and composite: returns (colors (n_rays,3), opacities (n_rays,1),
    depths (n_rays,1), extras).  `packed_info` (extension) spares the (start, count) reconstruction when the
    sampler already has it."""
    if rgb_sigma_fn is None and rgb_alpha_fn is None:
        raise ValueError("At least one of `rgb_sigma_fn` and `rgb_alpha_fn` should be specified.")
    flat = ray_indices is not None
    if flat and not (t_starts.shape == t_ends.shape == ray_indices.shape):
        raise AssertionError("Since nerfacc 0.5.0, t_starts, t_ends and ray_indices must have the same shape (N,). ")
    dev, empty = t_starts.device, t_starts.shape[0] == 0
    from_density = rgb_sigma_fn is not None
    positions = None
    if empty:
        rgbs, dens = torch.empty((0, 3), device=dev), torch.empty((0,), device=dev)
    elif from_density:
        rgbs, dens, positions = rgb_sigma_fn(t_starts, t_ends, ray_indices)
    else:
        rgbs, dens = rgb_alpha_fn(t_starts, t_ends, ray_indices)
    if rgbs.shape[-1] != 3:
        raise AssertionError("rgbs must have 3 channels, got {}".format(rgbs.shape))
    if dens.shape != t_starts.shape:
        raise AssertionError("{} must have shape of (N,)! Got {}".format("sigmas" if from_density else "alphas", dens.shape))

    chunks = _ray_chunks(t_starts, packed_info, ray_indices, n_rays) if flat else None
    if from_density and _fused(chunks, t_starts, rgbs, dens) and rgbs.dtype == torch.float32:
        colors, opacities, depths, weights, trans, alphas = _Composite.apply(
            dens, rgbs, t_starts.contiguous(), t_ends.contiguous(), chunks[0], chunks[1], render_bkgd)
        return colors, opacities, depths, {"weights": weights, "alphas": alphas, "trans": trans, "sigmas": dens,
                                            "rgbs": rgbs, "positions": positions}

    # generic route: alpha inputs, batched layout, CPU tensors
    if from_density:
        weights, trans, alphas = render_weight_from_density(t_starts, t_ends, dens, ray_indices=ray_indices,
                                                            n_rays=n_rays, packed_info=packed_info)
        extras = {"weights": weights, "alphas": alphas, "trans": trans, "sigmas": dens, "rgbs": rgbs,
                  "positions": positions}
    else:
        alphas = dens
        weights, trans = render_weight_from_alpha(alphas, ray_indices=ray_indices, n_rays=n_rays,
                                                  packed_info=packed_info)
        extras = {"weights": weights, "trans": trans, "rgbs": rgbs, "alphas": alphas}
    mids = ((t_starts + t_ends) / 2.0).unsqueeze(-1)
    colors = accumulate_along_rays(weights, rgbs, ray_indices, n_rays)
    opacities = accumulate_along_rays(weights, None, ray_indices, n_rays)
    depths = accumulate_along_rays(weights, mids, ray_indices, n_rays) / opacities.clamp_min(torch.finfo(rgbs.dtype).eps)
    if render_bkgd is not None:
        colors = colors + render_bkgd * (1.0 - opacities)
    return colors, opacities, depths, extras


# ---------------------------------------------------------------------------------------------------
# transmittance / weights / visibility
# ---------------------------------------------------------------------------------------------------
def _optical_depth_before(tau: Tensor) -> Tensor:
    """Sum of the earlier samples' sigma*dt along the last axis (an exclusive prefix sum)."""
    return torch.cumsum(torch.nn.functional.pad(tau[..., :-1], (1, 0)), dim=-1)


def _density_terms(t_starts, t_ends, sigmas, chunks, prefix_trans):
    """(trans, alphas) in plain torch: batched rows, or flattened samples walked ray by ray (CPU route)."""
    tau = sigmas * (t_ends - t_starts)
    if chunks is None:
        before = _optical_depth_before(tau)
    elif tau.numel():
        before = torch.cat([_optical_depth_before(tau[a:b]) for a, b in _loop_chunks(chunks)])
    else:
        before = tau
    trans = torch.exp(-before)
    if prefix_trans is not None:
        trans = trans * prefix_trans
    return trans, 1.0 - torch.exp(-tau)


def render_weight_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor,
                               packed_info: Optional[Tensor] = None, ray_indices: Optional[Tensor] = None,
                               n_rays: Optional[int] = None,
                               prefix_trans: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """w_i = T_i (1 - exp(-sigma_i dt_i)), T_i = exp(-sum_{j<i} sigma_j dt_j).  Returns (weights,
    transmittance, alphas).

    >>> render_weight_from_density(t_starts=[0,1,2,3,4,5,6], t_ends=[1,2,3,4,5,6,7],
    ...                            sigmas=[.4,.8,.1,.8,.1,.0,.9], ray_indices=[0,0,0,1,1,2,2])
    weights [0.33, 0.37, 0.03 | 0.55, 0.04 | 0.00, 0.59], trans [1.00, 0.67, 0.30 | 1.00, 0.45 | 1.00, 1.00]
    """
    chunks = _ray_chunks(t_starts, packed_info, ray_indices, n_rays)
    if _fused(chunks, t_starts, sigmas, prefix_trans) and sigmas.dtype == torch.float32:
        pt = None if prefix_trans is None else prefix_trans.contiguous()
        return _WeightsFromDensity.apply(sigmas, t_starts.contiguous(), t_ends.contiguous(), chunks[0], chunks[1], pt)
    trans, alphas = _density_terms(t_starts, t_ends, sigmas, chunks, prefix_trans)
    return trans * alphas, trans, alphas


def render_transmittance_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor,
                                      packed_info: Optional[Tensor] = None,
                                      ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None,
                                      prefix_trans: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """(transmittance, alphas); see `render_weight_from_density`."""
    _, trans, alphas = render_weight_from_density(t_starts, t_ends, sigmas, packed_info, ray_indices, n_rays,
                                                  prefix_trans)
    return trans, alphas


def render_transmittance_from_alpha(alphas: Tensor, packed_info: Optional[Tensor] = None,
                                    ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None,
                                    prefix_trans: Optional[Tensor] = None) -> Tensor:
    """T_i = prod_{j<i} (1 - alpha_j).

    >>> render_transmittance_from_alpha([.4,.8,.1,.8,.1,.0,.9], ray_indices=[0,0,0,1,1,2,2])
    [1.00, 0.60, 0.12 | 1.00, 0.20 | 1.00, 1.00]
    """
    from .scan import exclusive_prod
    chunks = _ray_chunks(alphas, packed_info, ray_indices, n_rays)
    survive = 1.0 - alphas
    if chunks is None:
        trans = exclusive_prod(survive)
    else:
        trans = exclusive_prod(survive, torch.stack(chunks, dim=-1))
    return trans if prefix_trans is None else trans * prefix_trans


def render_weight_from_alpha(alphas: Tensor, packed_info: Optional[Tensor] = None,
                             ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None,
                             prefix_trans: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """(weights = T * alpha, transmittance)."""
    trans = render_transmittance_from_alpha(alphas, packed_info, ray_indices, n_rays, prefix_trans)
    return trans * alphas, trans


def _visible(trans, alphas, early_stop_eps, alpha_thre):
    keep = trans >= early_stop_eps
    return keep & (alphas >= alpha_thre) if alpha_thre > 0 else keep


@torch.no_grad()
def render_visibility_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor,
                                   packed_info: Optional[Tensor] = None, ray_indices: Optional[Tensor] = None,
                                   n_rays: Optional[int] = None, early_stop_eps: float = 1e-4,
                                   alpha_thre: float = 0.0, prefix_trans: Optional[Tensor] = None) -> Tensor:
    """Boolean mask of the samples worth shading: transmittance still >= early_stop_eps, and (only when
    alpha_thre > 0) alpha >= alpha_thre."""
    chunks = _ray_chunks(t_starts, packed_info, ray_indices, n_rays)
    if _fused(chunks, t_starts, sigmas) and prefix_trans is None and sigmas.dtype == torch.float32:
        mask, _ = _K.render_visibility(chunks[0], chunks[1], sigmas.contiguous(), t_starts.contiguous(),
                                       t_ends.contiguous(), early_stop_eps=early_stop_eps, alpha_thre=alpha_thre,
                                       want_kept=False)
        return mask.bool()
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info, ray_indices, n_rays,
                                                      prefix_trans)
    return _visible(trans, alphas, early_stop_eps, alpha_thre)


@torch.no_grad()
def render_visibility_from_alpha(alphas: Tensor, packed_info: Optional[Tensor] = None,
                                 ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None,
                                 early_stop_eps: float = 1e-4, alpha_thre: float = 0.0,
                                 prefix_trans: Optional[Tensor] = None) -> Tensor:
    chunks = _ray_chunks(alphas, packed_info, ray_indices, n_rays)
    if _fused(chunks, alphas) and prefix_trans is None and alphas.dtype == torch.float32:
        mask, _ = _K.render_visibility(chunks[0], chunks[1], alphas.contiguous(), from_alpha=True,
                                       early_stop_eps=early_stop_eps, alpha_thre=alpha_thre, want_kept=False)
        return mask.bool()
    trans = render_transmittance_from_alpha(alphas, packed_info, ray_indices, n_rays, prefix_trans)
    return _visible(trans, alphas, early_stop_eps, alpha_thre)


# ---------------------------------------------------------------------------------------------------
# per-ray sums
# ---------------------------------------------------------------------------------------------------
def _terms(weights: Tensor, values: Optional[Tensor]) -> Tensor:
    if values is None:
        return weights.unsqueeze(-1)
    if values.dim() != weights.dim() + 1 or values.shape[:-1] != weights.shape:
        raise AssertionError("values must be (..., D) for weights (...)")
    return values * weights.unsqueeze(-1)


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor] = None,
                          ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None) -> Tensor:
    """sum_i w_i v_i per ray -> (n_rays, D); `values=None` sums the weights themselves (D = 1).  Batched
    weights (n_rays, n_samples) are summed along the sample axis."""
    terms = _terms(weights, values)
    if ray_indices is None:
        return terms.sum(dim=-2)
    if n_rays is None:
        raise AssertionError("n_rays must be provided")
    if weights.dim() != 1:
        raise AssertionError("weights must be flattened")
    out = terms.new_zeros((n_rays, terms.shape[-1]))
    return out.index_add_(0, ray_indices, terms)


def accumulate_along_rays_(weights: Tensor, values: Optional[Tensor] = None,
                           ray_indices: Optional[Tensor] = None, outputs: Optional[Tensor] = None) -> None:
    """In-place variant: adds the per-ray sums to `outputs` (n_rays, D)."""
    terms = _terms(weights, values)
    if ray_indices is None:
        outputs.add_(terms.sum(dim=-2))
        return
    if weights.dim() != 1:
        raise AssertionError("weights must be flattened")
    if outputs.dim() != 2 or outputs.shape[-1] != terms.shape[-1]:
        raise AssertionError("outputs must be of shape (n_rays, D)")
    outputs.index_add_(0, ray_indices, terms)
