"""Volume rendering over flattened ray samples (reference: nerfacc/volrend.py:14-575).

API and numerics follow the reference; the per-ray prefix sums / products go through the HIP
segmented scans when the samples are flattened (`ray_indices` / `packed_info` given) and through
plain torch ops when they are batched (n_rays, n_samples).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
from torch import Tensor

from .pack import pack_info
from .scan import exclusive_prod, exclusive_sum


def rendering(t_starts: Tensor, t_ends: Tensor, ray_indices: Optional[Tensor] = None,
              n_rays: Optional[int] = None, rgb_sigma_fn: Optional[Callable] = None,
              rgb_alpha_fn: Optional[Callable] = None,
              render_bkgd: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor, Dict]:
    """Composite colours / opacities / depths along rays.

    CNC's fork of nerfacc changes the contract of `rgb_sigma_fn`: it returns THREE values
    `(rgbs, sigmas, positions)` and the extras dict also carries `sigmas`, `rgbs`, `positions`
    (reference nerfacc/volrend.py:89,108-115).
    """
    if ray_indices is not None:
        assert t_starts.shape == t_ends.shape == ray_indices.shape, \
            "Since nerfacc 0.5.0, t_starts, t_ends and ray_indices must have the same shape (N,). "
    if rgb_sigma_fn is None and rgb_alpha_fn is None:
        raise ValueError("At least one of `rgb_sigma_fn` and `rgb_alpha_fn` should be specified.")

    if rgb_sigma_fn is not None:
        if t_starts.shape[0] != 0:
            rgbs, sigmas, positions = rgb_sigma_fn(t_starts, t_ends, ray_indices)
        else:
            positions = None
            rgbs = torch.empty((0, 3), device=t_starts.device)
            sigmas = torch.empty((0,), device=t_starts.device)
        assert rgbs.shape[-1] == 3, "rgbs must have 3 channels, got {}".format(rgbs.shape)
        assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(sigmas.shape)
        weights, trans, alphas = render_weight_from_density(t_starts, t_ends, sigmas,
                                                            ray_indices=ray_indices, n_rays=n_rays)
        extras = {"weights": weights, "alphas": alphas, "trans": trans, "sigmas": sigmas,
                  "rgbs": rgbs, "positions": positions}
    else:
        if t_starts.shape[0] != 0:
            rgbs, alphas = rgb_alpha_fn(t_starts, t_ends, ray_indices)
        else:
            rgbs = torch.empty((0, 3), device=t_starts.device)
            alphas = torch.empty((0,), device=t_starts.device)
        assert rgbs.shape[-1] == 3, "rgbs must have 3 channels, got {}".format(rgbs.shape)
        assert alphas.shape == t_starts.shape, "alphas must have shape of (N,)! Got {}".format(alphas.shape)
        weights, trans = render_weight_from_alpha(alphas, ray_indices=ray_indices, n_rays=n_rays)
        extras = {"weights": weights, "trans": trans, "rgbs": rgbs, "alphas": alphas}

    colors = accumulate_along_rays(weights, values=rgbs, ray_indices=ray_indices, n_rays=n_rays)
    opacities = accumulate_along_rays(weights, values=None, ray_indices=ray_indices, n_rays=n_rays)
    depths = accumulate_along_rays(weights, values=(t_starts + t_ends)[..., None] / 2.0,
                                   ray_indices=ray_indices, n_rays=n_rays)
    depths = depths / opacities.clamp_min(torch.finfo(rgbs.dtype).eps)
    if render_bkgd is not None:
        colors = colors + render_bkgd * (1.0 - opacities)
    return colors, opacities, depths, extras


def _packed(packed_info, ray_indices, n_rays):
    if ray_indices is not None and packed_info is None:
        packed_info = pack_info(ray_indices, n_rays)
    return packed_info


def render_transmittance_from_alpha(alphas: Tensor, packed_info: Optional[Tensor] = None,
                                    ray_indices: Optional[Tensor] = None,
                                    n_rays: Optional[int] = None,
                                    prefix_trans: Optional[Tensor] = None) -> Tensor:
    """T_i = prod_{j<i} (1 - alpha_j).
    alphas [0.4,0.8,0.1 | 0.8,0.1 | 0.0,0.9] -> [1.0,0.6,0.12 | 1.0,0.2 | 1.0,1.0]."""
    trans = exclusive_prod(1 - alphas, _packed(packed_info, ray_indices, n_rays))
    if prefix_trans is not None:
        trans *= prefix_trans
    return trans


def render_transmittance_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor,
                                      packed_info: Optional[Tensor] = None,
                                      ray_indices: Optional[Tensor] = None,
                                      n_rays: Optional[int] = None,
                                      prefix_trans: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """T_i = exp(-sum_{j<i} sigma_j * delta_j), alpha_i = 1 - exp(-sigma_i * delta_i)."""
    sigmas_dt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sigmas_dt)
    trans = torch.exp(-exclusive_sum(sigmas_dt, _packed(packed_info, ray_indices, n_rays)))
    if prefix_trans is not None:
        trans *= prefix_trans
    return trans, alphas


def render_weight_from_alpha(alphas: Tensor, packed_info: Optional[Tensor] = None,
                             ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None,
                             prefix_trans: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """w_i = T_i * alpha_i.  Returns (weights, transmittance)."""
    trans = render_transmittance_from_alpha(alphas, packed_info, ray_indices, n_rays, prefix_trans)
    return trans * alphas, trans


def render_weight_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor,
                               packed_info: Optional[Tensor] = None,
                               ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None,
                               prefix_trans: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """w_i = T_i * (1 - exp(-sigma_i delta_i)).  Returns (weights, transmittance, alphas)."""
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info,
                                                      ray_indices, n_rays, prefix_trans)
    return trans * alphas, trans, alphas


@torch.no_grad()
def render_visibility_from_alpha(alphas: Tensor, packed_info: Optional[Tensor] = None,
                                 ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None,
                                 early_stop_eps: float = 1e-4, alpha_thre: float = 0.0,
                                 prefix_trans: Optional[Tensor] = None) -> Tensor:
    """Visible = transmittance >= early_stop_eps (and alpha >= alpha_thre when alpha_thre > 0)."""
    trans = render_transmittance_from_alpha(alphas, packed_info, ray_indices, n_rays, prefix_trans)
    vis = trans >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis


@torch.no_grad()
def render_visibility_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor,
                                   packed_info: Optional[Tensor] = None,
                                   ray_indices: Optional[Tensor] = None,
                                   n_rays: Optional[int] = None, early_stop_eps: float = 1e-4,
                                   alpha_thre: float = 0.0,
                                   prefix_trans: Optional[Tensor] = None) -> Tensor:
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info,
                                                      ray_indices, n_rays, prefix_trans)
    vis = trans >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis


def _weighted(weights, values):
    if values is None:
        return weights[..., None]
    assert values.dim() == weights.dim() + 1
    assert weights.shape == values.shape[:-1]
    return weights[..., None] * values


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor] = None,
                          ray_indices: Optional[Tensor] = None,
                          n_rays: Optional[int] = None) -> Tensor:
    """sum_i w_i * v_i per ray -> (n_rays, D); values=None accumulates the weights (D=1)."""
    src = _weighted(weights, values)
    if ray_indices is None:
        return torch.sum(src, dim=-2)
    assert n_rays is not None, "n_rays must be provided"
    assert weights.dim() == 1, "weights must be flattened"
    outputs = torch.zeros((n_rays, src.shape[-1]), device=src.device, dtype=src.dtype)
    outputs.index_add_(0, ray_indices, src)
    return outputs


def accumulate_along_rays_(weights: Tensor, values: Optional[Tensor] = None,
                           ray_indices: Optional[Tensor] = None,
                           outputs: Optional[Tensor] = None) -> None:
    """In-place accumulate into `outputs` (n_rays, D)."""
    src = _weighted(weights, values)
    if ray_indices is None:
        outputs.add_(src.sum(dim=-2))
        return
    assert weights.dim() == 1, "weights must be flattened"
    assert outputs.dim() == 2 and outputs.shape[-1] == src.shape[-1], \
        "outputs must be of shape (n_rays, D)"
    outputs.index_add_(0, ray_indices, src)
