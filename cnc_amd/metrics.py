"""Image metrics of the results line.  PSNR as the reference computes it (train_CNC_nerf_synthetic.py:416-418);
SSIM in the standard 11x11 Gaussian (sigma 1.5) formulation of the reference's vendored `examples/pytorch_ssim.py`
(out of scope per SURVEY §2 row 15: restated from the published definition, parity unpinned);
LPIPS needs pretrained weights that cannot be had offline and is reported as NaN."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def psnr(rgb: torch.Tensor, target: torch.Tensor) -> float:
    mse = F.mse_loss(rgb, target)
    return float(-10.0 * torch.log(mse) / math.log(10.0))


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, sigma: float = 1.5) -> float:
    """Mean structural similarity of two (1, C, H, W) images in [0, 1]."""
    c = img1.shape[1]
    ax = torch.arange(window_size, dtype=img1.dtype, device=img1.device) - window_size // 2
    g = torch.exp(-(ax * ax) / (2 * sigma * sigma))
    g = (g / g.sum())[:, None]
    win = (g @ g.T)[None, None].expand(c, 1, window_size, window_size).contiguous()
    blur = lambda x: F.conv2d(x, win, padding=window_size // 2, groups=c)
    mu1, mu2 = blur(img1), blur(img2)
    s11, s22, s12 = blur(img1 * img1) - mu1 * mu1, blur(img2 * img2) - mu2 * mu2, blur(img1 * img2) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return float((((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))).mean())
