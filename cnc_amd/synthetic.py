"""Seeded synthetic inputs of BASELINE.md §3 (no dataset needed): the 16-level x 2^19 x F8 grid,
800x800 pinhole rays around the origin, and a ball-shaped occupancy grid."""
from __future__ import annotations

import math

import numpy as np
import torch

# floor(16 * 1.381913^l) + 2 (the reference's +2 ring, train_CNC_nerf_synthetic.py:151)
RES_16L = [18, 24, 32, 44, 60, 82, 113, 155, 214, 296, 408, 563, 778, 1074, 1484, 2049]
# the reference's own composition (train_CNC_nerf_synthetic.py:150-155)
RES_3D_REF = [18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514]
RES_2D_REF = [130, 258, 514, 1026]


def level_offsets(res_list, log2_hashmap_size, num_dim):
    """Row offsets exactly as GridEncoder builds them (ngp.py:197-210)."""
    offs = [0]
    for R in res_list:
        rows = min(2 ** log2_hashmap_size, int(R) ** num_dim)
        offs.append(offs[-1] + int(math.ceil(rows / 8) * 8))
    return np.asarray(offs, np.int32)


def pinhole_rays(height=800, width=800, camera_angle_x=0.6911, radius=4.0, azimuth=0.0,
                 elevation=0.5, device="cpu"):
    """Origins / unit view directions of one image, OpenGL camera looking at the origin
    (ray formula of examples/datasets/nerf_synthetic.py:200-223)."""
    focal = 0.5 * width / math.tan(0.5 * camera_angle_x)
    eye = torch.tensor([radius * math.cos(elevation) * math.cos(azimuth),
                        radius * math.cos(elevation) * math.sin(azimuth),
                        radius * math.sin(elevation)], dtype=torch.float64)
    fwd = -eye / eye.norm()
    up0 = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    right = torch.linalg.cross(fwd, up0)
    right = right / right.norm()
    up = torch.linalg.cross(right, fwd)
    c2w = torch.stack([right, up, -fwd], dim=1).to(torch.float32)     # columns: x, y, z(back)
    ys, xs = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
    x = xs.reshape(-1).to(torch.float32)
    y = ys.reshape(-1).to(torch.float32)
    cam = torch.stack([(x - width / 2 + 0.5) / focal, -(y - height / 2 + 0.5) / focal,
                       -torch.ones_like(x)], dim=-1)
    dirs = (cam[:, None, :] * c2w[None, :, :]).sum(-1)
    viewdirs = dirs / torch.linalg.norm(dirs, dim=-1, keepdim=True)
    origins = eye.to(torch.float32).expand_as(viewdirs).contiguous()
    return origins.to(device), viewdirs.contiguous().to(device)


def ball_binaries(resolution=128, aabb=(-1.5, -1.5, -1.5, 1.5, 1.5, 1.5), radius=1.0, device="cpu"):
    """binaries[1, r, r, r] = ||cell centre|| < radius (BASELINE.md §3 'A-points')."""
    lo, hi = aabb[0], aabb[3]
    c = (torch.arange(resolution, dtype=torch.float32) + 0.5) / resolution * (hi - lo) + lo
    gx, gy, gz = torch.meshgrid(c, c, c, indexing="ij")
    return ((gx * gx + gy * gy + gz * gz) < radius * radius)[None].contiguous().to(device)
