"""Regenerate tests/golden/render.npz (THIS container only: imports the reference's Python from
/root/reference; nothing of its source is stored, only seeded inputs and what its functions returned).

    python tests/golden/make_golden_render.py

 * nerfacc.volrend.{render_weight_from_density, accumulate_along_rays} — the batched (n_rays, n_samples)
   branches run on CPU — and the tail of `rendering` (depth / opacity, background blend, volrend.py:116-140)
   on seeded rays of equal length: pins oracle.render_weight_from_density / oracle.composite (and through
   them the fused HIP kernels) to the reference's arithmetic.
 * nerfacc.estimators.occ_grid.OccGridEstimator.mark_invisible_cells on seeded cameras: pins the rewritten
   method of cnc_amd.nerfacc.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)


def main():
    import nerfacc
    from nerfacc.volrend import accumulate_along_rays, render_weight_from_density
    out = {}
    g = torch.Generator().manual_seed(2024)
    R, M = 37, 45
    t_starts = torch.cumsum(torch.rand(R, M, generator=g) * 0.02 + 0.001, dim=1) + 2.0
    t_ends = t_starts + torch.rand(R, M, generator=g) * 0.02 + 0.001
    sigmas = torch.rand(R, M, generator=g) ** 4 * 60.0
    sigmas[3] = 0.0                       # an empty ray
    sigmas[5, 10:] = 1e4                  # saturates
    rgbs = torch.rand(R, M, 3, generator=g)
    bkgd = torch.tensor([0.25, 0.5, 1.0])
    for name, prefix in (("plain", None), ("prefix", torch.rand(R, 1, generator=g).expand(R, M).contiguous())):
        w, tr, al = render_weight_from_density(t_starts, t_ends, sigmas, prefix_trans=None if prefix is None else prefix.clone())
        colors = accumulate_along_rays(w, values=rgbs)
        opac = accumulate_along_rays(w, values=None)
        depth_sum = accumulate_along_rays(w, values=(t_starts + t_ends)[..., None] / 2.0)
        depth = depth_sum / opac.clamp_min(torch.finfo(rgbs.dtype).eps)
        colors_bk = colors + bkgd * (1.0 - opac)
        for k, v in (("weights", w), ("trans", tr), ("alphas", al), ("colors", colors), ("opacity", opac),
                     ("depth_sum", depth_sum), ("depth", depth), ("colors_bkgd", colors_bk)):
            out[f"{name}_{k}"] = v.numpy()
        if prefix is not None:
            out["prefix"] = prefix.numpy()
    out.update(t_starts=t_starts.numpy(), t_ends=t_ends.numpy(), sigmas=sigmas.numpy(), rgbs=rgbs.numpy(),
               bkgd=bkgd.numpy())

    # mark_invisible_cells: 5 cameras on a ring looking at the origin, 2 grid levels
    est = nerfacc.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=12, levels=2)
    n_cam, W, H = 5, 40, 30
    ang = torch.linspace(0, 2 * np.pi, n_cam + 1)[:-1]
    eye = torch.stack([2.5 * torch.cos(ang), 2.5 * torch.sin(ang), torch.full_like(ang, 0.4)], -1)
    fwd = -eye / eye.norm(dim=-1, keepdim=True)
    up0 = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.linalg.cross(fwd, up0); right = right / right.norm(dim=-1, keepdim=True)
    down = torch.linalg.cross(fwd, right)
    c2w = torch.cat([torch.stack([right, down, fwd], dim=-1), eye[..., None]], dim=-1)      # OpenCV: z forward
    K = torch.tensor([[[35.0, 0, W / 2], [0, 35.0, H / 2], [0, 0, 1]]])
    est.mark_invisible_cells(K, c2w, W, H, near_plane=1.2)
    out.update(mic_K=K.numpy(), mic_c2w=c2w.numpy(), mic_WH=np.array([W, H]), mic_near=np.float64(1.2),
               mic_occs=est.occs.numpy())
    print("invisible cells:", int((est.occs < 0).sum()), "of", est.occs.numel())
    np.savez_compressed(os.path.join(HERE, "render.npz"), **out)


if __name__ == "__main__":
    main()
