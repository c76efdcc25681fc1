"""Scratch: where a full-size training step spends its time (sync-bracketed phases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from cnc_amd.trainer import TrainConfig, Trainer
from cnc_amd.render import render_image_with_occgrid

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(100):
    tr.train_step(step)
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); t1 = time.perf_counter(); acc[name] = acc.get(name, 0) + (t1 - t0); return t1
n = 0
for step in range(101, 161):
    if step % cfg.step_update == 0:
        tr.train_step(step); continue
    c = tr.cfg
    torch.cuda.synchronize(); t = time.perf_counter()
    data = tr.dataset.fetch(); rays, pixels, bkgd = data["rays"], data["pixels"], data["color_bkgd"]
    t = tick("fetch", t)
    rgb, a_, d_, ns, extra = render_image_with_occgrid(tr.field, tr.estimator, rays, near_plane=c.near_plane,
        render_step_size=c.render_step_size, render_bkgd=bkgd, cone_angle=c.cone_angle, alpha_thre=c.alpha_thre, return_extra=True)
    t = tick("render fwd", t)
    mse = F.mse_loss(rgb, pixels)
    e = tr.field.mlp_base
    bpp, mb = tr.context.forward_binary_vxl_mixPg_3D2D(e.encoding_xyz, e.encoding_xy, e.encoding_xz, e.encoding_yz,
                                                       tr.estimator.binaries, sample_num=None, step=step)
    loss = mse + c.lmbda * bpp
    t = tick("context fwd", t)
    tr.opt.zero_grad(set_to_none=True); tr.opt2.zero_grad(set_to_none=True)
    (loss * tr.loss_scale).backward()
    t = tick("backward", t)
    tr.opt.step(); tr.opt2.step(); tr.sched.step(); tr.sched2.step()
    t = tick("optimizer", t)
    n += 1
tot = sum(acc.values())
for k, v in acc.items():
    print(f"{k:12s} {v / n * 1e3:7.2f} ms")
print(f"total        {tot / n * 1e3:7.2f} ms over {n} steps; samples {ns}, rays {len(pixels)}")
