"""Scratch: full-size evaluation render (800x800, F=8 field) after a short training run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd.trainer import TrainConfig, Trainer
from cnc_amd.render import render_image_with_occgrid_test

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=800, out_dir="/tmp/bits", lmbda=0)
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(200):
    tr.train_step(step)
tr.field.eval(); tr.estimator.eval()
d = tr.dataset.view(0)
c = cfg
with torch.no_grad():
    for i in range(2):
        render_image_with_occgrid_test(1024, tr.field, tr.estimator, d["rays"], near_plane=c.near_plane,
                                       render_step_size=c.render_step_size, render_bkgd=d["color_bkgd"], cone_angle=c.cone_angle)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(3):
        rgb, acc, depth, n = render_image_with_occgrid_test(1024, tr.field, tr.estimator, d["rays"], near_plane=c.near_plane,
                                       render_step_size=c.render_step_size, render_bkgd=d["color_bkgd"], cone_angle=c.cone_angle)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
print(f"eval render 800x800: {dt*1e3:.1f} ms/image, {n} samples, {640000/dt/1e6:.2f} M rays/s, {n/dt/1e6:.1f} M samples/s")
if "--profile" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof, torch.no_grad():
        render_image_with_occgrid_test(1024, tr.field, tr.estimator, d["rays"], near_plane=c.near_plane,
                                       render_step_size=c.render_step_size, render_bkgd=d["color_bkgd"], cone_angle=c.cone_angle)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=70))
