"""Kernel launches and GPU time per PHASE of a refresh-free full-size training step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from torch.profiler import ProfilerActivity, profile, record_function
from cnc_amd.trainer import TrainConfig, Trainer
from cnc_amd.render import render_image_with_occgrid

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(161):
    tr.train_step(step)
c = tr.cfg
def one(step):
    with record_function("PH/fetch"):
        data = tr.dataset.fetch(); rays, pixels, bkgd = data["rays"], data["pixels"], data["color_bkgd"]
    with record_function("PH/render_fwd"):
        rgb, a_, d_, ns, extra = render_image_with_occgrid(tr.field, tr.estimator, rays, near_plane=c.near_plane,
            render_step_size=c.render_step_size, render_bkgd=bkgd, cone_angle=c.cone_angle, alpha_thre=c.alpha_thre, return_extra=True)
        mse = F.mse_loss(rgb, pixels)
    with record_function("PH/context_fwd"):
        e = tr.field.mlp_base
        bpp, mb = tr.context.forward_binary_vxl_mixPg_3D2D(e.encoding_xyz, e.encoding_xy, e.encoding_xz, e.encoding_yz,
                                                           tr.estimator.binaries, sample_num=None, step=step)
        loss = mse + c.lmbda * bpp
    tr.opt.zero_grad(set_to_none=True); tr.opt2.zero_grad(set_to_none=True)
    with record_function("PH/backward_render"):
        (mse * tr.loss_scale).backward()
    with record_function("PH/backward_context"):
        (c.lmbda * bpp * tr.loss_scale).backward()
    with record_function("PH/optimizer"):
        tr.opt.step(); tr.opt2.step(); tr.sched.step(); tr.sched2.step()
for s in range(161, 165): one(s)
torch.cuda.synchronize()
n = 6
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for s in range(165, 165 + n): one(s)
    torch.cuda.synchronize()
evs = prof.events()
phases = [e for e in evs if e.name.startswith("PH/") and e.device_type == torch.autograd.DeviceType.CPU]
kerns = [e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA and not e.name.startswith("PH/")
         and not e.name.startswith("ctx/") and not e.name.startswith("aten::") and "Backward" not in e.name[:40]
         and not e.name.startswith("_") and not e.name.startswith("Optimizer")]
# attribute kernels to the phase whose CPU range launched them, through the correlation of launch times
import bisect
agg = {}
for p in phases:
    agg.setdefault(p.name, [0, 0.0, 0.0])
    agg[p.name][2] += p.cpu_time_total
# kernel launch (CPU side) time ranges: use the linked cpu op if available, else fall back to time containment of
# the kernel's own start (the GPU runs ~in order behind the CPU)
ph_sorted = sorted(phases, key=lambda e: e.time_range.start)
starts = [e.time_range.start for e in ph_sorted]
lag = 0
for k in kerns:
    t = k.time_range.start
    i = bisect.bisect_right(starts, t) - 1
    if i < 0: continue
    name = ph_sorted[i].name
    agg[name][0] += 1
    agg[name][1] += k.device_time_total if hasattr(k, "device_time_total") else k.cuda_time_total
print("phase                launches/step   GPU ms/step   CPU ms/step   (attribution by kernel start time: approximate)")
for name, (cnt, gpu, cpu) in agg.items():
    print(f"{name:22s} {cnt/n:10.0f} {gpu/1e3/n:12.2f} {cpu/1e3/n:12.2f}")
