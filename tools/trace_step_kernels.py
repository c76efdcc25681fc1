"""Scratch: the kernels of ONE steady-state training step in launch order (name, us, launching aten/custom op)."""
import os, sys, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from cnc_amd.trainer import TrainConfig, Trainer

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(245):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=False) as prof:
    tr.train_step(245, want_stats=False)
    torch.cuda.synchronize()
evs = prof.events()
cpu = sorted([e for e in evs if e.device_type == torch.autograd.DeviceType.CPU], key=lambda e: e.time_range.start)
gpu = sorted([e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
# top-level cpu ops (no parent) in order, with the kernels they launched (via e.kernels)
def short(n):
    n = re.sub(r"\(.*", "", n); n = n.replace("void ", "").replace("at::native::", "")
    return n[:90]
t0 = cpu[0].time_range.start
tops = [e for e in cpu if e.cpu_parent is None]
for e in tops:
    ks = []
    def walk(x):
        ks.extend(x.kernels)
        for c in x.cpu_children: walk(c)
    walk(e)
    print(f"{(e.time_range.start - t0):9.0f}us cpu {e.cpu_time_total:7.0f}us  {e.name[:60]:60s} kernels={len(ks)} gpu={sum(k.duration for k in ks):7.0f}us")
    for k in ks:
        print(f"                           . {short(k.name)}  {k.duration:.0f}us")
print("n cpu tops", len(tops), "n gpu events", len(gpu))
