"""What an occupancy-refresh step (every `step_update`-th) launches on top of an ordinary one: both profiled with the
sequential schedule, kernels grouped by name, refresh minus ordinary, largest first.
    python tools/refresh_cost.py"""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CNC_CTX_THREAD", "0")
import torch
from torch.profiler import ProfilerActivity, profile

from cnc_amd.trainer import TrainConfig, Trainer

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(255):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("at::native::", "")
    return re.sub(r"<.*", "", n)[:56]


def one(step):
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        tr.train_step(step, want_stats=False)
        torch.cuda.synchronize()
    acc = collections.defaultdict(lambda: [0, 0.0])
    host = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type != torch.autograd.DeviceType.CPU:
            continue
        if e.cpu_parent is None:
            host[e.name[:40]][0] += 1
            host[e.name[:40]][1] += e.cpu_time_total
        for k in e.kernels:
            a = acc[short(k.name)]
            a[0] += 1
            a[1] += k.duration
    return acc, host


ordinary, h0 = one(255)
refresh, h1 = one(256)
rows = []
for k in set(ordinary) | set(refresh):
    n0, t0 = ordinary.get(k, (0, 0.0))
    n1, t1 = refresh.get(k, (0, 0.0))
    rows.append((t1 - t0, n1 - n0, k))
rows.sort(reverse=True)
print(f"ordinary step: {sum(a[0] for a in ordinary.values())} launches, {sum(a[1] for a in ordinary.values()) / 1e3:.2f} ms of kernels; "
      f"refresh step: {sum(a[0] for a in refresh.values())} launches, {sum(a[1] for a in refresh.values()) / 1e3:.2f} ms")
print("  extra ms  extra launches  kernel")
for dt, dn, k in rows[:28]:
    print(f"  {dt / 1e3:8.3f}  {dn:6d}  {k}")
print("---- top-level host ops, refresh minus ordinary (ms)")
hr = sorted(((h1.get(k, (0, 0.0))[1] - h0.get(k, (0, 0.0))[1], h1.get(k, (0, 0))[0] - h0.get(k, (0, 0))[0], k) for k in set(h0) | set(h1)), reverse=True)
for dt, dn, k in hr[:16]:
    print(f"  {dt / 1e3:8.3f}  {dn:6d}  {k}")
