"""Scratch: where the GPU sits idle inside a steady-state training step.  Gaps between consecutive kernels are
attributed to the named host range (ctx/..., PH/...) or top-level op that was running when the gap began."""
import os, sys, bisect, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from cnc_amd.trainer import TrainConfig, Trainer

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(245):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
n = 4
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for s in range(245, 245 + n):
        with torch.profiler.record_function(f"STEP"):
            tr.train_step(s, want_stats=False)
    torch.cuda.synchronize()
evs = prof.events()
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
gpu = sorted([e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA and e.time_range.end > e.time_range.start],
             key=lambda e: e.time_range.start)
steps = sorted([e for e in cpu if e.name == "STEP"], key=lambda e: e.time_range.start)
# host ranges to attribute to: the innermost event among (children of STEP at depth <= 3)
def depth(e):
    d = 0
    while e.cpu_parent is not None:
        e = e.cpu_parent; d += 1
    return d
named = sorted([e for e in cpu if depth(e) in (1, 2) and e.name != "STEP"], key=lambda e: e.time_range.start)
starts = [e.time_range.start for e in named]
def owner(t):
    i = bisect.bisect_right(starts, t) - 1
    best = None
    while i >= 0 and i > bisect.bisect_right(starts, t) - 40:
        e = named[i]
        if e.time_range.start <= t <= e.time_range.end:
            if best is None or depth(e) > depth(best): best = e
        i -= 1
    return best.name[:48] if best else "(between ops)"
idle = collections.Counter(); busy = 0.0
t0, t1 = steps[1].time_range.start, steps[-1].time_range.end      # skip the first profiled step
last_end = None
for k in gpu:
    if k.time_range.end < t0 or k.time_range.start > t1: continue
    if last_end is not None and k.time_range.start > last_end:
        gap = k.time_range.start - last_end
        if gap > 2: idle[owner(last_end)] += gap
    busy += k.time_range.end - max(k.time_range.start, last_end or 0) if (last_end is None or k.time_range.end > last_end) else 0
    last_end = max(last_end or 0, k.time_range.end)
m = n - 1
print(f"per step: wall {(t1 - t0) / m / 1e3:.2f} ms, GPU busy {busy / m / 1e3:.2f} ms, GPU idle {sum(idle.values()) / m / 1e3:.2f} ms")
for name, v in idle.most_common(28):
    print(f"  idle {v / m:8.0f} us  while host in  {name}")
