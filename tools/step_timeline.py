"""Scratch: host clock vs GPU clock at the phase boundaries of a training step, WITHOUT a profiler: at each
boundary the host time is taken and a HIP event recorded; lag = GPU time - host time at that boundary (how far the
GPU runs behind the host: > 0 = the GPU is the bottleneck there, ~0 = the GPU waits for launches)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnc_amd.trainer as T
from cnc_amd.trainer import TrainConfig, Trainer

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=2000, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record()
    marks.append((name, time.perf_counter(), e))
def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def g(*a, **k):
        r = f(*a, **k); mark(name); return r
    setattr(obj, attr, g)
wrap(tr.dataset, "fetch", "fetch")
wrap(T, "render_image_with_occgrid", "render_fwd")
wrap(tr.context, "forward_binary_vxl_mixPg_3D2D", "context_fwd")
_bw = torch.Tensor.backward
def bw(self, *a, **k):
    r = _bw(self, *a, **k); mark("backward"); return r
torch.Tensor.backward = bw
wrap(tr.opt2, "step", "optimizer")
step = 0
for _ in range(250):
    tr.train_step(step, want_stats=False); step += 1
torch.cuda.synchronize()
marks.clear()
N = 40
rows = []
for _ in range(N):
    mark("start"); tr.train_step(step, want_stats=False); step += 1
torch.cuda.synchronize()
# per step: host and gpu time of each boundary relative to the step's own "start" mark
agg = collections.OrderedDict()
i = 0
first_h, first_e = marks[0][1], marks[0][2]
while i < len(marks):
    assert marks[i][0] == "start"
    h0, e0 = marks[i][1], marks[i][2]
    j = i + 1
    while j < len(marks) and marks[j][0] != "start":
        name, h, e = marks[j]
        a = agg.setdefault(name, [0.0, 0.0, 0])
        a[0] += (h - h0) * 1e3; a[1] += e0.elapsed_time(e); a[2] += 1
        j += 1
    a = agg.setdefault("lag at step start", [0.0, 0.0, 0])
    a[0] += (h0 - first_h) * 1e3; a[1] += first_e.elapsed_time(e0); a[2] += 1
    i = j
print(f"{'boundary':20s} host ms   gpu ms   (since the step's start mark; averages over {N} steps)")
for name, (h, g, c) in agg.items():
    if name == "lag at step start":
        print(f"GPU behind host at the start of a step: {(g - h) / c:.2f} ms on average")
    else:
        print(f"{name:20s} {h / c:7.2f} {g / c:8.2f}")
wall = (marks[-1][1] - first_h) / (N - 1) * 1e3 if N > 1 else 0
print(f"wall per step (host, start to start) {wall:.2f} ms")
