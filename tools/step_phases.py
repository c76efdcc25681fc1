"""Where a training step's wall time goes, both host threads: host clock and a HIP event at every phase boundary of
`Trainer.train_step` (no profiler).  Per boundary: host ms since the step began, GPU ms (the event's completion) since
the step's first event.  host << gpu: the GPU is behind (GPU-bound there); gpu ~ host: the GPU waits for launches."""
import collections
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cnc_amd.trainer as T
from cnc_amd.trainer import TrainConfig, Trainer

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=2000, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
marks = []
lock = threading.Lock()


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()                      # on the calling thread's current stream
    with lock:
        marks.append((name, threading.current_thread().name[:3], time.perf_counter(), e))


def wrap(obj, attr, name, before=False):
    f = getattr(obj, attr)

    def g(*a, **k):
        if before:
            mark(name + ":begin")
        r = f(*a, **k)
        mark(name)
        return r
    setattr(obj, attr, g)


wrap(tr.dataset, "fetch", "fetch")
wrap(tr.estimator, "update_every_n_steps", "occ_update")
wrap(T, "render_image_with_occgrid", "render_fwd")
wrap(tr.context, "forward_binary_vxl_mixPg_3D2D", "ctx_fwd", before=True)
wrap(tr, "_context_pass", "ctx_pass_done", before=True)
_bw = torch.Tensor.backward


def bw(self, *a, **k):
    r = _bw(self, *a, **k)
    mark("backward_returned")
    return r


torch.Tensor.backward = bw
wrap(tr.opt, "step", "opt_field")
wrap(tr.opt2, "step", "opt_ctx")
step = 0
for _ in range(245):
    tr.train_step(step, want_stats=False)
    step += 1
torch.cuda.synchronize()
marks.clear()
N = 48
for _ in range(N):
    if step % cfg.step_update == 0:          # keep refresh steps out of the averages
        tr.train_step(step, want_stats=False)
        step += 1
        torch.cuda.synchronize()
        marks.clear() if not any(m[0] == "start" for m in marks) else None
        continue
    mark("start")
    tr.train_step(step, want_stats=False)
    step += 1
    mark("end")
torch.cuda.synchronize()
agg = collections.OrderedDict()
cur = None
n_steps = 0
for name, th, h, e in sorted(marks, key=lambda m: m[2]):
    if name == "start":
        cur = (h, e)
        n_steps += 1
        continue
    if cur is None:
        continue
    a = agg.setdefault((name, th), [0.0, 0.0, 0])
    a[0] += (h - cur[0]) * 1e3
    a[1] += cur[1].elapsed_time(e)
    a[2] += 1
    if name == "end":
        cur = None
print(f"{'boundary':28s} thr  host ms   gpu ms    (since the step's start; mean over {n_steps} non-refresh steps)")
for (name, th), (h, g, c) in sorted(agg.items(), key=lambda t: t[1][0] / max(t[1][2], 1)):
    print(f"{name:28s} {th}  {h / c:7.2f} {g / c:8.2f}   x{c / max(n_steps, 1):.1f}")
