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
    if torch.cuda.is_current_stream_capturing():       # the planes' graph being recorded: nothing runs, nothing to time
        return
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
wrap(tr.estimator, "sampling", "  sampling", before=True)
wrap(tr.estimator, "_march", "    march")
import cnc_amd.render as _R
_cd = _R._FieldOnRays.colour_and_density


def _cd_marked(self, *a, **k):
    r = _cd(self, *a, **k)
    mark("  field(colour, density)")
    return r


_R._FieldOnRays.colour_and_density = _cd_marked
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
REFRESH = "--refresh" in sys.argv           # average the refresh steps (every `step_update`-th) instead of the others
wrap(tr.context, "get_idx_coords2", "ctx/idx_coords2")
wrap(tr.context, "fetch_2D_batches", "ctx/fetch_2D")
wrap(tr.context, "_sorted_slots_2D", "ctx/sorted_slots_2D")
wrap(tr.context, "get_pn_embed_frac_planes", "ctx/pn_frac")
import cnc_amd.context as _C
wrap(_C._backend.VotePlan, "from_occupancy", "ctx/vote_plan")
wrap(tr.context, "_project", "ctx/project")
N = 48 if not REFRESH else 16 * 12
for _ in range(N):
    if (step % cfg.step_update == 0) != REFRESH:          # keep the other kind of step out of the averages
        tr.train_step(step, want_stats=False)
        step += 1
        torch.cuda.synchronize()
        continue
    torch.cuda.synchronize()
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
        seen = collections.Counter()
        continue
    if cur is None:
        continue
    seen[name, th] += 1                     # a boundary crossed several times in a step: one row per crossing
    if seen[name, th] > 1 or name in ("ctx/fetch_2D", "ctx/sorted_slots_2D"):
        name = f"{name} #{seen[name, th]}"
    a = agg.setdefault((name, th), [0.0, 0.0, 0])
    a[0] += (h - cur[0]) * 1e3
    a[1] += cur[1].elapsed_time(e)
    a[2] += 1
    if name.startswith("end"):
        cur = None
print(f"{'boundary':28s} thr  host ms   gpu ms    (since the step's start; mean over {n_steps} {'refresh' if REFRESH else 'non-refresh'} steps)")
for (name, th), (h, g, c) in sorted(agg.items(), key=lambda t: t[1][0] / max(t[1][2], 1)):
    print(f"{name:28s} {th}  {h / c:7.2f} {g / c:8.2f}   x{c / max(n_steps, 1):.1f}")
