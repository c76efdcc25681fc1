"""Scratch: HOST time per named range of the context forward (no profiler, no syncs added)."""
import os, sys, time, collections, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnc_amd.context as C
from cnc_amd.trainer import TrainConfig, Trainer
acc = collections.OrderedDict()
@contextlib.contextmanager
def timed(name):
    t0 = time.perf_counter()
    try:
        yield
    finally:
        a = acc.setdefault(name, [0.0, 0]); a[0] += time.perf_counter() - t0; a[1] += 1
C._range = timed
cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=2000, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
f = tr.context.forward_binary_vxl_mixPg_3D2D
def g(*a, **k):
    with timed("TOTAL context_fwd"):
        return f(*a, **k)
tr.context.forward_binary_vxl_mixPg_3D2D = g
for step in range(250):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize(); acc.clear()
N = 40
for step in range(250, 250 + N):
    if step % 16 == 0: continue
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
n = acc["TOTAL context_fwd"][1]
for k, (t, c) in acc.items():
    print(f"{k:24s} {t / n * 1e6:8.0f} us/step  ({c / n:.1f} calls/step, {t / c * 1e6:6.0f} us each)")
