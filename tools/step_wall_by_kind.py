"""Scratch: wall time of refresh steps (step % 16 == 0) against the others (each step synchronised: upper bounds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd.trainer import TrainConfig, Trainer
cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=2000, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(250):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
t = {True: [], False: []}
for step in range(250, 250 + 96):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train_step(step, want_stats=False)
    torch.cuda.synchronize()
    t[step % 16 == 0].append((time.perf_counter() - t0) * 1e3)
for k in (False, True):
    v = sorted(t[k]); print("refresh step" if k else "plain step  ", f"n={len(v)} median {v[len(v)//2]:.2f} ms  min {v[0]:.2f} max {v[-1]:.2f}")
a = (sum(t[True]) + sum(t[False])) / 96
print(f"mean over all {a:.2f} ms (synchronised per step)")
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
tr.train_step(352, want_stats=False); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); o = s.getvalue(); print(o[o.index("ncalls"):][:4000])
