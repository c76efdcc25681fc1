"""Scratch: wall time of refresh steps (step % 16 == 0), of the step behind a refresh (the planes' graph is recaptured
there) and of the others, each step synchronised (upper bounds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd.trainer import TrainConfig, Trainer
cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=2000, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(250):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
t = {"refresh": [], "after": [], "plain": []}
for step in range(250, 250 + 160):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train_step(step, want_stats=False)
    torch.cuda.synchronize()
    kind = "refresh" if step % 16 == 0 else ("after" if step % 16 == 1 else "plain")
    t[kind].append((time.perf_counter() - t0) * 1e3)
    if kind == "plain" and t[kind][-1] > 10.0:
        print(f"  slow plain step {step} (step % 16 = {step % 16}): {t[kind][-1]:.2f} ms", flush=True)
for k, v in t.items():
    v = sorted(v); print(f"{k:8s} n={len(v)} median {v[len(v)//2]:.2f} ms  min {v[0]:.2f} max {v[-1]:.2f}")
a = sum(sum(v) for v in t.values()) / 160
pg = tr.planes_graph
print(f"mean over all {a:.2f} ms (synchronised per step); planes graph: {None if pg is None else (pg.captures, pg.replays)}; "
      f"refreshes {tr.context.refresh_stats}")
