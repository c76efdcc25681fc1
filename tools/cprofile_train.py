"""Scratch: where the HOST time of a training step goes (cProfile over 30 steady-state steps)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd.trainer import TrainConfig, Trainer
cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=2000, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(245):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for step in range(245, 275):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    out = s.getvalue()
    print(out[out.index("ncalls"):][:9000])
