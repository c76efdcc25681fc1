"""Scratch: where the HOST time of a training step goes (cProfile over 30 steady-state steps)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd.trainer import TrainConfig, Trainer
cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=2000, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
if "--sequential" in sys.argv:       # both passes launched by this thread: the profile sees the entropy pass's host time too
    tr.ctx_thread = False
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
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(70)
    out = s.getvalue()
    print(out[out.index("ncalls"):][:14000])
