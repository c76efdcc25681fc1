"""Scratch: full-size encode / decode wall time (12x3D T=2^19 + 3x4 planes, F=8) after a short training run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd.trainer import TrainConfig, Trainer

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(60):
    tr.train_step(step)
torch.cuda.synchronize(); t0 = time.time()
Pgs, est_MB, coded_MB, prefix = tr.encode()
torch.cuda.synchronize(); t1 = time.time()
tr.decode_into_field(Pgs, prefix)
torch.cuda.synchronize(); t2 = time.time()
print(f"encode {t1-t0:.2f} s  decode {t2-t1:.2f} s  est {est_MB:.3f} MB coded {coded_MB:.3f} MB")
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    tr.encode()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
