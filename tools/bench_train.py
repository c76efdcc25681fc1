"""Scratch: full-size training step (BASELINE config 2/3: 12x3D T=2^19 + 3x4x2D T=2^17, F=8, sample_num=150000)
on the procedural scene; prints ms/step and a kernel-time breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd.trainer import TrainConfig, Trainer

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits",
                  fused_features="--unfused" not in sys.argv)
t0 = time.time()
tr = Trainer(cfg, device=torch.device("cuda:0"))
torch.cuda.synchronize()
print(f"setup {time.time()-t0:.1f}s mem {torch.cuda.memory_allocated()/2**30:.1f} GiB")
for step in range(300):
    if step == 100:
        torch.cuda.synchronize(); t0 = time.time()
    s = tr.train_step(step)
torch.cuda.synchronize()
dt = (time.time() - t0) / 200
print(f"train step: {dt*1e3:.1f} ms  last: {s}")
if "--no-profile" in sys.argv:
    sys.exit(0)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for step in range(301, 305):
        tr.train_step(step)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=80))
