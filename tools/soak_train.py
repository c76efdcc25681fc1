"""Scratch: 3000 steps of the full-size training loop on the default (threaded) schedule: throughput per 500 steps,
loss scalars, allocator high-water mark — looks for drift, leaks, hangs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd.trainer import TrainConfig, Trainer
cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=3000, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
t0 = time.time()
for step in range(3001):
    s = tr.train_step(step, want_stats=(step % 500 == 0))
    if step % 500 == 0 and s is not None:
        torch.cuda.synchronize()
        print(f"step {step:5d}  {(time.time() - t0) / max(step, 1) * 1e3:6.2f} ms/step avg  psnr {s['psnr']:.2f}  bpp {s['bpp']:.4f}  "
              f"samples {s['n_rendering_samples']}  rays {s['num_rays']}  alloc {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB  "
              f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
print("psnr on 2 test views:", tr.evaluate(2))
