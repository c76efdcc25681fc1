"""Marched vs surviving samples of the sampler over a full-size training run (the ratio that decides whether the
front-to-back density evaluation pays, docs/engineering_log.md §10)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd.trainer import TrainConfig, Trainer
dev = torch.device("cuda:0")
cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=dev)
for step in range(300):
    s = tr.train_step(step, want_stats=False)
    if step % 20 == 0:
        ratio = getattr(tr.estimator, "_marched_per_kept", 1.0)
        print(step, "rays", s["num_rays"], "kept", s["n_rendering_samples"], "marched / kept", round(ratio, 2))
