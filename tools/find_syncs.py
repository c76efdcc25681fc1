"""Scratch: list the source lines of a training step that force a device->host sync."""
import os, sys, warnings, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd.trainer import TrainConfig, Trainer

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(1, 20):
    tr.train_step(step)
counts = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message):
        st = traceback.extract_stack()
        mine = [f for f in st if "/cnc_amd/" in f.filename]
        key = f"{os.path.basename(mine[-1].filename)}:{mine[-1].lineno} {mine[-1].line}" if mine else f"{filename}:{lineno}"
        counts[key] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
for step in (21, 22):
    tr.train_step(step)
torch.cuda.set_sync_debug_mode("default")
for k, v in counts.most_common(60):
    print(v // 2, k[:150])
print("total per step", sum(counts.values()) // 2)
