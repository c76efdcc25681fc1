import sys, os, cProfile, pstats
sys.path.insert(0, os.getcwd())
os.environ.setdefault("CNC_CTX_THREAD", "0")
import torch
from cnc_amd.trainer import TrainConfig, Trainer
cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(241):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
import time
t0 = time.perf_counter()
pr.enable()
n = 0
for step in range(241, 256):
    tr.train_step(step, want_stats=False); n += 1
pr.disable()
torch.cuda.synchronize()
print("ms/step under profile", (time.perf_counter() - t0) / n * 1e3)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(60)
