"""Steady-state training step under the bench's conditions (no per-step read-back), for A/B runs of a switch on ONE box:
    python tools/train_ab.py CNC_CTX_PLANE_BATCH=0 CNC_CTX_PLANE_BATCH=1 ...
runs each setting in a fresh process (warm 240 steps, time 128 = 8 refresh periods) twice, alternating, and prints
ms/step."""
import os
import subprocess
import sys
import time

if os.environ.get("_TRAIN_AB_CHILD"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from cnc_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(n_features=8, sample_num=150000, image_size=400, out_dir="/tmp/cnc_bench_bits")
    tr = Trainer(cfg, device=torch.device("cuda:0"))
    for step in range(240):
        tr.train_step(step, want_stats=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(240, 368):
        tr.train_step(step, want_stats=False)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / 128 * 1e3:.3f}")
    sys.exit(0)

settings = sys.argv[1:] or [""]
res = {s: [] for s in settings}
for rep in range(2):
    for s in settings:
        env = dict(os.environ, _TRAIN_AB_CHILD="1")
        for kv in s.split(","):
            if kv:
                k, v = kv.split("=")
                env[k] = v
        out = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        try:
            res[s].append(float(out.stdout.strip().splitlines()[-1]))
        except (ValueError, IndexError):
            print(out.stderr[-2000:])
            raise
for s in settings:
    print(f"{s or '(default)':40s} ms/step: " + "  ".join(f"{v:.2f}" for v in res[s]))
