"""Scratch: per-stream GPU busy time and the union over streams for steady-state training steps (kineto trace)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from cnc_amd.trainer import TrainConfig, Trainer
cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(245):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for s in range(245, 249):
        tr.train_step(s, want_stats=False)
    torch.cuda.synchronize()
import json, tempfile
f = tempfile.mktemp(suffix=".json"); prof.export_chrome_trace(f)
ev = [e for e in json.load(open(f))["traceEvents"] if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset")]
by = collections.defaultdict(list)
for e in ev: by[e["args"].get("stream", e.get("tid"))].append((e["ts"], e["ts"] + e["dur"]))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
allv = [x for v in by.values() for x in v]
print("streams:", {k: f"{sum(e - s for s, e in v) / 4e3:.2f} ms/step in {len(v) // 4} kernels" for k, v in by.items()})
print(f"sum of kernel time {sum(e - s for s, e in allv) / 4e3:.2f} ms/step, union {union(allv) / 4e3:.2f} ms/step, span {(max(e for s, e in allv) - min(s for s, e in allv)) / 4e3:.2f}")
if os.environ.get("DUMP"):
    evs = sorted(ev, key=lambda e: e["ts"])
    t0 = evs[0]["ts"]
    # third step only
    span = (max(e["ts"] + e["dur"] for e in evs) - t0) / 4
    for e in evs:
        t = e["ts"] - t0
        if 2 * span <= t < 3 * span:
            print(f"{t - 2 * span:9.0f} {e['dur']:7.0f} s{e['args'].get('stream')} {e['name'][:70]}")
