"""Full-size training step (configs[1]/[2]: F=8, 12x3-D + 3x4 planes, sample_num=150000) under the torch
profiler, refresh-free steps only: GPU time and kernel launches per step, the top kernels, and the time per
`ctx/...` range.  `python tools/profile_train_step.py [n_steps]`"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from cnc_amd.trainer import TrainConfig, Trainer

n_prof = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(161):
    tr.train_step(step)
torch.cuda.synchronize()
t0 = time.perf_counter()
for step in range(161, 161 + 15):          # 161..175: no refresh (176 = 11*16 is the next one)
    s = tr.train_step(step)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 15
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for step in range(177, 177 + n_prof):
        tr.train_step(step)
    torch.cuda.synchronize()
ev = prof.key_averages()
kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
rng = [e for e in kern if e.key.startswith("ctx/") or e.key.startswith("Optimizer") or "Backward" in e.key or e.key.startswith("_")]
real = [e for e in kern if e not in rng and not e.key.startswith("aten::")]
tot = sum(e.self_device_time_total for e in real) / 1e3 / n_prof
cnt = sum(e.count for e in real) / n_prof
print(f"wall {wall*1e3:.2f} ms/step (refresh-free), {s['n_rendering_samples']} samples, {s['num_rays']} rays; "
      f"GPU kernel time {tot:.2f} ms/step in {cnt:.0f} launches/step")
cnc = sum(e.self_device_time_total for e in real if "cnc::" in e.key) / 1e3 / n_prof
gemm = sum(e.self_device_time_total for e in real if e.key.startswith("Cijk") or "gemm" in e.key.lower()) / 1e3 / n_prof
print(f"  cnc:: kernels {cnc:.2f} ms ({100*cnc/tot:.0f} %), hipBLASLt GEMMs {gemm:.2f} ms ({100*gemm/tot:.0f} %), "
      f"ATen/rocprim rest {tot-cnc-gemm:.2f} ms")
print("top kernels (ms/step, launches/step):")
for e in sorted(real, key=lambda e: -e.self_device_time_total)[:45]:
    print(f"  {e.self_device_time_total/1e3/n_prof:7.3f} {e.count/n_prof:7.1f}  {e.key[:150]}")
print("ranges (ms/step of GPU time inside):")
for e in sorted(rng, key=lambda e: -e.device_time_total)[:25]:
    print(f"  {e.device_time_total/1e3/n_prof:7.3f} {e.count/n_prof:6.1f}  {e.key[:90]}")
