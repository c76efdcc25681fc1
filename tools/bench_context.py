"""Scratch: full-size context pass (BASELINE config C: sample_num=150000, 12x3D T=2^19 + 3x4x2D T=2^17, F=8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd import synthetic
from cnc_amd.context import CNC_context_models
from cnc_amd.gridencoder import GridEncoder

dev = torch.device("cuda:0")
F = 8
torch.manual_seed(42)
t0 = time.time()
m = CNC_context_models(num_dim=3, resolutions_list=synthetic.RES_3D_REF, resolutions_list_2D=synthetic.RES_2D_REF,
                       log2_hashmap_size=19, log2_hashmap_size_2D=17, n_features=F, sample_num=150000,
                       max_context_layer_num=3, ste_binary=True, Pg_level=12, Pg_level_2D=4, Rb=128, step_update=16,
                       skip_levels_3D=[0, 1, 2], skip_levels_2D=[0], device=dev)
torch.cuda.synchronize()
print(f"context tables built in {time.time()-t0:.1f}s; mem {torch.cuda.memory_allocated()/2**30:.2f} GiB; "
      f"hashparams/level {m.hashparams_num_levels.tolist()} samples/level {m.sample_num_levels.tolist()}")
encs = [GridEncoder(3, F, synthetic.RES_3D_REF, 19, ste_binary=True).to(dev)] + \
       [GridEncoder(2, F, synthetic.RES_2D_REF, 17, ste_binary=True).to(dev) for _ in range(3)]
binaries = synthetic.ball_binaries(128, radius=1.0, device=dev)
params = [p for e in encs for p in e.parameters()] + list(m.parameters())
for step in range(20):
    if step == 4:
        torch.cuda.synchronize(); t0 = time.time()
    for p in params:
        p.grad = None
    bpp, mb = m.forward_binary_vxl_mixPg_3D2D(*encs, binaries, step=step)
    bpp.backward()
torch.cuda.synchronize()
print(f"context pass fwd+bwd: {(time.time()-t0)/16*1e3:.1f} ms/step  bpp={bpp.item():.4f} est {mb:.3f} MB  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for step in range(17, 19):
        for p in params:
            p.grad = None
        bpp, mb = m.forward_binary_vxl_mixPg_3D2D(*encs, binaries, step=step)
        bpp.backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=70))
