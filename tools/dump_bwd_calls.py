"""Scratch: the inputs of every `grid_encode_backward` call of ONE steady-state training step (full model, F = 8), written
to gpurun_out/bwd_calls.npz for offline analysis (distinct target rows per block, runs, per-level point counts).
Run with CNC_PLANES_GRAPH=0 (the planes' calls are replayed from a graph otherwise and never reach Python)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CNC_PLANES_GRAPH", "0")
import numpy as np
import torch
from cnc_amd.trainer import TrainConfig, Trainer
from cnc_amd.backends import gridencoder_backend as be

cfg = TrainConfig(n_features=8, sample_num=150000, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
warm = int(os.environ.get("WARM", "250"))
for step in range(warm):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()

calls = []
orig = be.grid_encode_backward


def spy(grad, inputs, embeddings, offsets_list, resolutions_list, grad_embeddings, N, num_dim, n_features, n_levels,
        max_level, Rb, dy_dx=None, grad_inputs=None, binary_vxl=None, min_level_id=None, **kw):
    torch.cuda.synchronize()
    ld, col = kw.get("grad_ld", 0), kw.get("grad_col", 0)
    if ld:
        g = grad.view(-1, ld)[:N, col:col + n_levels * n_features].reshape(N, n_levels, n_features)
    else:
        g = grad.view(n_levels, N, n_features).permute(1, 0, 2)
    nz = (g != 0).any(-1).to(torch.uint8).cpu().numpy()
    gabs = g.abs().amax(-1).to(torch.float16).cpu().numpy()
    calls.append(dict(N=N, D=num_dim, F=n_features, L=n_levels, Rb=Rb, inputs=inputs[:N].detach().cpu().numpy(),
                      offsets=offsets_list.cpu().numpy(), resolutions=resolutions_list.cpu().numpy(),
                      vxl=None if binary_vxl is None else np.packbits(binary_vxl.cpu().numpy().astype(np.uint8)),
                      vxl_shape=None if binary_vxl is None else np.array(binary_vxl.shape),
                      mli=None if min_level_id is None else min_level_id.cpu().numpy(), nz=nz, gabs=gabs,
                      ste=bool(kw.get("ste_binary", False)), binned=kw.get("binned"),
                      stream=torch.cuda.current_stream().cuda_stream))
    return orig(grad, inputs, embeddings, offsets_list, resolutions_list, grad_embeddings, N, num_dim, n_features, n_levels,
                max_level, Rb, dy_dx, grad_inputs, binary_vxl, min_level_id, **kw)


be.grid_encode_backward = spy
# the mirrors import the backend module, not the function: patching the module attribute reaches them
step = warm
while step % cfg.step_update == 0 or (step - 1) % cfg.step_update == 0:
    tr.train_step(step, want_stats=False); step += 1; calls.clear()
tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
be.grid_encode_backward = orig
out = {}
for i, c in enumerate(calls):
    print(i, {k: (v.shape if isinstance(v, np.ndarray) else v) for k, v in c.items()})
    for k, v in c.items():
        if v is not None and k != "binned" and k != "stream":
            out[f"c{i}_{k}"] = np.asarray(v)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/bwd_calls.npz", **out)
print("saved", os.path.getsize("gpurun_out/bwd_calls.npz") / 2**20, "MiB")
