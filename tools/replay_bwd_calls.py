"""The encoder backward calls of one training step (scratch/bwd_calls.npz from tools/dump_bwd_calls.py) replayed one
by one on an idle GPU: the run kernel of each call against the cell-merging scatter (CNC_FLAG_CELL_MERGE, with and without
CNC_FLAG_CELL_CARRY) — time per call (HIP events, median of 20) and the largest difference between the two gradient tables.

    python tools/replay_bwd_calls.py [calls.npz] [--calls 0,4,6]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cnc_amd.backends import gridencoder_backend as be

path = next((a for a in sys.argv[1:] if a.endswith(".npz")), "scratch/bwd_calls.npz")
z = np.load(path)
dev = torch.device("cuda:0")
n_calls = len([k for k in z.files if k.endswith("_N")])
only = None
if "--calls" in sys.argv:
    only = [int(c) for c in sys.argv[sys.argv.index("--calls") + 1].split(",")]
torch.manual_seed(0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


tot_old = tot_new = 0.0
for ci in range(n_calls):
    if only is not None and ci not in only:
        continue
    g = lambda k: z[f"c{ci}_{k}"] if f"c{ci}_{k}" in z.files else None
    N, D, F, L, Rb = int(g("N")), int(g("D")), int(g("F")), int(g("L")), int(g("Rb"))
    x = torch.from_numpy(g("inputs")).to(dev).contiguous()
    offs = torch.from_numpy(g("offsets")).to(dev).to(torch.int32)
    res = torch.from_numpy(g("resolutions")).to(dev).to(torch.int32)
    mli = None if g("mli") is None else torch.from_numpy(g("mli")).to(dev).to(torch.int32)
    ste = bool(g("ste"))
    vxl = vb = sat = None
    if g("vxl") is not None:
        shape = tuple(int(s) for s in g("vxl_shape"))
        vxl = torch.from_numpy(np.unpackbits(g("vxl"))[: int(np.prod(shape))].reshape(shape).astype(bool)).to(dev)
        sat = be.occupancy_sat(vxl)
        vb = be.occupancy_vertex_bits(vxl, sat, [int(r) for r in g("resolutions")])
    rows = int(g("offsets")[-1])
    table = (torch.rand(rows, F, device=dev) * 2 - 1) * (1e-4 if ste else 1.0)
    nz = torch.from_numpy(g("nz")).to(dev)
    grad = torch.randn(N, L, F, device=dev) * nz.unsqueeze(-1)
    grad = grad.reshape(N, L * F).contiguous()
    clip = torch.zeros(1, dtype=torch.int32, device=dev) if ste else None

    def run(out, **kw):
        be.grid_encode_backward(grad, x, table, offs, res, out, N, D, F, L, 0, Rb, None, None, vxl, mli, ste_binary=ste,
                                ste_clip_count=clip, occ_sat=sat, grad_ld=L * F, grad_col=0, vertex_bits=vb, **kw)

    a, b = torch.zeros_like(table), torch.zeros_like(table)
    run(a)
    run(b, cell_merge=True)
    torch.cuda.synchronize()
    scale = a.abs().max().item()
    err = (a - b).abs().max().item()
    # summation-order bound: compare both with a float64 scatter of the same contributions? here: against each other
    scratch = torch.zeros_like(table)
    t_old = timed(lambda: run(scratch))
    line = f"call {ci:2d} D={D} L={L} N={N:7d} masked={vxl is not None} perpoint={mli is not None} ste={ste}: old {t_old*1e3:7.1f} us"
    t_new = timed(lambda: run(scratch, cell_merge=True))
    tot_new += t_new
    line += f" | cells: {t_new*1e3:7.1f} us | cells + carry: {timed(lambda: run(scratch, cell_merge=True, cell_carry=True))*1e3:7.1f} us"
    if D == 3 and F == 8 and vxl is None and mli is None:
        line += f" | merge(all levels): {timed(lambda: run(scratch, interleave_levels=True))*1e3:7.1f} us"
        k = L - 1
        def split():
            be.grid_encode_backward(grad, x, table, offs[:k + 1], res[:k], scratch, N, D, F, k, 0, Rb, None, None, None, None, ste_binary=ste,
                                    ste_clip_count=clip, grad_ld=L * F, grad_col=0, interleave_levels=True)
            be.grid_encode_backward(grad, x, table, offs[k:], res[k:], scratch, N, D, F, 1, 0, Rb, None, None, None, None, ste_binary=ste,
                                    ste_clip_count=clip, grad_ld=L * F, grad_col=k * F)
        line += f" | merge(L-1)+runs(1): {timed(split)*1e3:7.1f} us"
    tot_old += t_old
    print(line + f" | max|diff| {err:.3e} of {scale:.3e}", flush=True)
print(f"sum over calls: old {tot_old:.3f} ms, cells {tot_new:.3f} ms")
