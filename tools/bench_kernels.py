"""Per-kernel timings at BASELINE sizes for the kernels that are not on bench.py's timed path
(context-pass config C, eval march, scans).  Prints a markdown table: algorithmic bytes, time, GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cnc_amd import synthetic
from cnc_amd.backends import gridencoder_backend as ge, nerfacc_cuda as nc, pack_and_align as pa
from cnc_amd.nerfacc import grid as ngrid

dev = torch.device("cuda:0")
rows = []
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def rec(name, units, unit_name, bytes_per_unit, ms):
    rows.append((name, units, unit_name, bytes_per_unit, ms, units / ms / 1e6, units * bytes_per_unit / ms / 1e6))

torch.manual_seed(0)
binaries = synthetic.ball_binaries(128, radius=1.0, device=dev)
vxl = binaries[0].contiguous()
# --- query_mask_3D_qlist at the training size (8.44 M vertices) and scalar at a 2e7-vertex encode chunk
res_list = torch.tensor(synthetic.RES_3D_REF, device=dev)
for N, label in ((8_440_000, "training step (qlist)"), (20_000_000, "encode chunk (R=514)")):
    lv = torch.randint(3, 12, (N,), device=dev)
    R = res_list[lv]
    pts = (torch.rand((N, 3), device=dev) * R[:, None]).to(torch.int16).contiguous()
    mask = torch.zeros(N, dtype=torch.int16, device=dev); ov = torch.zeros(N, dtype=torch.int32, device=dev)
    if "qlist" in label:
        ms = timeit(lambda: pa.query_mask_3D_qlist(pts, vxl, mask, ov, R.contiguous(), N))
        rec(f"query_mask_3D_qlist, {label}", N, "vertices", 6 + 8 + 6, ms)
    else:
        pts = (torch.rand((N, 3), device=dev) * 514).to(torch.int16).contiguous()
        ms = timeit(lambda: pa.query_mask_3D(pts, vxl, mask, ov, 514, N))
        rec(f"query_mask_3D, {label}", N, "vertices", 6 + 6, ms)
# --- align_and_pack (reference dataflow) vs segment_weighted_sum on the same ragged input
cnt = torch.randint(1, 40, (150_000,), device=dev); cnt[::1000] = 288
cum = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(cnt, 0)])
T, M = int(cum[-1]), int(cnt.max())
feat = torch.randn(T, 8, device=dev); w = torch.rand(T, device=dev)
ms = timeit(lambda: pa.align_and_pack_forward(feat, cnt, cum, 150_000, M, 8, 0.0, 3).sum(1))
rec(f"align_and_pack_forward + sum (N=150k slots, M={M}, F=8)", T, "rows", 32 + 150_000 * M * 32 * 2 / T, ms)
ms = timeit(lambda: pa.segment_weighted_sum(feat, w, cum, 1))
rec("segment_weighted_sum (same input, fused)", T, "rows", 36, ms)
# --- cnt_np_embed at full size
m_idx = torch.nonzero(binaries[0])  # occupied cells
base = (m_idx[:, None, :] * 4 + torch.stack(torch.meshgrid(*[torch.arange(-1, 5, device=dev)] * 3, indexing="ij"), -1).reshape(1, -1, 3) + 1).reshape(-1, 3)
lin = torch.unique(base[:, 0] * 514 * 514 + base[:, 1] * 514 + base[:, 2])
verts = torch.stack([lin // (514 * 514), (lin // 514) % 514, lin % 514], -1).to(torch.int16).contiguous()
emb = torch.sign(torch.randn(2 ** 19, 8, device=dev))
out = torch.zeros((512, 512, 8, 2), device=dev)
for ax, nm in ((0, "xy"), (1, "xz")):
    ms = timeit(lambda: ge.cnt_np_embed(verts, emb, out, verts.shape[0], 514, 8, 2 ** 19, ax))
    rec(f"cnt_np_embed axis={nm} ({verts.shape[0]/1e6:.1f} M vertices)", verts.shape[0], "vertices", 6 + 32, ms)
# --- the same votes from the per-refresh sorted plan (no atomics)
plan = ge.VotePlan(verts, 514, 2 ** 19)
outp = torch.empty((512, 512, 8, 2), device=dev)
gos = torch.randn((512, 512, 8, 2), device=dev)
gemb = torch.zeros_like(emb)
for ax, nm in ((0, "xy"), (1, "xz")):
    ms = timeit(lambda: ge.cnt_np_embed_planned(plan, emb, outp, 8, ax))
    rec(f"cnt_np_embed_planned axis={nm} (sorted plan, no atomics)", verts.shape[0], "vertices", 4 + 32, ms)
    ms = timeit(lambda: ge.cnt_np_embed_planned_backward(plan, emb, gos, gemb, 8, ax))
    rec(f"cnt_np_embed_planned_backward axis={nm}", verts.shape[0], "vertices", 4 + 64, ms)
# --- eval march: 800x800, over-allocated 64-step rounds
o, d = synthetic.pinhole_rays(800, 800, 0.6911, 4.0, 0.7, 0.5, device=dev)
aabbs = torch.tensor([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], device=dev)
t0, t1, hits = ngrid.ray_aabb_intersect(o, d, aabbs)
ts = torch.cat([t0, t1], -1); ti = torch.arange(2, device=dev).expand(o.shape[0], 2).contiguous()
near = torch.zeros(o.shape[0], device=dev); far = torch.full((o.shape[0],), 1e10, device=dev)
maskr = torch.ones(o.shape[0], dtype=torch.bool, device=dev)
box = {}
def ev():
    box["r"] = ngrid.traverse_grids(o, d, binaries, aabbs, near, far, 5e-3, 0.0, 64, True, maskr, ts, ti, hits)
ms = timeit(ev, n=5)
ns = int(box["r"][1].packed_info[:, 1].sum())
rec("traverse_grids eval round (640k rays, <=64 samples/ray, over-allocated)", 640_000, "rays", 24 + 27 * ns / 640_000, ms)
def tr():
    box["t"] = ngrid.traverse_grids(o, d, binaries, aabbs, step_size=5e-3, cone_angle=0.0)
ms = timeit(tr, n=5)
S = box["t"][1].vals.shape[0]
rec("traverse_grids training form (two passes + cumsum + host sync)", 640_000, "rays", 24 + 27 * S / 640_000, ms)
# --- sample positions for the field
sm = box["t"][1]
ms = timeit(lambda: nc.sample_positions(o, d, sm.ray_indices, sm.vals, None, aabbs[0]))
rec(f"sample_positions ({S/1e6:.0f} M samples)", S, "samples", 8 + 4 + 12, ms)
# --- segmented scans over the frame's samples
starts, cnts = sm.packed_info[:, 0].contiguous(), sm.packed_info[:, 1].contiguous()
x = torch.rand(S, device=dev)
ms = timeit(lambda: nc.exclusive_sum(starts, cnts, x, False, False))
rec(f"exclusive_sum ({S/1e6:.0f} M samples, 640k rays)", S, "samples", 8, ms)
ms = timeit(lambda: nc.exclusive_sum(starts, cnts, x, False, True))
rec("exclusive_sum backward (reverse)", S, "samples", 8, ms)
print("| kernel | units | algorithmic B/unit | ms | M units/s | GB/s (algorithmic) | frac of 8 TB/s |")
print("|---|---|---|---|---|---|---|")
for name, units, un, bpu, ms, ups, gbs in rows:
    print(f"| {name} | {units:,} {un} | {bpu:.0f} | {ms:.3f} | {ups:,.0f} | {gbs:,.0f} | {gbs/8000:.3f} |")
