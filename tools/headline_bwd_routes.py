"""The headline backward call (bench.py: one 2^20-sample chunk from the middle of the 800x800 frame, 16 levels, F = 8,
STE) through each scatter route the library has — what `profiles/<tag>_headline_backward_routes.md` tabulates:

    --route product   the bench's call: k_grid_encode_bwd_merge on the coarse levels + the binned finest six
    --route runs      k_grid_encode_bwd on all 16 levels (runs of one cell along a ray, 256 samples per block)
    --route merge     k_grid_encode_bwd_merge on all 16 levels (1,024-sample blocks, one atomic set per distinct cell)
    --route cells     k_grid_encode_bwd_cells (as merge + lanes of 4 cells per wave)
    --route carry     ... with the x-neighbour carry (a vertex shared by two cells of the block is written once)
    --levels K        only the K coarsest levels (10 = the ones the product route gives the merge kernel), not for product
    --count           no kernel: distinct cells per group of 1,024 / 2,048 / 4,096 consecutive samples, per level — what a
                      block that carried its cell table across 2 / 4 blocks would send (an upper bound on what it saves)

One route per process so that a counter pass (rocprofv3 --pmc TCC_ATOMIC_sum) sees only that route's kernels; prints the
time per call (HIP events, median of 20).   tools/collect_profiles.sh <tag> routes  runs all of them."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bench import enc, synthetic, D, F, L

route = sys.argv[sys.argv.index("--route") + 1] if "--route" in sys.argv else "product"
dev = torch.device("cuda:0")
w = bench.build_workload(dev, 0)
box = {}
bench.march_frame(w, box)
xs = bench.probe_chunk_of(box["ex"]["positions"]).contiguous()
n = xs.shape[0]
torch.cuda.synchronize()

if "--count" in sys.argv:
    res = synthetic.RES_16L
    x = xs.double().cpu().numpy()
    rows = []
    for l, R in enumerate(res):
        # the cell of a sample as Corners::setup floors it, floor(x (R - 2) + 0.5) (in float64 here: flooring differs from
        # the kernel's fp32 on a vanishing share of samples, and the count only needs cell identity)
        c = np.floor(x * (R - 2) + 0.5).astype(np.int64)
        key = c[:, 0] | c[:, 1] << 16 | c[:, 2] << 32
        runs = int((np.diff(key) != 0).sum()) + 1
        per = {}
        for G in (256, 1024, 2048, 4096):
            k = key[: n // G * G].reshape(-1, G)
            k = np.sort(k, axis=1)
            per[G] = int((np.diff(k, axis=1) != 0).sum() + k.shape[0])
        rows.append((l, R, runs, per))
        print(f"level {l:2d} R={R:5d}: runs {runs:8d} | distinct cells per 256: {per[256]:8d}  1,024: {per[1024]:8d}  "
              f"2,048: {per[2048]:8d}  4,096: {per[4096]:8d}", flush=True)
    tot = {G: sum(r[3][G] for r in rows[:10]) for G in (256, 1024, 2048, 4096)}
    print("coarse ten levels, cells summed:", json.dumps(tot), "runs:", sum(r[2] for r in rows[:10]))
    sys.exit(0)

out = torch.empty((L, n, F), device=dev)
enc.pack_sign_bits(w["table"], w["bits"], w["clip"])
enc.grid_encode_forward_bits(xs, w["bits"], w["offsets"], w["resolutions"], out, n, D, F, L, 128)
gt = torch.zeros_like(w["table"])
plan = enc.plan_binned_levels(synthetic.RES_16L, w["offsets_host"], D, F, n)
kw = dict(ste_binary=True, ste_clip_count=w["clip"])
if route == "product":
    kw["binned"] = plan
elif route == "merge":
    kw["interleave_levels"] = True
elif route == "cells":
    kw["cell_merge"] = True
elif route == "carry":
    kw["cell_merge"] = kw["cell_carry"] = True
elif route != "runs":
    raise SystemExit(f"unknown route {route}")


K = int(sys.argv[sys.argv.index("--levels") + 1]) if "--levels" in sys.argv else L
offs_k, res_k = w["offsets"][:K + 1].contiguous(), w["resolutions"][:K].contiguous()
grad = out.permute(1, 0, 2).reshape(n, L * F).contiguous()       # [n, L * F]: a level subset is a column range (grad_ld)


def call(table=gt, kw=kw):
    if K == L:
        enc.grid_encode_backward(out, xs, w["table"], w["offsets"], w["resolutions"], table, n, D, F, L, 0, 128, None, None,
                                 None, None, **kw)
    else:
        enc.grid_encode_backward(grad, xs, w["table"], offs_k, res_k, table, n, D, F, K, 0, 128, None, None, None, None,
                                 grad_ld=L * F, grad_col=0, **kw)


for _ in range(3):
    call()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); call(); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b))
ref = torch.zeros_like(gt)
gt.zero_(); call()
call(ref, dict(ste_binary=True, ste_clip_count=w["clip"]))
torch.cuda.synchronize()
print(json.dumps({"route": route, "levels": K, "samples": n, "ms_per_call": round(sorted(ts)[10], 4),
                  "max_abs_diff_vs_runs": float((gt - ref).abs().max()), "largest": float(ref.abs().max())}))
