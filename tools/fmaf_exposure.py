"""How much of `query_mask_3D`'s integer output rests on the oracle's reading of nvcc's contraction.

The reference is CUDA; nvcc's default -fmad=true fuses a same-type multiply into the add it feeds, and whether it did so for
`float(idx) * Rb_re + Rb_re` and `overlap += a * b * c` (aligner_kernel.cu:57,71,216-233) cannot be observed without a CUDA
build.  The oracle (and the HIP kernel that follows it) takes both as fused.  This script runs the oracle under all four
readings on the vertex sets the context pass hands the kernel — every vertex of a level, int16, against a 128^3 occupancy
ball (configs[2]'s shapes) — and counts the outputs that change.  CPU only (the oracle is test infrastructure).

    python tools/fmaf_exposure.py            -> profiles/r06_fmaf_exposure.md
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle

oracle.build()
Rb = 128
ax = (np.arange(Rb) + 0.5) / Rb - 0.5
g = np.meshgrid(ax, ax, ax, indexing="ij")
vxl = ((g[0] ** 2 + g[1] ** 2 + g[2] ** 2) < (1.0 / 3.0) ** 2)          # the bench's ball: radius 1.0 in a box of +-1.5
rng = np.random.default_rng(0)
vxl ^= rng.uniform(size=vxl.shape) < 0.01                                # speckle: boxes that straddle set / unset cells
rows = []
for R in (108, 201, 376, 514):
    n_all = (R - 2) ** 3
    if n_all <= 9_000_000:
        v = np.stack(np.meshgrid(*[np.arange(1, R - 1)] * 3, indexing="ij"), -1).reshape(-1, 3)
    else:                                                               # a uniform sample of the level's vertices
        v = rng.integers(1, R - 1, size=(8_000_000, 3))
    v = v.astype(np.int16)
    ref_m, ref_o = oracle.query_mask(v, vxl, resolution=R, contraction=3)
    near = int((ref_m != 0).sum())
    for c, name in ((2, "edge unfused"), (1, "accumulation unfused"), (0, "both unfused")):
        m, o = oracle.query_mask(v, vxl, resolution=R, contraction=c)
        d = o != ref_o
        assert np.array_equal(m, ref_m)                                 # the mask is integer arithmetic on unfused products
        worst = int(np.abs(o.astype(np.int64) - ref_o).max())
        # what the pass does with the value: clamp to >= 1 and normalise per hash slot — a change of 1 in a weight of ~10^3..10^6
        rel = float((np.abs(o.astype(np.float64) - ref_o) / np.maximum(ref_o, 1))[d].max()) if d.any() else 0.0
        rows.append((R, v.shape[0], near, name, int(d.sum()), worst, rel))
        print(rows[-1], flush=True)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_fmaf_exposure.md")
with open(out, "w") as fh:
    fh.write("# r06 — exposure of `query_mask_3D`'s integer outputs to the contraction nvcc may or may not have applied\n\n"
             "`python tools/fmaf_exposure.py` (CPU, the oracle under four readings of aligner_kernel.cu:57,71,216-233; reference\n"
             "reading = both multiply-add pairs fused, as `oracle/cnc_oracle.c` and `csrc/aligner.hip` have them).  Vertices: every\n"
             "vertex of the level (a uniform sample of 8 M above 9 M), int16, against a 128^3 ball with 1 % speckle.\n"
             "`mask` never changes (asserted): it is integer arithmetic on products that no reading fuses.\n\n"
             "| level R | vertices | next to occupied space | reading | `overlap` values that differ | largest difference | largest relative difference |\n|---|---|---|---|---|---|---|\n")
    for r in rows:
        fh.write(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]} | {r[4]} ({100.0 * r[4] / max(r[2], 1):.3f} % of the near ones) | {r[5]} | {r[6]:.2e} |\n")
    fh.write("\n`overlap = int(area * Rb^3 * 1000)` feeds `clamp(min = 1)` and a per-slot normalisation (utils_bpp_acc.py:680-682): a\n"
             "difference of one unit is a relative change of the weight by the last column.\n")
print(open(out).read())
