"""Regenerate tests/golden/march_query.npz (THIS container only: imports the reference's Python).

    python tests/golden/make_golden_march.py

Cross-check of the occupancy-grid marcher against the reference's OWN occupancy lookup: seeded rays are
marched by the oracle (oracle.traverse_grids, the restatement of nerfacc/cuda/csrc/grid.cu:71-318), the
sample positions o + d*t are handed to the reference's `nerfacc.grid._query` (nerfacc/grid.py:204-240)
together with the grid, and its answer per sample is stored.  A marcher that walks the wrong cells (an
off-by-one in setup_traversal, a wrong tie-break in single_traversal) emits samples in cells the
reference's lookup calls empty.  For completeness as well as soundness the CANDIDATE set is stored: the
same rays marched through an all-occupied grid (the march advances t by repeated `+= dt` whether a cell is
empty or not, so these are exactly the mid points the real march can ever emit, same floats), with the
reference lookup's answer on the real grid for every candidate.  The real march must emit exactly the
candidates the reference calls occupied (samples whose position rounds onto a cell face excepted).
Stored: inputs (rays, grid seed parameters), candidate t-values, the reference's answers.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402


def scene(res, seed):
    c = (np.arange(res, dtype=np.float32) + 0.5) / res * 3.0 - 1.5
    gx, gy, gz = np.meshgrid(c, c, c, indexing="ij")
    b = ((gx * gx + gy * gy + gz * gz) < 1.0)[None]
    b ^= (np.random.default_rng(seed).uniform(size=b.shape) < 0.03)
    return b


def main():
    sys.path.insert(0, REF)
    from nerfacc.grid import _query          # the reference's lookup (pure torch)
    from test_np_twins import _rays
    out = {}
    aabb = np.array([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], np.float32)
    for k, (res, step, n, seed) in enumerate([(32, 2e-2, 600, 31), (128, 5e-3, 300, 32)]):
        o, d = _rays(n, seed)
        b = scene(res, seed + 100)
        iv, sm, term = oracle.traverse_grids(o, d, np.ones_like(b), aabb, None, None, step, 0.0)
        t = np.asarray(sm["vals"])
        ri = np.asarray(sm["ray_indices"])
        pos = o[ri].astype(np.float64) + d[ri].astype(np.float64) * t[:, None].astype(np.float64)
        occ, sel = _query(torch.from_numpy(pos.astype(np.float32)), torch.from_numpy(b), torch.from_numpy(aabb[0]))
        out[f"c{k}_res"] = np.int64(res); out[f"c{k}_step"] = np.float64(step); out[f"c{k}_n"] = np.int64(n)
        out[f"c{k}_seed"] = np.int64(seed)
        out[f"c{k}_t"] = t
        out[f"c{k}_ray"] = ri.astype(np.int32)
        out[f"c{k}_ref_occupied"] = (occ.numpy().astype(bool) & sel.numpy().astype(bool))
        print(f"case {k}: {len(t)} candidates, reference _query says occupied for {out[f'c{k}_ref_occupied'].mean():.4f}")
    out["n_cases"] = np.int64(2)
    np.savez_compressed(os.path.join(HERE, "march_query.npz"), **out)


if __name__ == "__main__":
    main()
