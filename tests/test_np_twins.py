"""The oracle (oracle/cnc_oracle.c) against the independent NumPy restatements in tests/np_twins.py,
which were written from the reference's CUDA sources by a different route (vectorised, no shared
code).  These are the second pins for the oracle functions that have no executable reference twin:
kernel_grid / kernel_grid_backward, cnt_np_embed, query_mask_3D(_qlist), align_and_pack and the
occupancy-grid DDA.  CPU only."""
import numpy as np
import pytest

import np_twins as tw
from conftest import ball_occupancy, make_grid

pytestmark = pytest.mark.filterwarnings("ignore::RuntimeWarning")     # 1/0 for axis-parallel rays, as on the GPU


def _points(n, D, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, size=(n, D)).astype(np.float32)
    x[:: 97] = rng.integers(0, 2, size=x[:: 97].shape)        # exact 0 / 1 coordinates
    x[5:: 211] += 1.5                                          # out of range
    x[7:: 223, 0] = np.float32(0.5)                            # exactly on a vertex plane of even levels
    return x


@pytest.mark.parametrize("D,res,log2T", [(3, [6, 9, 14, 20, 31, 44, 66], 10), (2, [10, 18, 34, 66, 130], 10),
                                          (3, [18, 33, 59, 108], 14)])
@pytest.mark.parametrize("F", [2, 8])
def test_forward_bitexact_on_sign_tables(oracle, D, res, log2T, F):
    """STE tables (+-1): every product w*e is exact, so fma == mul+add and the comparison is bit-exact
    with no emulation caveat."""
    offs, resl, emb = make_grid(res, log2T, D, F, seed=3)
    x = _points(6000, D, seed=4)
    want = oracle.grid_encode_forward(x, emb, offs, resl, ste_binary=True)
    got = tw.grid_encode_forward(x, emb, offs, resl, ste_binary=True)
    assert np.array_equal(got, want)


def test_forward_bitexact_on_power_of_two_tables(oracle):
    offs, resl, emb = make_grid([6, 9, 14, 20, 31], 10, 3, 4, seed=5)
    rng = np.random.default_rng(6)
    emb = (np.sign(emb) * np.exp2(rng.integers(-6, 3, size=emb.shape))).astype(np.float32)
    x = _points(4000, 3, seed=7)
    assert np.array_equal(tw.grid_encode_forward(x, emb, offs, resl), oracle.grid_encode_forward(x, emb, offs, resl))


def test_forward_general_table_within_one_ulp_of_the_fma_emulation(oracle):
    offs, resl, emb = make_grid([6, 9, 14, 20, 31], 10, 3, 8, seed=8)
    x = _points(4000, 3, seed=9)
    want = oracle.grid_encode_forward(x, emb, offs, resl)
    got = tw.grid_encode_forward(x, emb, offs, resl)
    mism = got != want
    # the float64 emulation of fmaf can double-round; it must be rare and never exceed one ulp
    assert mism.mean() < 1e-5
    assert np.all(np.abs(got - want)[mism] <= np.spacing(np.abs(want[mism])))


def test_forward_with_occupancy_mask(oracle):
    offs, resl, emb = make_grid([6, 9, 14, 20, 31, 44], 10, 3, 4, seed=10)
    vxl = ball_occupancy(16, 3)
    x = _points(5000, 3, seed=11)
    want = oracle.grid_encode_forward(x, emb, offs, resl, binary_vxl=vxl, ste_binary=True)
    got = tw.grid_encode_forward(x, emb, offs, resl, vxl=vxl, ste_binary=True)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("ste", [False, True])
def test_backward_float64_sums_agree(oracle, ste):
    offs, resl, emb = make_grid([6, 9, 14, 20, 31, 44], 10, 3, 8, seed=12)
    x = _points(7000, 3, seed=13)
    g = np.random.default_rng(14).normal(size=(len(resl), x.shape[0], 8)).astype(np.float32)
    _, acc = oracle.grid_encode_backward(g, x, emb, offs, resl, ste_binary=ste, want_acc64=True)
    acc_np, mag_np, cnt = tw.grid_encode_backward64(g, x, emb, offs, resl, ste_binary=ste)
    # same float32 contributions, summed in float64 in two different orders
    assert np.all(np.abs(acc - acc_np) <= 1e-12 * (mag_np + 1e-300) * np.maximum(cnt, 1)[:, None])
    assert np.array_equal(acc == 0, acc_np == 0) or np.abs(acc - acc_np).max() < 1e-12


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_cnt_np_embed_counts(oracle, axis):
    R, T, F = 34, 2 ** 12, 4
    rng = np.random.default_rng(15)
    emb = np.where(rng.uniform(size=(T, F)) > 0.4, 1.0, -1.0).astype(np.float32)
    pts = rng.integers(0, R, size=(20000, 3)).astype(np.int16)        # includes border vertices
    want = oracle.cnt_np_embed(pts, emb, R, T, axis)
    got = tw.cnt_np_embed(pts, emb, R, T, axis)
    assert np.array_equal(got.reshape(want.shape), want)


@pytest.mark.parametrize("Rb", [16, 32, 128])
@pytest.mark.parametrize("R", [18, 59, 201, 514])
def test_query_mask_3D(oracle, Rb, R):
    vxl = ball_occupancy(Rb, 3)
    rng = np.random.default_rng(16)
    pts = rng.integers(0, R, size=(3000, 3)).astype(np.int16)
    m, ov = oracle.query_mask(pts, vxl, resolution=R)
    m2, ov2 = tw.query_mask_3D(pts, vxl, R)
    assert np.array_equal(m, m2)
    assert np.array_equal(ov, ov2)


def test_query_mask_3D_qlist(oracle):
    vxl = ball_occupancy(32, 3)
    rng = np.random.default_rng(17)
    rl = rng.choice([18, 24, 33, 44, 59, 80, 108, 148], size=4000).astype(np.int64)
    pts = (rng.uniform(size=(4000, 3)) * rl[:, None]).astype(np.int16)
    m, ov = oracle.query_mask(pts, vxl, resolution_list=rl)
    m2, ov2 = tw.query_mask_3D(pts, vxl, rl)
    assert np.array_equal(m, m2) and np.array_equal(ov, ov2)


def test_align_and_pack(oracle):
    rng = np.random.default_rng(18)
    cnt = rng.integers(0, 9, size=500).astype(np.int64)
    cnt[::7] = 0
    feat = rng.normal(size=(int(cnt.sum()), 4)).astype(np.float32)
    for V in (0.0, -3.5):
        assert np.array_equal(oracle.align_and_pack_forward(feat, cnt, V=V), tw.align_and_pack_forward(feat, cnt, V))


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = (o / np.linalg.norm(o, axis=1, keepdims=True) * rng.uniform(2.0, 4.5, size=(n, 1))).astype(np.float32)
    tgt = rng.uniform(-1.2, 1.2, size=(n, 3))
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o[::50] = rng.uniform(-0.5, 0.5, size=o[::50].shape)           # cameras inside the box
    d[3::60, 1] = 0.0                                              # an axis with zero direction
    d[4::70] = np.array([0.0, 0.0, 1.0], np.float32)               # two zero axes
    o[4::70] = np.array([0.1, -0.2, -3.0], np.float32)
    o[9::80] = np.array([5.0, 5.0, 5.0], np.float32)               # misses
    d[9::80] = np.array([1.0, 0.0, 0.0], np.float32)
    return o, d


@pytest.mark.parametrize("limit", [-1, 5])
@pytest.mark.parametrize("res,step", [(32, 2e-2), (128, 5e-3)])
def test_traverse_grids(oracle, res, step, limit):
    n = 1500 if res == 32 else 400
    o, d = _rays(n, seed=19)
    aabb = np.array([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], np.float32)
    c = (np.arange(res, dtype=np.float32) + 0.5) / res * 3.0 - 1.5
    gx, gy, gz = np.meshgrid(c, c, c, indexing="ij")
    binaries = ((gx * gx + gy * gy + gz * gz) < 1.0)[None]
    binaries ^= (np.random.default_rng(20).uniform(size=binaries.shape) < 0.05)   # holes and floaters
    near = np.random.default_rng(21).uniform(0, step, size=n).astype(np.float32)  # stratified jitter
    far = np.full(n, 1e10, np.float32)
    tmin, tmax, hits = oracle.ray_aabb_intersect(o, d, aabb)
    iv, sm, term = oracle.traverse_grids(o, d, binaries, aabb, near, far, step, 0.0,
                                         traverse_steps_limit=limit, over_allocate=limit > 0)
    got = tw.traverse_grids(o, d, binaries, aabb[0], tmin[:, 0], tmax[:, 0], hits[:, 0], near, far, step, limit)
    cnts = np.asarray(sm["chunk_cnts"])
    assert np.array_equal(got["counts"], cnts)
    assert cnts.sum() > (10 if limit < 0 else 2) * n and (cnts == 0).any()
    valid = np.asarray(sm["is_valid"]).astype(bool) if sm.get("is_valid") is not None else np.ones(len(sm["vals"]), bool)
    assert np.array_equal(np.asarray(sm["vals"])[valid], got["t_mid"])
    assert np.array_equal(np.asarray(sm["ray_indices"])[valid], got["ray"])
    left = np.asarray(iv["vals"])[np.asarray(iv["is_left"]).astype(bool)]
    right = np.asarray(iv["vals"])[np.asarray(iv["is_right"]).astype(bool)]
    assert np.array_equal(left, got["t_left"]) and np.array_equal(right, got["t_right"])
    # two-pass mode: the fill pass skips rays without samples (grid.cu:103-106), their terminate plane is
    # never written (torch::empty in the reference) — compare the rays that marched
    wrote = np.ones(n, bool) if limit > 0 else cnts > 0
    assert np.array_equal(np.asarray(term)[wrote], got["terminate"][wrote])
    # interval count: one edge per sample + one more per interval run
    assert np.array_equal(np.asarray(iv["chunk_cnts"]),
                          cnts + np.bincount(got["ray"][got["first"]], minlength=n))


@pytest.mark.parametrize("k", [0, 1])
def test_marcher_emits_exactly_what_the_reference_lookup_calls_occupied(oracle, k):
    """tests/golden/march_query.npz (made by make_golden_march.py with the reference's own
    `nerfacc.grid._query`): candidate mid points of every ray + the reference lookup's verdict on the real
    grid.  The march must emit exactly the occupied candidates; the only tolerated differences are
    candidates whose position lies within float rounding of a cell face."""
    o, d, b, aabb, step, cand_t, cand_ray, ref_occ = march_query_case(k)
    iv, sm, term = oracle.traverse_grids(o, d, b, aabb, None, None, step, 0.0)
    check_against_reference_lookup(o, d, np.asarray(sm["vals"]), np.asarray(sm["ray_indices"]), cand_t, cand_ray,
                                   ref_occ, b.shape[-1])


def march_query_case(k):
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "march_query.npz"))
    res, step, n, seed = int(G[f"c{k}_res"]), float(G[f"c{k}_step"]), int(G[f"c{k}_n"]), int(G[f"c{k}_seed"])
    o, d = _rays(n, seed)
    c = (np.arange(res, dtype=np.float32) + 0.5) / res * 3.0 - 1.5
    gx, gy, gz = np.meshgrid(c, c, c, indexing="ij")
    b = ((gx * gx + gy * gy + gz * gz) < 1.0)[None]
    b ^= (np.random.default_rng(seed + 100).uniform(size=b.shape) < 0.03)
    aabb = np.array([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], np.float32)
    return o, d, b, aabb, step, G[f"c{k}_t"], G[f"c{k}_ray"].astype(np.int64), G[f"c{k}_ref_occupied"]


def check_against_reference_lookup(o, d, t, ri, cand_t, cand_ray, ref_occ, res):
    # every emitted sample is a candidate (same float): match by (ray, t) key
    key = lambda r, tt: (r.astype(np.int64) << 32) | tt.view(np.uint32).astype(np.int64)
    kc, ke = key(cand_ray, cand_t), key(ri, t)
    assert np.isin(ke, kc).all()
    emitted = np.isin(kc, ke)
    diff = emitted != ref_occ
    assert diff.mean() < 2e-4
    # the disagreements sit on cell faces
    pos = o[cand_ray[diff]].astype(np.float64) + d[cand_ray[diff]].astype(np.float64) * cand_t[diff, None]
    u = (pos + 1.5) / 3.0 * res
    assert np.all(np.abs(u - np.round(u)).min(axis=1) < 1e-3)
