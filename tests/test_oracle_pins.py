"""Pin the CPU oracle (CPU-only tests): against vectors generated from the reference's own Python
(tests/golden/make_golden.py), against the known answers in the reference's docstrings, and
against an independent NumPy restatement of the interpolation written from the paper semantics."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_grid_index_matches_reference_python_twin(oracle):
    """examples/utils.py:492-511 (int64 torch) vs gridencoder.cu:45-87 (uint32 wrap) restated in C."""
    g = np.load(os.path.join(GOLD, "grid_index.npz"))
    for k in range(int(g["n_cases"])):
        D, R, hs = int(g[f"c{k}_D"]), int(g[f"c{k}_R"]), int(g[f"c{k}_hs"])
        rows = oracle.grid_index(g[f"c{k}_pos"], hs, R)
        if R ** D <= hs or (hs & (hs - 1)) == 0:
            # the Python twin uses int64 products: it agrees with uint32 wrap-around only for
            # dense levels and power-of-two tables (SURVEY.md §7 "Integer semantics")
            assert np.array_equal(rows.astype(np.int64), g[f"c{k}_rows"]), (D, R, hs)
        assert rows.max() < hs


def test_grid_index_dense_vs_hashed_switch(oracle):
    # dense while R^D <= table size: row = x + y*R + z*R^2
    pos = np.array([[1, 2, 3], [17, 17, 17], [0, 0, 0]], np.uint32)
    assert oracle.grid_index(pos, 5832, 18).tolist() == [1 + 2 * 18 + 3 * 324, 17 + 17 * 18 + 17 * 324, 0]
    # hashed otherwise: xor of coordinate * prime, uint32 wrap, mod T
    p = np.array([[5, 7, 11]], np.uint32)
    want = ((5 * 1) ^ ((7 * 2654435761) & 0xFFFFFFFF) ^ ((11 * 805459861) & 0xFFFFFFFF)) % 2 ** 19
    assert oracle.grid_index(p, 2 ** 19, 514)[0] == want
    # a level whose R^D exceeds the table only at the last dimension still hashes (stride loop stops early)
    assert oracle.grid_index(np.array([[3, 4, 5]], np.uint32), 1000, 11)[0] == \
        ((3 ^ ((4 * 2654435761) & 0xFFFFFFFF) ^ ((5 * 805459861) & 0xFFFFFFFF)) % 1000)


def test_slab_test_matches_reference_torch_twin(oracle):
    """nerfacc/grid.py:55-91 (_ray_aabb_intersect).  The torch twin divides by d where the kernel
    multiplies by 1/d, so t-values agree to rounding; hit flags agree away from grazing rays."""
    g = np.load(os.path.join(GOLD, "ray_aabb.npz"))
    for k in range(3):
        near, far, miss = (float(v) for v in g[f"nfm_{k}"])
        t0, t1, hit = oracle.ray_aabb_intersect(g["rays_o"], g["rays_d"], g["aabbs"], near, far, miss)
        ref_hit = g[f"hit_{k}"]
        agree = hit == ref_hit
        assert agree.mean() > 0.999
        both = hit & ref_hit
        # the kernel clips tmin only from below and tmax only from above (utils_grid.cuh:53-54);
        # the torch twin clamps both to [near, far] (grid.py:83-85): compare after the same clamp
        assert np.allclose(np.clip(t0[both], near, far), g[f"t0_{k}"][both], rtol=2e-6, atol=2e-6)
        assert np.allclose(np.clip(t1[both], near, far), g[f"t1_{k}"][both], rtol=2e-6, atol=2e-6)
        assert np.all(t0[~hit] == np.float32(miss)) and np.all(t1[~hit] == np.float32(miss))


def test_scan_docstring_known_answers(oracle):
    """nerfacc/scan.py:36-39,78-81,127-130,170-173."""
    x = np.arange(1, 10, dtype=np.float32)
    starts, cnts = np.array([0, 2, 5]), np.array([2, 3, 4])
    assert oracle.segmented_scan(x, starts, cnts, exclusive=False).tolist() == [1, 3, 3, 7, 12, 6, 13, 21, 30]
    assert oracle.segmented_scan(x, starts, cnts, exclusive=True).tolist() == [0, 1, 0, 3, 7, 0, 6, 13, 21]
    assert oracle.segmented_scan(x, starts, cnts, exclusive=False, prod=True).tolist() == [1, 2, 3, 12, 60, 6, 42, 336, 3024]
    assert oracle.segmented_scan(x, starts, cnts, exclusive=True, prod=True).tolist() == [1, 1, 1, 3, 12, 1, 6, 42, 336]
    # backward of a prefix sum = suffix sum
    assert oracle.segmented_scan(x, starts, cnts, exclusive=False, reverse=True).tolist() == [3, 2, 12, 9, 5, 30, 24, 17, 9]


def test_scan_long_rows_against_float64(oracle):
    rng = np.random.default_rng(0)
    cnts = np.array([0, 1, 31, 32, 33, 64, 65, 1040, 7], np.int64)
    starts = np.cumsum(cnts) - cnts
    x = rng.uniform(0, 1, size=int(cnts.sum())).astype(np.float32)
    inc = oracle.segmented_scan(x, starts, cnts, exclusive=False)
    exc = oracle.segmented_scan(x, starts, cnts, exclusive=True)
    for s, n in zip(starts, cnts):
        ref = np.cumsum(x[s:s + n].astype(np.float64))
        assert np.allclose(inc[s:s + n], ref, rtol=1e-5)
        if n:
            assert exc[s] == 0 and np.allclose(exc[s + 1:s + n], ref[:-1], rtol=1e-5)
    nrm = oracle.segmented_scan(x, starts, cnts, exclusive=False, normalize=True)
    for s, n in zip(starts, cnts):
        if n:
            assert abs(nrm[s + n - 1] - 1) < 1e-6


def test_volrend_docstring_known_answers(oracle):
    """render_transmittance_from_density / render_weight_from_density examples
    (nerfacc/volrend.py:248-255,349-357) recomputed with the oracle's exclusive sum."""
    t0 = np.arange(0, 7, dtype=np.float32)
    t1 = t0 + 1
    sig = np.array([0.4, 0.8, 0.1, 0.8, 0.1, 0.0, 0.9], np.float32)
    starts, cnts = np.array([0, 3, 5]), np.array([3, 2, 2])
    sdt = sig * (t1 - t0)
    trans = np.exp(-oracle.segmented_scan(sdt, starts, cnts, exclusive=True))
    alphas = 1 - np.exp(-sdt)
    assert np.allclose(trans, [1.00, 0.67, 0.30, 1.00, 0.45, 1.00, 1.00], atol=5e-3)
    assert np.allclose(alphas, [0.33, 0.55, 0.095, 0.55, 0.095, 0.00, 0.59], atol=5e-3)
    assert np.allclose(trans * alphas, [0.33, 0.37, 0.03, 0.55, 0.04, 0.00, 0.59], atol=6e-3)


def _numpy_trilinear(x, emb, offs, res, D):
    """Independent restatement from the paper semantics (float64): ring-padded grid, samples map
    to [0.5, R-1.5], corners on the ring are dropped and the weights renormalised."""
    N = x.shape[0]
    F = emb.shape[1]
    out = np.zeros((len(res), N, F))
    primes = [1, 2654435761, 805459861]
    for li, R in enumerate(res):
        hs = int(offs[li + 1] - offs[li])
        p = x.astype(np.float64) * (R - 2) + 0.5
        g = np.floor(p).astype(np.int64)
        fr = p - g
        wsum = np.zeros(N)
        acc = np.zeros((N, F))
        for c in range(2 ** D):
            w = np.ones(N)
            q = np.zeros((N, D), np.int64)
            for d in range(D):
                bit = (c >> d) & 1
                w *= fr[:, d] if bit else (1 - fr[:, d])
                q[:, d] = np.minimum(g[:, d] + bit, R - 1)
            ok = np.all((q > 0) & (q < R - 1), axis=1)
            if R ** D <= hs:
                idx = sum(q[:, d] * R ** d for d in range(D))
            else:
                idx = np.zeros(N, np.int64)
                for d in range(D):
                    idx ^= (q[:, d] * primes[d]) & 0xFFFFFFFF
            idx = idx % hs + offs[li]
            acc += np.where(ok, w, 0)[:, None] * emb[idx]
            wsum += np.where(ok, w, 0)
        out[li] = acc / np.where(wsum == 0, 1e-9, wsum)[:, None]
    return out


@pytest.mark.parametrize("D", [2, 3])
def test_encoder_forward_against_independent_numpy(oracle, D):
    from conftest import make_grid
    res = [6, 9, 14, 20, 31, 44] if D == 3 else [10, 18, 34, 66]
    offs, resl, emb = make_grid(res, 10, D, 4, seed=1)
    x = np.random.default_rng(2).uniform(0, 1, size=(2000, D)).astype(np.float32)
    got = oracle.grid_encode_forward(x, emb, offs, resl)
    want = _numpy_trilinear(x, emb, offs.astype(np.int64), res, D)
    # fp32 vs fp64 arithmetic; a point within 1e-6 of a cell boundary may pick another cell
    close = np.isclose(got, want, rtol=1e-4, atol=2e-5)
    assert close.mean() > 0.9995


def test_encoder_backward_is_adjoint_of_forward(oracle):
    """<forward(E), G> == <E, backward(G)> (the encoder is linear in the table)."""
    from conftest import ball_occupancy, make_grid
    offs, resl, emb = make_grid([6, 9, 14, 20], 10, 3, 8, seed=4)
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, size=(500, 3)).astype(np.float32)
    vxl = ball_occupancy(16, 3)
    G = rng.normal(size=(4, 500, 8)).astype(np.float32)
    y = oracle.grid_encode_forward(x, emb, offs, resl, binary_vxl=vxl)
    ge = oracle.grid_encode_backward(G, x, emb, offs, resl, binary_vxl=vxl)
    lhs = float((y.astype(np.float64) * G).sum())
    rhs = float((emb.astype(np.float64) * ge).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_bernoulli_entropy_golden():
    """utils_bpp_acc.py:1002-1013 evaluated by the reference vs the host mirror."""
    import torch
    from cnc_amd.context import Bernoulli_entropy
    g = np.load(os.path.join(GOLD, "entropy.npz"))
    bits = Bernoulli_entropy()(torch.from_numpy(g["x"]), torch.from_numpy(g["p"])).numpy()
    assert np.array_equal(bits, g["bits"])


def test_range_coder_roundtrip_and_size(oracle):
    rng = np.random.default_rng(0)
    for n in (0, 1, 2, 17, 100000):
        p = rng.uniform(1e-6, 1 - 1e-6, size=n).astype(np.float32)
        s = (rng.uniform(size=n) < p).astype(np.int16)
        bs = oracle.rc_encode(p, s)
        assert np.array_equal(oracle.rc_decode(p, bs), s)
        if n >= 1000:
            ideal = -(np.log2(np.where(s == 1, p, 1 - p).astype(np.float64))).sum()
            assert ideal <= len(bs) * 8 <= ideal * 1.002 + 64
    # extreme but legal probabilities (the reference clamps to [1e-6, 1-1e-6]) incl. unlikely symbols
    p = np.array([1e-6, 1 - 1e-6, 1e-6, 1 - 1e-6, 0.5] * 50, np.float32)
    s = np.array([1, 0, 0, 1, 1] * 50, np.int16)
    assert np.array_equal(oracle.rc_decode(p, oracle.rc_encode(p, s)), s)


@pytest.mark.parametrize("F,ste", [(2, True), (2, False), (8, True)])
def test_torch_cpu_encoder_matches_c_restatement(oracle, F, ste):
    """The pure-PyTorch-CPU encoder (oracle/torch_cpu_encoder.py — the "PyTorch-CPU gridencoder
    fallback" of BASELINE config 1: 16 levels, log2T=19, F=2) against the C restatement on the same
    inputs.  Binarised table: every product is by +-1, so the forward is bit-exact; raw table:
    the C path uses fmaf where torch rounds the product first (<= 1 ulp of the running sum per corner)."""
    import torch
    from cnc_amd.synthetic import RES_16L, level_offsets
    from oracle import torch_cpu_encoder as tce
    rng = np.random.default_rng(7)
    offs = level_offsets(RES_16L, 19, 3)
    N = 3000
    x = rng.random((N, 3), dtype=np.float32)
    x[:8] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0.25], [1e-7, 0.9999999, 0.5],
             [0.03, 0.97, 0.5], [0.999, 0.001, 0.5], [0.25, 0.75, 1.0]]
    emb = (rng.random((int(offs[-1]), F), dtype=np.float32) * 2 - 1) * (1.5 if ste else 1e-4)
    grad = rng.standard_normal((len(RES_16L), N, F)).astype(np.float32)
    want = oracle.grid_encode_forward(x, emb, offs, RES_16L, ste_binary=ste)
    want_g, acc64 = oracle.grid_encode_backward(grad, x, emb, offs, RES_16L, ste_binary=ste, want_acc64=True)
    torch.set_num_threads(4)
    out, g = tce.forward_backward(torch.from_numpy(x), torch.from_numpy(emb), offs, RES_16L,
                                  torch.from_numpy(grad), ste_binary=ste)
    if ste:
        assert np.array_equal(out.numpy(), want)
    else:
        np.testing.assert_allclose(out.numpy(), want, rtol=0, atol=2e-11)     # |emb| <= 1e-4
    # the gradient differs by summation order only: compare with the float64 shadow
    scale = np.abs(acc64).max()
    assert np.abs(g.numpy() - acc64).max() <= 2e-6 * scale
    assert np.abs(want_g - acc64).max() <= 2e-6 * scale
    assert np.array_equal(g.numpy() == 0, acc64 == 0)


@pytest.mark.parametrize("D", [2, 3])
def test_dy_dx_is_the_derivative_of_the_interpolation(oracle, D):
    """The dy_dx branch of kernel_grid (gridencoder.cu:319-395) restated in the oracle, against central
    differences of the oracle's own forward on interior points (all corners valid there, so the
    renormalised forward is plain multilinear interpolation and dy_dx is its exact gradient inside a
    cell), and kernel_input_backward (:588-614) against a float64 contraction."""
    from cnc_amd.synthetic import level_offsets
    res = [6, 9, 14, 20, 31] if D == 3 else [10, 18, 34]
    F = 4
    offs = level_offsets(res, 10, D)
    rng = np.random.default_rng(5)
    emb = rng.standard_normal((int(offs[-1]), F)).astype(np.float32)
    x = rng.uniform(0.3, 0.7, size=(200, D)).astype(np.float32)
    dy = oracle.grid_dy_dx(x, emb, offs, res)
    assert dy.shape == (200, len(res), D, F)
    h = np.float32(2e-4)
    for d in range(D):
        xp, xm = x.copy(), x.copy()
        xp[:, d] += h
        xm[:, d] -= h
        num = (oracle.grid_encode_forward(xp, emb, offs, res) - oracle.grid_encode_forward(xm, emb, offs, res)) \
            / (xp[:, d] - xm[:, d])[None, :, None]                       # [L, N, F]
        for l, R in enumerate(res):
            # only points whose +-h neighbours stay in the same cell along d (piecewise linear)
            s = x[:, d].astype(np.float64) * (R - 2) + 0.5
            same = np.floor(s - h * (R - 2) * 1.01) == np.floor(s + h * (R - 2) * 1.01)
            assert same.sum() > 50
            np.testing.assert_allclose(dy[same, l, d, :], num[l, same, :], rtol=0, atol=2e-2 * (R - 2))
    # out-of-range points give zeros, border vertices read as zero (no renormalisation here)
    xo = np.array([[1.5] + [0.5] * (D - 1), [0.5] * D], np.float32)
    dyo = oracle.grid_dy_dx(xo, emb, offs, res)
    assert np.all(dyo[0] == 0) and np.any(dyo[1] != 0)
    g = rng.standard_normal((len(res), 200, F)).astype(np.float32)
    gi = oracle.input_backward(g, dy)
    want = np.einsum("lnf,nldf->nd", g.astype(np.float64), dy.astype(np.float64))
    np.testing.assert_allclose(gi, want, rtol=0, atol=1e-5 * np.abs(want).max())


def _render_golden():
    g = np.load(os.path.join(GOLD, "render.npz"))
    R, M = g["sigmas"].shape
    starts = np.arange(R, dtype=np.int64) * M
    cnts = np.full(R, M, np.int64)
    ri = np.repeat(np.arange(R, dtype=np.int64), M)
    return g, R, M, starts, cnts, ri


@pytest.mark.parametrize("case", ["plain", "prefix"])
def test_volume_rendering_matches_reference_functions(oracle, case):
    """tests/golden/render.npz: the reference's own render_weight_from_density / accumulate_along_rays /
    rendering tail (batched branches, run on CPU by make_golden_render.py) vs the oracle's flattened
    restatement on the same rays.  The reference's batched branch sums with torch.cumsum, the flattened one
    (and the oracle) with the 32-wide tile tree: same values up to float32 association."""
    g, R, M, starts, cnts, ri = _render_golden()
    flat = lambda a: np.ascontiguousarray(a.reshape(R * M, *a.shape[2:]))
    prefix = flat(g["prefix"]) if case == "prefix" else None
    w, tr, al = oracle.render_weight_from_density(flat(g["t_starts"]), flat(g["t_ends"]), flat(g["sigmas"]), starts,
                                                  cnts, prefix_trans=prefix)
    # alpha = 1 - exp(-x): an ulp of exp() near 1 is 1.2e-7 ABSOLUTE on alpha, whatever its size
    assert np.allclose(al, flat(g[f"{case}_alphas"]), rtol=2e-6, atol=2.5e-7)
    assert np.allclose(tr, flat(g[f"{case}_trans"]), rtol=3e-5, atol=1e-9)     # exp of a 45-term sum
    assert np.allclose(w, flat(g[f"{case}_weights"]), rtol=3e-5, atol=2.5e-7)
    col, op, dsum = oracle.composite(w, flat(g["rgbs"]), flat(g["t_starts"]), flat(g["t_ends"]), ri, R, finalize=False)
    assert np.allclose(col, g[f"{case}_colors"], rtol=1e-5, atol=1e-7)
    assert np.allclose(op, g[f"{case}_opacity"], rtol=1e-5, atol=1e-7)
    assert np.allclose(dsum, g[f"{case}_depth_sum"], rtol=1e-5, atol=1e-7)
    col, op, dep = oracle.composite(w, flat(g["rgbs"]), flat(g["t_starts"]), flat(g["t_ends"]), ri, R,
                                    render_bkgd=g["bkgd"], finalize=True)
    assert np.allclose(col, g[f"{case}_colors_bkgd"], rtol=1e-5, atol=1e-6)
    assert np.allclose(dep, g[f"{case}_depth"], rtol=1e-5, atol=1e-6)
