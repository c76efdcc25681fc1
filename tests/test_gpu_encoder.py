"""HIP hash-grid encoder (through the `_gridencoder` mirror -> C ABI) vs the CPU oracle.
Forward: bit-exact.  Backward: fp32 atomics reorder the sum, so the bound is derived from the
oracle's float64 shadow and the sum of |contributions| instead of a guessed tolerance."""
import numpy as np
import pytest
import torch

from conftest import ball_occupancy, make_grid

pytestmark = pytest.mark.gpu

RES3 = [6, 9, 14, 20, 31, 44]
RES2 = [10, 18, 34, 66]


def _points(N, D, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, size=(N, D)).astype(np.float32)
    if N >= 16:   # edge cases the reference handles explicitly
        x[0] = 0.0
        x[1] = 1.0
        x[2, 0] = -1e-3           # out of range -> zeros, no gradient
        x[3, D - 1] = 1.0 + 1e-3
        x[4] = 0.5
        x[5] = np.float32(1.0) - np.float32(2 ** -24)
        x[6] = np.float32(2 ** -30)
    return x


def _vertex_bits(dev, vxl, res, use):
    """(sat, (words, offsets)) of an occupancy grid for the levels `res`: the per-corner box test read from one bit
    per vertex (cnc_grid_vertex_bits) instead of the scan / the summed-volume table."""
    if not use or vxl is None:
        return None, None
    from cnc_amd.backends import gridencoder_backend as be
    v = torch.as_tensor(vxl, device=dev)
    sat = be.occupancy_sat(v)
    return sat, be.occupancy_vertex_bits(v, sat, [int(r) for r in res])


def _fwd_gpu(dev, x, emb, offs, res, L, vxl=None, mli=None, ste=False, vbits=False):
    from cnc_amd.backends import gridencoder_backend as be
    t = lambda a, dt=None: None if a is None else torch.as_tensor(a, device=dev)
    N, D = x.shape
    F = emb.shape[1]
    out = torch.empty((L, N, F), dtype=torch.float32, device=dev)
    Rb = 128 if vxl is None else vxl.shape[-1]
    sat, vb = _vertex_bits(dev, vxl, res, vbits)
    be.grid_encode_forward(t(x), t(emb), t(offs), t(res), out, N, D, F, L, 0, Rb, 0.0, None,
                           t(vxl), t(mli), ste_binary=ste, occ_sat=sat, vertex_bits=vb)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _bwd_gpu(dev, g, x, emb, offs, res, vxl=None, mli=None, ste=False, vbits=False, route="runs"):
    from cnc_amd.backends import gridencoder_backend as be
    t = lambda a: None if a is None else torch.as_tensor(a, device=dev)
    L, N, F = g.shape
    D = x.shape[1]
    ge = torch.zeros(emb.shape, dtype=torch.float32, device=dev)
    Rb = 128 if vxl is None else vxl.shape[-1]
    sat, vb = _vertex_bits(dev, vxl, res, vbits)
    be.grid_encode_backward(t(g), t(x), t(emb), t(offs), t(res), ge, N, D, F, L, 0, Rb, None, None,
                            t(vxl), t(mli), ste_binary=ste, occ_sat=sat, vertex_bits=vb,
                            cell_merge=route != "runs", cell_carry=route == "cells+carry")
    torch.cuda.synchronize()
    return ge.cpu().numpy()


# which scatter serves the call: the run-merging atomic kernel, or the cell-merging one (CNC_FLAG_CELL_MERGE: the context
# pass's route), without and with its x-neighbour carry
ROUTES = pytest.mark.parametrize("route", ["runs", "cells", "cells+carry"])


@pytest.mark.parametrize("D", [2, 3])
@pytest.mark.parametrize("F", [1, 2, 4, 8, 16, 32])
def test_forward_bit_exact(cuda, oracle, D, F):
    res = RES3 if D == 3 else RES2
    offs, resl, emb = make_grid(res, 10, D, F, seed=F)
    x = _points(3001, D, seed=D * 10 + F)
    want = oracle.grid_encode_forward(x, emb, offs, resl)
    got = _fwd_gpu(cuda, x, emb, offs, resl, len(res))
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert np.all(got[:, 2] == 0) and np.all(got[:, 3] == 0)   # OOB points


VBITS = pytest.mark.parametrize("vbits", [False, True], ids=["box_scan", "vertex_bits"])


@VBITS
@pytest.mark.parametrize("D", [2, 3])
@pytest.mark.parametrize("ste", [False, True])
def test_forward_with_occupancy_mask(cuda, oracle, D, ste, vbits):
    res = RES3 if D == 3 else RES2
    offs, resl, emb = make_grid(res, 10, D, 8, seed=3)
    vxl = ball_occupancy(16 if D == 3 else 32, D)
    x = _points(2500, D, seed=5)
    want = oracle.grid_encode_forward(x, emb, offs, resl, binary_vxl=vxl, ste_binary=ste)
    got = _fwd_gpu(cuda, x, emb, offs, resl, len(res), vxl=vxl, ste=ste, vbits=vbits)
    assert np.array_equal(got, want)


@VBITS
def test_forward_level_window_slice_and_per_point_levels(cuda, oracle, vbits):
    offs, resl, emb = make_grid(RES3, 10, 3, 4, seed=7)
    vxl = ball_occupancy(16, 3)
    x = _points(1777, 3, seed=8)
    # scalar window: the reference slices the tables in Python (ngp.py:90-91)
    lo, hi = 2, 5
    want = oracle.grid_encode_forward(x, emb, offs[lo:hi + 1], resl[lo:hi], binary_vxl=vxl)
    got = _fwd_gpu(cuda, x, emb, offs[lo:hi + 1].copy(), resl[lo:hi].copy(), hi - lo, vxl=vxl, vbits=vbits)
    assert np.array_equal(got, want)
    # per-point window (forward_diff_levels, ngp.py:265-297)
    rng = np.random.default_rng(1)
    mli = rng.integers(0, len(RES3) - 3 + 1, size=x.shape[0]).astype(np.int32)
    want = oracle.grid_encode_forward(x, emb, offs, resl, n_levels_calc=3, binary_vxl=vxl, min_level_id=mli)
    got = _fwd_gpu(cuda, x, emb, offs, resl, 3, vxl=vxl, mli=mli, vbits=vbits)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("D", [2, 3])
def test_vertex_bits_equal_the_box_test_of_every_vertex(cuda, oracle, D):
    """cnc_grid_vertex_bits against the oracle: a vertex q of level R sits exactly on the point (q - 0.5) / (R - 2)
    of that level, where corner 0 has weight 1 — encoding a table of ones there returns 1 iff the oracle's
    per-corner box test (gridencoder.cu:221-276) passes for q.  Also: a level above the size cap gets no plane."""
    from cnc_amd.backends import gridencoder_backend as be
    res = [5, 9, 14] if D == 3 else [6, 19, 40]
    Rb = 16 if D == 3 else 32
    vxl = (np.random.default_rng(D).uniform(size=(Rb,) * D) < 0.02).astype(np.uint8)    # sparse: both answers occur
    v = torch.as_tensor(vxl, device=cuda).to(torch.bool)
    sat = be.occupancy_sat(v)
    words, offs = be.occupancy_vertex_bits(v, sat, res, max_vertices=res[1] ** D)
    offs = offs.cpu().numpy()
    assert offs[0] == 0 and offs[1] > 0 and offs[2] == -1
    words = words.cpu().numpy().view(np.uint32)
    for l in (0, 1):
        R = res[l]
        q = np.stack(np.meshgrid(*[np.arange(1, R - 1)] * D, indexing="ij"), -1).reshape(-1, D)   # inner vertices
        x = ((q.astype(np.float32) - np.float32(0.5)) / np.float32(R - 2)).astype(np.float32)
        offs_l = np.array([0, R ** D + 8], dtype=np.int32)
        emb = np.ones((R ** D + 8, 1), dtype=np.float32)
        want = oracle.grid_encode_forward(x, emb, offs_l, np.array([R], dtype=np.int32), binary_vxl=vxl)[0, :, 0]
        idx = sum(q[:, d].astype(np.int64) * R ** d for d in range(D))
        got = (words[offs[l] + idx // 32] >> (idx % 32).astype(np.uint32)) & 1
        # with a single valid corner of weight 1 the renormalised output is 1; 0 when the corner is masked out
        # (other corners have weight 0: they add nothing either way)
        assert np.array_equal(got.astype(np.float32), (want > 0.5).astype(np.float32))
        if l == 1:
            assert got.min() == 0 and got.max() == 1


@pytest.mark.parametrize("D,R", [(2, 1026), (3, 5), (2, 9), (3, 22), (1, 70)])
def test_vertex_bits_stay_inside_an_exact_size_plane(cuda, D, R):
    """A C caller sizes the plane with cnc_grid_vertex_bits_words and nothing more: the waves of the last block that
    lie past the last vertex must not store (R^D mod 256 in (0, 192] is the case that used to spill up to six zero
    words; 1026^2, the default finest 2-D level, is one of them)."""
    from cnc_amd import _lib
    from cnc_amd.backends import gridencoder_backend as be
    L = _lib.lib()
    n = R ** D
    assert 0 < n % 256 <= 192
    Rb = 16
    v = torch.ones((Rb,) * D, dtype=torch.bool, device=cuda)
    sat = be.occupancy_sat(v)
    nw = int(L.cnc_grid_vertex_bits_words(D, R))
    assert nw == (n + 63) // 64 * 2
    CANARY = 0x5A5A5A5A
    buf = torch.full((nw + 16,), CANARY, dtype=torch.int32, device=cuda)
    be.check(L.cnc_grid_vertex_bits(be.ptr(sat), D, Rb, R, buf.data_ptr(), be.stream(cuda)), "grid_vertex_bits")
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    assert np.all(got[nw:] == CANARY), "cnc_grid_vertex_bits wrote past the plane"
    words = got[:nw].view(np.uint32)
    idx = np.arange(n)
    bits = (words[idx // 32] >> (idx % 32).astype(np.uint32)) & 1
    assert bits.max() == 1                                 # a full occupancy grid: inner vertices pass the box test
    pad = np.arange(n, nw * 32)
    assert np.all(((words[pad // 32] >> (pad % 32).astype(np.uint32)) & 1) == 0)   # padding bits are written zero


@pytest.mark.parametrize("N", [0, 1, 63, 64, 65, 257])
def test_forward_ragged_sizes(cuda, oracle, N):
    offs, resl, emb = make_grid(RES3, 10, 3, 8, seed=2)
    x = np.random.default_rng(N).uniform(0, 1, size=(N, 3)).astype(np.float32)
    got = _fwd_gpu(cuda, x, emb, offs, resl, len(RES3))
    if N == 0:
        assert got.shape == (len(RES3), 0, 8)
        return
    assert np.array_equal(got, oracle.grid_encode_forward(x, emb, offs, resl))


def _check_bwd(got, want32, acc64, abs64, n_terms_max):
    # |float-sum in any order - exact| <= (n-1) * eps * sum|terms|  (+ rounding of the terms)
    eps = np.finfo(np.float32).eps
    bound = (n_terms_max + 2) * eps * abs64 + 1e-30
    assert np.all(np.abs(got.astype(np.float64) - acc64) <= bound)
    # and the oracle's own fp32 serial sum obeys the same bound (sanity of the bound)
    assert np.all(np.abs(want32.astype(np.float64) - acc64) <= bound)
    untouched = abs64 == 0
    assert np.all(got[untouched] == 0)


@ROUTES
@pytest.mark.parametrize("D,F", [(3, 8), (3, 2), (3, 1), (2, 8), (2, 4), (3, 16)])
@pytest.mark.parametrize("ste", [False, True])
def test_backward_against_float64_shadow(cuda, oracle, D, F, ste, route):
    res = RES3 if D == 3 else RES2
    offs, resl, emb = make_grid(res, 10, D, F, seed=11)
    x = _points(2049, D, seed=12)
    rng = np.random.default_rng(13)
    g = rng.normal(size=(len(res), x.shape[0], F)).astype(np.float32)
    want32, acc64 = oracle.grid_encode_backward(g, x, emb, offs, resl, ste_binary=ste, want_acc64=True)
    _, abs64 = oracle.grid_encode_backward(np.abs(g), x, emb, offs, resl, ste_binary=ste, want_acc64=True)
    got = _bwd_gpu(cuda, g, x, emb, offs, resl, ste=ste, route=route)      # (F = 1 / 16 ignore the flag: the run kernel)
    _check_bwd(got, want32, acc64, abs64, n_terms_max=x.shape[0] * 8)
    if ste:
        assert np.all(got[np.abs(emb) > 1] == 0)


@ROUTES
@VBITS
def test_backward_with_mask_and_per_point_levels(cuda, oracle, vbits, route):
    offs, resl, emb = make_grid(RES3, 10, 3, 8, seed=21)
    vxl = ball_occupancy(16, 3)
    x = _points(1500, 3, seed=22)
    rng = np.random.default_rng(23)
    mli = rng.integers(0, len(RES3) - 3 + 1, size=x.shape[0]).astype(np.int32)
    g = rng.normal(size=(3, x.shape[0], 8)).astype(np.float32)
    want32, acc64 = oracle.grid_encode_backward(g, x, emb, offs, resl, binary_vxl=vxl, min_level_id=mli, want_acc64=True)
    _, abs64 = oracle.grid_encode_backward(np.abs(g), x, emb, offs, resl, binary_vxl=vxl, min_level_id=mli, want_acc64=True)
    got = _bwd_gpu(cuda, g, x, emb, offs, resl, vxl=vxl, mli=mli, vbits=vbits, route=route)
    _check_bwd(got, want32, acc64, abs64, n_terms_max=x.shape[0] * 8)


@pytest.mark.parametrize("route", ["cells", "cells+carry"])
@pytest.mark.parametrize("D,F,masked", [(3, 8, True), (3, 8, False), (2, 8, True), (3, 2, True), (2, 4, False)])
def test_cell_merging_backward_on_lattice_vertices_in_hash_order(cuda, oracle, D, F, masked, route):
    """What the context pass hands the scatter: the lattice vertices of a fine level in hash-slot order, encoded at the
    coarser levels below it (per-point windows for the volume) — many vertices per cell, x-neighbours a few entries apart,
    some gradient rows all zero (levels outside a vertex's window), 2.5 blocks of points so that cells repeat across
    blocks and the last block is ragged.  Dense and hashed levels are both in the windows."""
    res = [6, 9, 14, 20, 31, 44] if D == 3 else [10, 18, 34, 66]
    log2T = 10 if D == 3 else 9
    offs, resl, emb = make_grid(res, log2T, D, F, seed=41)
    emb[::7] *= 3.0                                         # some parameters outside [-1, 1]: the STE mask is live
    vxl = ball_occupancy(16, D) if masked else None
    rng = np.random.default_rng(42)
    n = len(res) - 1                                        # the vertices of the finest level ...
    R = res[n]
    grid = np.stack(np.meshgrid(*[np.arange(1, R - 1)] * D, indexing="ij"), -1).reshape(-1, D)
    primes = np.array([1, 2654435761, 805459861], np.uint64)[:D]
    h = np.bitwise_xor.reduce((grid.astype(np.uint64) * primes) & np.uint64(0xFFFFFFFF), axis=1) % np.uint64(1 << log2T)
    grid = grid[np.argsort(h, kind="stable")][:2600]        # ... in hash-slot order (x and x ^ 1 in neighbouring slots)
    x = ((grid.astype(np.float32) - np.float32(0.5)) / np.float32(R - 2)).astype(np.float32)
    L = 3
    mli = g_levels = None
    if D == 3:
        mli = rng.integers(0, n - L + 1, size=x.shape[0]).astype(np.int32)
        mli[: x.shape[0] // 2] = n - L                      # half of them right below their own level, as the windows are
    else:
        g_levels = slice(n - L, n)
    g = rng.normal(size=(L, x.shape[0], F)).astype(np.float32)
    g[:, rng.random(x.shape[0]) < 0.2] = 0                   # whole points without gradient
    g[0, rng.random(x.shape[0]) < 0.3] = 0                   # and single level slots
    o = offs if D == 3 else offs[n - L:n + 1]
    r = resl if D == 3 else resl[n - L:n]
    kw = dict(binary_vxl=vxl, min_level_id=mli, ste_binary=True, want_acc64=True)
    want32, acc64 = oracle.grid_encode_backward(g, x, emb, o, r, **kw)
    _, abs64 = oracle.grid_encode_backward(np.abs(g), x, emb, o, r, **kw)
    got = _bwd_gpu(cuda, g, x, emb, o, r, vxl=vxl, mli=mli, ste=True, vbits=masked, route=route)
    _check_bwd(got, want32, acc64, abs64, n_terms_max=x.shape[0] * (1 << D))
    assert np.all(got[np.abs(emb) > 1] == 0)
    assert np.abs(got).sum() > 0


def test_backward_is_exact_without_collisions(cuda, oracle):
    """One point per call -> every table entry receives at most one term per (level): the atomic
    result must equal the oracle bit for bit."""
    offs, resl, emb = make_grid(RES3, 12, 3, 8, seed=31)
    x = np.array([[0.3137, 0.7211, 0.5523]], np.float32)
    g = np.random.default_rng(32).normal(size=(len(RES3), 1, 8)).astype(np.float32)
    want = oracle.grid_encode_backward(g, x, emb, offs, resl)
    got = _bwd_gpu(cuda, g, x, emb, offs, resl)
    assert np.array_equal(got, want)


def test_linearity_and_partition_of_unity_full_size(cuda):
    """BASELINE size (16 levels x 2^19 x F8, 2^20 points): properties that need no oracle.
    (a) with an all-ones table every in-range point encodes to 1 (weights renormalised to 1);
    (b) backward of an all-ones gradient deposits exactly N*L*F in total;
    (c) encode is linear in the table."""
    from cnc_amd.backends import gridencoder_backend as be
    dev = cuda
    res_list = [18, 24, 32, 44, 60, 82, 113, 155, 214, 296, 408, 563, 778, 1074, 1484, 2049]
    offs, resl, _ = make_grid(res_list, 19, 3, 1, seed=0)
    F, L, N = 8, 16, 1 << 20
    rows = int(offs[-1])
    assert rows == 6120776
    gen = torch.Generator(device=dev).manual_seed(42)
    x = torch.rand((N, 3), device=dev, generator=gen)
    o_t, r_t = torch.as_tensor(offs, device=dev), torch.as_tensor(resl, device=dev)
    ones = torch.ones((rows, F), device=dev)
    out = torch.empty((L, N, F), device=dev)
    be.grid_encode_forward(x, ones, o_t, r_t, out, N, 3, F, L, 0, 128, 0.0, None, None, None)
    assert torch.all((out - 1).abs() <= 1e-6)
    ge = torch.zeros((rows, F), device=dev)
    be.grid_encode_backward(torch.ones_like(out), x, ones, o_t, r_t, ge, N, 3, F, L, 0, 128, None, None, None, None)
    total = ge.double().sum().item()
    assert abs(total - N * L * F) <= 1e-5 * N * L * F
    a = torch.randn((rows, F), device=dev, generator=gen)
    b = torch.randn((rows, F), device=dev, generator=gen)
    oa, ob, oab = (torch.empty_like(out) for _ in range(3))
    be.grid_encode_forward(x, a, o_t, r_t, oa, N, 3, F, L, 0, 128, 0.0, None, None, None)
    be.grid_encode_forward(x, b, o_t, r_t, ob, N, 3, F, L, 0, 128, 0.0, None, None, None)
    be.grid_encode_forward(x, a + b, o_t, r_t, oab, N, 3, F, L, 0, 128, 0.0, None, None, None)
    assert torch.allclose(oab, oa + ob, atol=2e-5, rtol=0)


def test_error_behaviour(cuda):
    from cnc_amd.backends import gridencoder_backend as be
    dev = cuda
    offs, resl, emb = make_grid(RES3, 10, 3, 8)
    x = torch.rand((8, 3), device=dev)
    e = torch.as_tensor(emb, device=dev)
    o, r = torch.as_tensor(offs, device=dev), torch.as_tensor(resl, device=dev)
    out = torch.empty((len(RES3), 8, 8), device=dev)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        be.grid_encode_forward(x.cpu(), e, o, r, out, 8, 3, 8, len(RES3), 0, 128, 0.0, None, None, None)
    with pytest.raises(RuntimeError, match="contiguous"):
        be.grid_encode_forward(x.t().contiguous().t(), e, o, r, out, 8, 3, 8, len(RES3), 0, 128, 0.0, None, None, None)
    with pytest.raises(RuntimeError, match="int tensor"):
        be.grid_encode_forward(x, e, o.long(), r, out, 8, 3, 8, len(RES3), 0, 128, 0.0, None, None, None)
    with pytest.raises(RuntimeError, match="n_fearures"):
        be.grid_encode_forward(x, e, o, r, out, 8, 3, 3, len(RES3), 0, 128, 0.0, None, None, None)


@pytest.mark.parametrize("axis", [0, 1, 2])
@pytest.mark.parametrize("F", [2, 8])
def test_cnt_np_embed_forward_backward(cuda, oracle, axis, F):
    from cnc_amd.backends import gridencoder_backend as be
    dev = cuda
    R, hs = 34, 2 ** 12
    rng = np.random.default_rng(axis * 7 + F)
    pts = rng.integers(0, R, size=(5000, 3)).astype(np.int16)   # includes border vertices (skipped)
    emb = np.where(rng.uniform(size=(hs, F)) < 0.5, 1.0, -1.0).astype(np.float32)
    emb[::7] *= 0.5   # values that are not +-1: 0.5 counts as "negative" (> 0.9 test)
    want = oracle.cnt_np_embed(pts, emb, R, hs, axis)
    out = torch.zeros((R - 2, R - 2, F, 2), device=dev)
    be.cnt_np_embed(torch.as_tensor(pts, device=dev), torch.as_tensor(emb, device=dev), out,
                    pts.shape[0], R, F, hs, axis)
    got = out.cpu().numpy()
    assert np.array_equal(got, want)            # integer-valued counts: exact in fp32
    assert got.sum() == want.sum()
    s = (want.sum(-1, keepdims=True) + 1e-6).astype(np.float32)
    grad = rng.normal(size=want.shape).astype(np.float32)
    want_g, acc = oracle.cnt_np_embed_backward(pts, emb, s, grad, R, hs, axis, want_acc64=True)
    _, absacc = oracle.cnt_np_embed_backward(pts, emb, s, np.abs(grad), R, hs, axis, want_acc64=True)
    ge = torch.zeros((hs, F), device=dev)
    be.cnt_np_embed_backward(torch.as_tensor(pts, device=dev), torch.as_tensor(emb, device=dev),
                             torch.as_tensor(s, device=dev), torch.as_tensor(grad, device=dev), ge,
                             pts.shape[0], R, F, hs, axis)
    got_g = ge.cpu().numpy()
    bound = 64 * np.finfo(np.float32).eps * np.abs(absacc) + 1e-30
    assert np.all(np.abs(got_g.astype(np.float64) - acc) <= bound)


@pytest.mark.parametrize("D", [2, 3])
@pytest.mark.parametrize("F", [1, 2, 4, 8, 16, 32])
def test_bit_plane_forward_equals_fp32_ste_forward(cuda, oracle, D, F):
    """cnc_pack_sign_bits + cnc_grid_encode_forward_bits == fp32 gather with the STE flag == oracle."""
    from cnc_amd.backends import gridencoder_backend as be
    res = RES3 if D == 3 else RES2
    offs, resl, emb = make_grid(res, 10, D, F, seed=40 + F)
    emb[::5] = 0.0          # sign(0) = +1
    emb[1::5] = -0.0
    vxl = ball_occupancy(16 if D == 3 else 32, D)
    x = _points(2111, D, seed=41)
    rng = np.random.default_rng(42)
    mli = rng.integers(0, len(res) - 2, size=x.shape[0]).astype(np.int32)
    t = lambda a: None if a is None else torch.as_tensor(a, device=cuda)
    bits = be.pack_sign_bits(t(emb))
    want_bits = np.packbits((emb >= 0).reshape(-1), bitorder="little")
    assert np.array_equal(bits.cpu().numpy(), want_bits)
    for kw in (dict(), dict(vxl=vxl), dict(vxl=vxl, mli=mli, L=2)):
        L = kw.get("L", len(res))
        want = oracle.grid_encode_forward(x, emb, offs, resl, n_levels_calc=L, binary_vxl=kw.get("vxl"),
                                          min_level_id=kw.get("mli"), ste_binary=True)
        out = torch.empty((L, x.shape[0], F), device=cuda)
        be.grid_encode_forward_bits(t(x), bits, t(offs), t(resl), out, x.shape[0], D, F, L,
                                    128 if "vxl" not in kw else vxl.shape[-1], t(kw.get("vxl")), t(kw.get("mli")))
        assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("N", [1, 63, 64, 65, 255, 256, 257, 1000])
@pytest.mark.parametrize("D,F,L", [(3, 8, 6), (3, 8, 5), (3, 8, 1), (3, 2, 6), (3, 2, 3), (3, 4, 5), (2, 8, 4), (2, 8, 3),
                                   (3, 16, 3), (3, 32, 2), (3, 1, 6)])
def test_bit_plane_forward_layouts_and_ragged_sizes(cuda, oracle, N, D, F, L):
    """The wave-transposed stores of k_grid_encode_fwd_bits (round 4): level-major [L, N, F] and point-major rows inside
    a wider [N, ld] matrix (out_ld / out_col), every point count around the wave and block sizes, level counts that
    leave a short last pass (which falls back to per-lane stores when its piece is not whole 16-byte chunks) — all
    bit-equal to the oracle, nothing written outside the encoder's columns."""
    from cnc_amd.backends import gridencoder_backend as be
    res = (RES3 if D == 3 else RES2)[:L]
    offs, resl, emb = make_grid(res, 10, D, F, seed=60 + F)
    x = _points(N, D, seed=61 + N)
    t = lambda a: torch.as_tensor(a, device=cuda)
    bits = be.pack_sign_bits(t(emb))
    want = oracle.grid_encode_forward(x, emb, offs, resl, ste_binary=True)          # [L, N, F]
    out = torch.full((L, N, F), 7.0, device=cuda)
    be.grid_encode_forward_bits(t(x), bits, t(offs), t(resl), out, N, D, F, L, 128)
    assert np.array_equal(out.cpu().numpy(), want)
    for ld, col in ((L * F, 0), (L * F + 12, 4), (L * F + 8, 8)):
        feat = torch.full((N, ld), 7.0, device=cuda)
        be.grid_encode_forward_bits(t(x), bits, t(offs), t(resl), feat, N, D, F, L, 128, None, None, None,
                                    out_ld=ld, out_col=col)
        got = feat.cpu().numpy()
        assert np.array_equal(got[:, col:col + L * F].reshape(N, L, F).transpose(1, 0, 2), want)
        assert np.all(got[:, :col] == 7.0) and np.all(got[:, col + L * F:] == 7.0)     # the neighbours' columns


def test_gridencoder_bit_plane_cache_tracks_in_place_updates(cuda):
    from cnc_amd.gridencoder import GridEncoder
    a = GridEncoder(3, 8, RES3, 10, ste_binary=True, bitplane=True).to(cuda)
    b = GridEncoder(3, 8, RES3, 10, ste_binary=True, bitplane=False).to(cuda)
    with torch.no_grad():
        a.params.uniform_(-1, 1)
        b.params.copy_(a.params)
    x = torch.rand(3000, 3, device=cuda)
    assert torch.equal(a(x), b(x))
    bits0, key0, plane0 = a._bits, a._bits_key, a._bits.clone()
    assert torch.equal(a(x), b(x)) and a._bits is bits0 and a._bits_key == key0          # cached
    with torch.no_grad():
        a.params.mul_(-1.0)                                       # optimizer-style in-place update
        b.params.mul_(-1.0)
    # repacked — in place: the plane's address is a constant of the run (captured graphs read it, cnc_amd/_planes_graph.py)
    assert torch.equal(a(x), b(x)) and a._bits_key != key0 and a._bits.data_ptr() == bits0.data_ptr()
    assert not torch.equal(a._bits, plane0)
    w = torch.randn(3000, 8 * len(RES3), device=cuda)
    (a(x) * w).sum().backward()
    (b(x) * w).sum().backward()
    assert (a.params.grad - b.params.grad).abs().max() <= 1e-5 * b.params.grad.abs().max()


def test_gridencoder_caches_survive_reset_and_dot_data_writes(cuda):
    """`reset_parameters()` and writes through `.data` (which do not bump `_version`) followed by
    `invalidate_caches()` must not leave a stale sign plane behind; the plane attributes never register as
    module parameters / buffers."""
    from cnc_amd.gridencoder import GridEncoder
    torch.manual_seed(1)
    a = GridEncoder(3, 8, RES3, 10, ste_binary=True, bitplane=True).to(cuda)
    ref = GridEncoder(3, 8, RES3, 10, ste_binary=True, bitplane=False).to(cuda)
    x = torch.rand(2000, 3, device=cuda)
    y0 = a(x).clone()
    assert [n for n, _ in a.named_parameters()] == ["params"]            # _bits_src is not a parameter
    a.reset_parameters()
    with torch.no_grad():
        ref.params.copy_(a.params)
    y1 = a(x)
    assert torch.equal(y1, ref(x)) and not torch.equal(y1, y0)
    a.params.data.mul_(-1.0)                 # no version bump
    a.invalidate_caches()
    with torch.no_grad():
        ref.params.mul_(-1.0)
    assert torch.equal(a(x), ref(x))
    other = torch.randn_like(a.params)       # outspace_params of another tensor, then back
    with torch.no_grad():
        ref.params.copy_(other)
    assert torch.equal(a(x, outspace_params=other), ref(x))
    assert [n for n, _ in a.named_parameters()] == ["params"]


@pytest.mark.parametrize("outliers", [False, True])
def test_backward_ste_clip_count_hint(cuda, oracle, outliers):
    """pack_sign_bits counts |v| > 1; backward skips the STE-mask gather iff the count is 0 and
    gives the oracle's masked gradient either way."""
    from cnc_amd.backends import gridencoder_backend as be
    offs, resl, emb = make_grid(RES3, 10, 3, 8, seed=77)
    emb = np.clip(emb, -1, 1)
    if outliers:
        emb[::11] *= 3.0
    n_clip = int((np.abs(emb) > 1).sum())
    assert (n_clip > 0) == outliers
    x = _points(1500, 3, seed=78)
    g = np.random.default_rng(79).normal(size=(len(RES3), x.shape[0], 8)).astype(np.float32)
    t = lambda a: torch.as_tensor(a, device=cuda)
    cc = torch.full((1,), 123, dtype=torch.int32, device=cuda)
    be.pack_sign_bits(t(emb), None, cc)
    assert int(cc.item()) == n_clip
    ge = torch.zeros(emb.shape, device=cuda)
    be.grid_encode_backward(t(g), t(x), t(emb), t(offs), t(resl), ge, x.shape[0], 3, 8, len(RES3), 0, 128,
                            None, None, None, None, ste_binary=True, ste_clip_count=cc)
    want32, acc64 = oracle.grid_encode_backward(g, x, emb, offs, resl, ste_binary=True, want_acc64=True)
    _, abs64 = oracle.grid_encode_backward(np.abs(g), x, emb, offs, resl, ste_binary=True, want_acc64=True)
    _check_bwd(ge.cpu().numpy(), want32, acc64, abs64, n_terms_max=x.shape[0] * 8)
    if outliers:
        assert np.all(ge.cpu().numpy()[np.abs(emb) > 1] == 0)


@pytest.mark.parametrize("D,Rb", [(3, 16), (3, 128), (2, 32), (2, 128)])
def test_occupancy_sat_gives_the_scans_answer(cuda, oracle, D, Rb):
    """Masked encode with the summed-volume table == masked encode with the box scan == oracle
    (forward bit-exact for fp32 / STE / bit-plane kernels; backward within the bound)."""
    from cnc_amd.backends import gridencoder_backend as be
    res = RES3 if D == 3 else RES2
    offs, resl, emb = make_grid(res, 10, D, 8, seed=90)
    vxl = ball_occupancy(Rb, D, radius=0.3, seed=Rb)
    x = _points(3000, D, seed=91)
    t = lambda a: None if a is None else torch.as_tensor(a, device=cuda)
    sat = be.occupancy_sat(t(vxl))
    ref = np.zeros([n + 1 for n in vxl.shape], np.int64)
    c = vxl.astype(np.int64)
    for d in range(D):
        c = np.cumsum(c, axis=d)
    ref[tuple(slice(1, None) for _ in range(D))] = c
    assert np.array_equal(sat.cpu().numpy(), ref)
    L = len(res)
    want = oracle.grid_encode_forward(x, emb, offs, resl, binary_vxl=vxl)
    out = torch.empty((L, x.shape[0], 8), device=cuda)
    be.grid_encode_forward(t(x), t(emb), t(offs), t(resl), out, x.shape[0], D, 8, L, 0, Rb, 0.0, None, t(vxl), None, occ_sat=sat)
    assert np.array_equal(out.cpu().numpy(), want)
    want_s = oracle.grid_encode_forward(x, emb, offs, resl, binary_vxl=vxl, ste_binary=True)
    bits = be.pack_sign_bits(t(emb))
    be.grid_encode_forward_bits(t(x), bits, t(offs), t(resl), out, x.shape[0], D, 8, L, Rb, t(vxl), None, sat)
    assert np.array_equal(out.cpu().numpy(), want_s)
    g = np.random.default_rng(92).normal(size=(L, x.shape[0], 8)).astype(np.float32)
    want32, acc64 = oracle.grid_encode_backward(g, x, emb, offs, resl, binary_vxl=vxl, want_acc64=True)
    _, abs64 = oracle.grid_encode_backward(np.abs(g), x, emb, offs, resl, binary_vxl=vxl, want_acc64=True)
    ge = torch.zeros(emb.shape, device=cuda)
    be.grid_encode_backward(t(g), t(x), t(emb), t(offs), t(resl), ge, x.shape[0], D, 8, L, 0, Rb, None, None, t(vxl), None, occ_sat=sat)
    _check_bwd(ge.cpu().numpy(), want32, acc64, abs64, n_terms_max=x.shape[0] * 8)


@pytest.mark.parametrize("axis", [0, 1, 2])
@pytest.mark.parametrize("F,R,log2T", [(8, 34, 12), (2, 34, 12), (4, 12, 12), (16, 20, 10)])
def test_cnt_np_embed_planned_equals_oracle(cuda, oracle, axis, F, R, log2T):
    """The sorted-plan vote kernels (no atomics): counts equal the oracle's exactly, gradients within
    the float64-shadow bound; duplicates, border vertices and a dense finest level (R^3 < T) included."""
    from cnc_amd.backends import gridencoder_backend as be
    hs = 2 ** log2T
    rng = np.random.default_rng(axis * 11 + F)
    pts = rng.integers(0, R, size=(6000, 3)).astype(np.int16)
    rows = min(hs, R ** 3)
    emb = np.where(rng.uniform(size=(hs, F)) < 0.5, 1.0, -1.0).astype(np.float32)
    emb[::5] *= 0.5
    t = lambda a: torch.as_tensor(a, device=cuda)
    plan = be.VotePlan(t(pts), R, hs)
    want = oracle.cnt_np_embed(pts, emb, R, hs, axis)
    out = torch.full((R - 2, R - 2, F, 2), -7.0, device=cuda)        # must be overwritten everywhere
    be.cnt_np_embed_planned(plan, t(emb[:rows].copy()), out, F, axis)
    assert np.array_equal(out.cpu().numpy(), want)
    s = (want.sum(-1, keepdims=True) + 1e-6).astype(np.float32)
    grad = rng.normal(size=want.shape).astype(np.float32)
    _, acc = oracle.cnt_np_embed_backward(pts, emb, s, grad, R, hs, axis, want_acc64=True)
    _, absacc = oracle.cnt_np_embed_backward(pts, emb, s, np.abs(grad), R, hs, axis, want_acc64=True)
    ge = torch.zeros((rows, F), device=cuda)
    g_over_sum = (torch.reciprocal(t(s)) * t(grad)).contiguous()
    be.cnt_np_embed_planned_backward(plan, t(emb[:rows].copy()), g_over_sum, ge, F, axis)
    got = ge.cpu().numpy().astype(np.float64)
    bound = 64 * np.finfo(np.float32).eps * np.abs(absacc[:rows]) + 1e-30
    assert np.all(np.abs(got - acc[:rows]) <= bound)
    assert np.all(acc[rows:] == 0)


def test_baseline_config0_against_oracle_and_torch_cpu_fallback(cuda, oracle):
    """BASELINE.json configs[0]: 16 levels, log2T=19, F=2, one batch of 4096 rays — the case the
    "PyTorch-CPU gridencoder fallback" (oracle/torch_cpu_encoder.py) runs.  HIP forward: bit-exact
    against both CPU paths; HIP backward (binned + atomic levels, as GridEncoder routes it): within
    the float64-shadow bound."""
    from cnc_amd.backends import gridencoder_backend as be
    from cnc_amd.nerfacc.estimators.occ_grid import OccGridEstimator
    from cnc_amd.synthetic import RES_16L, ball_binaries, level_offsets, pinhole_rays
    from oracle import torch_cpu_encoder as tce
    dev, F, L = cuda, 2, 16
    offs = level_offsets(RES_16L, 19, 3)
    aabb = torch.tensor([-1.5] * 3 + [1.5] * 3, device=dev)
    est = OccGridEstimator(roi_aabb=aabb, resolution=128, levels=1).to(dev)
    est.binaries = ball_binaries(128, device=dev)
    ro, rd = pinhole_rays(device=dev)
    pick = torch.randperm(640000, generator=torch.Generator().manual_seed(3))[:4096].to(dev)
    ro, rd = ro.reshape(-1, 3)[pick].contiguous(), rd.reshape(-1, 3)[pick].contiguous()
    ri, ts, te = est.sampling(ro, rd, render_step_size=5e-3, stratified=False)
    p = ro[ri] + rd[ri] * ((ts + te) * 0.5)[:, None]
    x = ((p - aabb[:3]) / (aabb[3:] - aabb[:3])).clamp(0, 1).contiguous()
    N = x.shape[0]
    assert N > 100000
    rng = np.random.default_rng(11)
    emb = ((rng.random((int(offs[-1]), F), dtype=np.float32) * 2 - 1) * 1e-4)   # ngp.py:221-223 init
    g = rng.standard_normal((L, N, F)).astype(np.float32)
    o_t, r_t = torch.as_tensor(offs, device=dev), torch.tensor(RES_16L, dtype=torch.int32, device=dev)
    emb_t, g_t = torch.as_tensor(emb, device=dev), torch.as_tensor(g, device=dev)
    out = torch.empty((L, N, F), dtype=torch.float32, device=dev)
    be.grid_encode_forward(x, emb_t, o_t, r_t, out, N, 3, F, L, 0, 128, 0.0, None, None, None, ste_binary=True)
    ge = torch.zeros_like(emb_t)
    plan = be.plan_binned_levels(RES_16L, [int(v) for v in offs], 3, F, N)      # as GridEncoder routes this size
    be.grid_encode_backward(g_t, x, emb_t, o_t, r_t, ge, N, 3, F, L, 0, 128, None, None, None, None,
                            ste_binary=True, binned=plan)
    torch.cuda.synchronize()
    xn = x.cpu().numpy()
    want = oracle.grid_encode_forward(xn, emb, offs, RES_16L, ste_binary=True, threads=8)
    assert np.array_equal(out.cpu().numpy(), want)
    thr = oracle.max_threads()
    want_g, acc64 = oracle.grid_encode_backward(g, xn, emb, offs, RES_16L, ste_binary=True, want_acc64=True, threads=thr)
    _, abs64 = oracle.grid_encode_backward(np.abs(g), xn, emb, offs, RES_16L, ste_binary=True, want_acc64=True, threads=thr)
    # per-entry float32 summation bound with the entry's own term count (independent NumPy restatement)
    import np_twins as tw
    n_e = tw.grid_entry_counts(xn, offs, RES_16L)
    err = np.abs(ge.cpu().numpy().astype(np.float64) - acc64)
    assert np.all(err <= (n_e[:, None] + 2) * np.finfo(np.float32).eps * abs64 + 1e-30)
    assert err.max() <= 1e-5 * np.abs(acc64).max()
    # the torch-CPU fallback on a slice of the batch (it is ~10^4 samples/s)
    n_t = 20000
    out_t, g_t_cpu = tce.forward_backward(torch.from_numpy(xn[:n_t]), torch.from_numpy(emb), offs, RES_16L,
                                          torch.from_numpy(np.ascontiguousarray(g[:, :n_t])), ste_binary=True)
    assert np.array_equal(out_t.numpy(), want[:, :n_t])
    ge2 = torch.zeros_like(emb_t)
    be.grid_encode_backward(g_t[:, :n_t].contiguous(), x[:n_t].contiguous(), emb_t, o_t, r_t, ge2, n_t, 3, F, L,
                            0, 128, None, None, None, None, ste_binary=True)
    torch.cuda.synchronize()
    ref = g_t_cpu.numpy()
    assert np.abs(ge2.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("D,F", [(3, 8), (3, 2), (2, 4), (2, 16)])
@pytest.mark.parametrize("ste", [False, True])
def test_dy_dx_and_input_backward_bit_exact(cuda, oracle, D, F, ste):
    """kernel_grid's dy_dx branch and kernel_input_backward (gridencoder.cu:319-395, :588-614): dead in
    CNC but part of the `_gridencoder` interface.  Same operation order as the oracle -> bit-exact,
    including out-of-range points, the border ring and a per-point level window."""
    from cnc_amd.backends import gridencoder_backend as be
    res = RES3 if D == 3 else RES2
    offs, resl, emb = make_grid(res, 10, D, F, seed=70 + F)
    x = _points(777, D, seed=71)
    N, L = x.shape[0], len(res)
    t = lambda a: torch.as_tensor(a, device=cuda)
    out = torch.empty((L, N, F), device=cuda)
    dy = torch.full((N, L, D, F), 7.0, device=cuda)
    be.grid_encode_forward(t(x), t(emb), t(offs), t(resl), out, N, D, F, L, 0, 128, 0.0, dy, None, None,
                           ste_binary=ste)
    torch.cuda.synchronize()
    want_dy = oracle.grid_dy_dx(x, emb, offs, resl, ste_binary=ste)
    assert np.array_equal(dy.cpu().numpy(), want_dy)
    assert np.array_equal(out.cpu().numpy(), oracle.grid_encode_forward(x, emb, offs, resl, ste_binary=ste))
    g = np.random.default_rng(72).normal(size=(L, N, F)).astype(np.float32)
    ge, gi = torch.zeros(emb.shape, device=cuda), torch.full((N, D), 3.0, device=cuda)
    be.grid_encode_backward(t(g), t(x), t(emb), t(offs), t(resl), ge, N, D, F, L, 0, 128, dy, gi, None, None,
                            ste_binary=ste)
    torch.cuda.synchronize()
    assert np.array_equal(gi.cpu().numpy(), oracle.input_backward(g, want_dy))
    # the embedding gradient is unaffected by the extra outputs
    ge2 = torch.zeros(emb.shape, device=cuda)
    be.grid_encode_backward(t(g), t(x), t(emb), t(offs), t(resl), ge2, N, D, F, L, 0, 128, None, None, None, None,
                            ste_binary=ste)
    torch.cuda.synchronize()
    assert torch.allclose(ge, ge2, rtol=0, atol=1e-4 * float(ge2.abs().max()))
    # per-point level window (2 of the levels, starting at min_level_id[b])
    mli = np.random.default_rng(73).integers(0, L - 1, size=N).astype(np.int32)
    dy2 = torch.empty((N, 2, D, F), device=cuda)
    out2 = torch.empty((2, N, F), device=cuda)
    be.grid_encode_forward(t(x), t(emb), t(offs), t(resl), out2, N, D, F, 2, 0, 128, 0.0, dy2, None, t(mli),
                           ste_binary=ste)
    torch.cuda.synchronize()
    assert np.array_equal(dy2.cpu().numpy(), oracle.grid_dy_dx(x, emb, offs, resl, n_levels_calc=2,
                                                               min_level_id=mli, ste_binary=ste))
    # dy_dx without grad_inputs (or the reverse) is a caller error
    with pytest.raises(RuntimeError):
        be.grid_encode_backward(t(g), t(x), t(emb), t(offs), t(resl), ge2, N, D, F, L, 0, 128, dy, None, None, None)


@pytest.mark.parametrize("n", [1, 5, 4096, 100003 * 8])
def test_ste_binary_kernels_equal_the_op_chain(cuda, n):
    """cnc_ste_binary_{forward,backward} vs the reference's op chain (ngp.py:22-39) on the same device: +1 / -1
    (0 for NaN), gradient passed where clamp(x, -1, 1) == x — edges at 0, -0, +-1, beyond, inf, NaN."""
    from cnc_amd.gridencoder import STE_binary
    g = torch.Generator(device="cpu").manual_seed(n)
    x = (torch.rand(n, generator=g) * 3 - 1.5)
    edge = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.0000001, -1.0000001, float("inf"), -float("inf"), float("nan")])
    x[: min(n, edge.numel())] = edge[: min(n, edge.numel())]
    x = x.to(cuda).requires_grad_()
    go = torch.randn(n, generator=g).to(cuda)
    y = STE_binary.apply(x)
    y.backward(go)
    xd = x.detach()
    c = torch.clamp(xd, min=-1, max=1)
    want = (c >= 0) * 1.0 + (c < 0) * -1.0
    assert torch.equal(y.detach().nan_to_num(nan=7.0), want.nan_to_num(nan=7.0))
    want_g = go * ((c == xd) + 0.0)
    assert torch.equal(x.grad, want_g)


def test_vertex_bits_at_the_context_pass_size(cuda):
    """Full-size check of the vertex bit planes (no oracle needed: the box scan / summed-volume path is oracle-pinned
    above): the 12-level 3-D grid of the reference composition (R up to 514, T = 2^19, F = 8), a 128^3 ball occupancy
    with speckle, 2^18 grid vertices as points with per-point level windows of 3 — what the context pass encodes.  The
    masked bit-plane forward must return the SAME bits with the planes as with the summed-volume test, the finest
    level (514^3 > 2^26 vertices) must have no plane and fall back, and the backward must agree to atomic order."""
    from cnc_amd import synthetic
    from cnc_amd.backends import gridencoder_backend as be
    res, F, T = list(synthetic.RES_3D_REF), 8, 19
    offs = synthetic.level_offsets(res, T, 3)
    o_t = torch.as_tensor(offs, device=cuda)
    r_t = torch.tensor(res, dtype=torch.int32, device=cuda)
    g = torch.Generator(device="cpu").manual_seed(5)
    emb = torch.sign(torch.rand((int(offs[-1]), F), generator=g) * 2 - 1).to(cuda)
    bits = be.pack_sign_bits(emb)
    vxl = synthetic.ball_binaries(128, radius=1.0, device=cuda).squeeze(0)
    vxl = vxl ^ (torch.rand(vxl.shape, generator=g) < 0.01).to(cuda)
    sat = be.occupancy_sat(vxl)
    words, voff = be.occupancy_vertex_bits(vxl, sat, res)
    assert int(voff[-1]) == -1 and all(int(v) >= 0 for v in voff[:-1])
    N, Lc = 1 << 18, 3
    lvl = torch.randint(Lc, len(res), (N,), generator=g)                       # vertex level n, encoded at n-3 .. n-1
    R = torch.tensor(res)[lvl]
    q = (torch.rand((N, 3), generator=g) * (R[:, None] - 2).float()).floor() + 1     # inner vertices of level n
    x = ((q - 0.5) / (R[:, None] - 2).float()).to(torch.float32).to(cuda).contiguous()
    mli = (lvl - Lc).to(torch.int32).to(cuda)
    outs = []
    for vb in (None, (words, voff)):
        out = torch.empty((N, Lc * F), dtype=torch.float32, device=cuda)
        be.grid_encode_forward_bits(x, bits, o_t, r_t, out, N, 3, F, Lc, 128, vxl, mli, sat, out_ld=Lc * F, out_col=0,
                                    vertex_bits=vb)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert float(outs[0].abs().sum()) > 0 and float((outs[0] == 0).float().mean()) > 0.01      # masked and unmasked corners
    grad = torch.randn((N, Lc * F), generator=g).to(cuda)
    gs = []
    for vb in (None, (words, voff)):
        ge = torch.zeros_like(emb)
        be.grid_encode_backward(grad, x, emb, o_t, r_t, ge, N, 3, F, Lc, 0, 128, None, None, vxl, mli, ste_binary=True,
                                occ_sat=sat, grad_ld=Lc * F, grad_col=0, vertex_bits=vb)
        gs.append(ge)
    scale = float(gs[0].abs().max())
    assert scale > 0 and float((gs[0] - gs[1]).abs().max()) <= 2e-5 * scale
