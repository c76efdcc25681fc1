"""HIP marcher + scans (through the `nerfacc.cuda` mirror) vs the CPU oracle.  Counts, masks,
indices and t-values are bit-exact; the scans keep the reference's 32-wide tile association, so
they are bit-exact too."""
import numpy as np
import pytest
import torch

from conftest import ball_occupancy

pytestmark = pytest.mark.gpu


def _rays(n, seed, radius=4.0):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = (o / np.linalg.norm(o, axis=1, keepdims=True) * radius).astype(np.float32)
    target = rng.uniform(-0.8, 0.8, size=(n, 3))
    d = target - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    if n > 8:
        o[0] = [0, 0, -4]; d[0] = [0, 0, 1]          # axis-aligned (two zero components)
        o[1] = [0.3, -4, 0.1]; d[1] = [0, 1, 0]
        o[2] = [4, 4, 4]; d[2] = [0.57735026, 0.57735026, 0.57735026]   # points away: miss
        o[3] = [0.1, 0.2, 0.3]; d[3] = [1, 0, 0]     # origin inside the box
    return o, d


AABB = np.array([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], np.float32)


def test_ray_aabb_intersect(cuda, oracle):
    from cnc_amd.backends import nerfacc_cuda as nc
    o, d = _rays(5000, 1)
    aabbs = np.concatenate([AABB, AABB * 2, AABB * 0.25], 0)
    for near, far, miss in ((-np.inf, np.inf, np.inf), (0.5, 4.2, -1.0)):
        w0, w1, wh = oracle.ray_aabb_intersect(o, d, aabbs, near, far, miss)
        g0, g1, gh = nc.ray_aabb_intersect(torch.as_tensor(o, device=cuda), torch.as_tensor(d, device=cuda),
                                           torch.as_tensor(aabbs, device=cuda), near, far, miss)
        assert np.array_equal(gh.cpu().numpy(), wh)
        assert np.array_equal(g0.cpu().numpy(), w0)
        assert np.array_equal(g1.cpu().numpy(), w1)
        assert 0 < wh.sum() < wh.size


def _traverse_gpu(dev, o, d, binaries, aabbs, **kw):
    from cnc_amd.nerfacc.grid import traverse_grids
    t = lambda a: None if a is None else torch.as_tensor(a, device=dev)
    kw = {k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
    iv, sm, term = traverse_grids(t(o), t(d), t(binaries), t(aabbs), **kw)
    torch.cuda.synchronize()
    return iv, sm, term


def _cmp_segments(iv, sm, term, oiv, osm, oterm, live=None):
    assert np.array_equal(iv.packed_info[:, 1].cpu().numpy(), oiv["chunk_cnts"])
    assert np.array_equal(iv.packed_info[:, 0].cpu().numpy(), oiv["chunk_starts"])
    assert np.array_equal(sm.packed_info[:, 1].cpu().numpy(), osm["chunk_cnts"])
    assert np.array_equal(iv.is_left.cpu().numpy(), oiv["is_left"])
    assert np.array_equal(iv.is_right.cpu().numpy(), oiv["is_right"])
    # terminate_planes is only written for rays the fill pass visits (grid.cu:100-106,310-311):
    # skipped rays keep whatever torch.empty held, in the reference as here
    if live is None:
        live = osm["chunk_cnts"] > 0
    assert np.array_equal(term.cpu().numpy()[live], oterm[live])


@pytest.mark.parametrize("step,cone", [(5e-3, 0.0), (2e-2, 0.004), (0.0, 0.0)])
def test_traverse_two_pass(cuda, oracle, step, cone):
    occ = ball_occupancy(32, 3, radius=0.33, seed=4)[None]
    o, d = _rays(3000, 7)
    rng = np.random.default_rng(3)
    near = (rng.uniform(size=o.shape[0]) * max(step, 1e-3)).astype(np.float32)   # stratified jitter
    far = np.full(o.shape[0], 1e10, np.float32)
    oiv, osm, oterm = oracle.traverse_grids(o, d, occ, AABB, near, far, step, cone)
    iv, sm, term = _traverse_gpu(cuda, o, d, occ, AABB, near_planes=near, far_planes=far,
                                 step_size=step, cone_angle=cone)
    _cmp_segments(iv, sm, term, oiv, osm, oterm)
    assert np.array_equal(iv.vals.cpu().numpy(), oiv["vals"])
    assert np.array_equal(iv.ray_indices.cpu().numpy(), oiv["ray_indices"])
    assert np.array_equal(sm.vals.cpu().numpy(), osm["vals"])
    assert np.array_equal(sm.ray_indices.cpu().numpy(), osm["ray_indices"])
    assert osm["chunk_cnts"].sum() > 10000 and (osm["chunk_cnts"] == 0).any()
    # sortedness: ordered by ray, then by t
    ri = sm.ray_indices.cpu().numpy()
    assert np.all(np.diff(ri) >= 0)
    v = sm.vals.cpu().numpy()
    same = np.diff(ri) == 0
    assert np.all(np.diff(v)[same] > 0)


def test_traverse_over_allocate_iterative(cuda, oracle):
    """The evaluation loop's form (examples/utils.py:395-478): bounded steps, over-allocation,
    rays_mask, restart from the termination planes until every ray is done; the concatenation
    of all rounds must equal one unbounded march."""
    occ = ball_occupancy(32, 3, radius=0.33, seed=4)[None]
    o, d = _rays(2000, 11)
    n = o.shape[0]
    t0, t1, hits = oracle.ray_aabb_intersect(o, d, AABB)
    t_sorted = np.concatenate([t0, t1], -1)
    t_indices = np.broadcast_to(np.arange(2, dtype=np.int64), (n, 2)).copy()
    near = np.zeros(n, np.float32)
    far = np.full(n, 1e10, np.float32)
    mask = np.ones(n, bool)
    limit = 24
    totals = np.zeros(n, np.int64)
    full_iv, full_sm, _ = oracle.traverse_grids(o, d, occ, AABB, near, far, 5e-3, 0.0)
    for it in range(64):
        if not mask.any():
            break
        oiv, osm, oterm = oracle.traverse_grids(o, d, occ, AABB, near, far, 5e-3, 0.0,
                                                traverse_steps_limit=limit, over_allocate=True,
                                                rays_mask=mask, t_sorted=t_sorted, t_indices=t_indices, hits=hits)
        iv, sm, term = _traverse_gpu(cuda, o, d, occ, AABB, near_planes=near, far_planes=far,
                                     step_size=5e-3, cone_angle=0.0, traverse_steps_limit=limit,
                                     over_allocate=True, rays_mask=mask, t_sorted=t_sorted,
                                     t_indices=t_indices, hits=hits)
        _cmp_segments(iv, sm, term, oiv, osm, oterm, live=mask)
        # over-allocated buffers: compare the values selected by the masks, as the caller does
        gl, gr = iv.vals[iv.is_left].cpu().numpy(), iv.vals[iv.is_right].cpu().numpy()
        assert np.array_equal(gl, oiv["vals"][oiv["is_left"]])
        assert np.array_equal(gr, oiv["vals"][oiv["is_right"]])
        assert np.array_equal(sm.ray_indices[sm.is_valid].cpu().numpy(), osm["ray_indices"][osm["is_valid"]])
        cnt = osm["chunk_cnts"]
        totals += cnt
        near = np.where(mask, oterm, near).astype(np.float32)
        mask = mask & (cnt == limit)
    assert not mask.any()
    # restarting the DDA from a termination plane recomputes the cell-boundary distances from a
    # different origin, so a sample that sits on a boundary may flip: near-equality only
    diff = np.abs(totals - full_sm["chunk_cnts"])
    assert diff.max() <= 2 and (diff > 0).mean() < 0.05


def test_traverse_multi_grid(cuda, oracle):
    rng = np.random.default_rng(2)
    occ = rng.uniform(size=(2, 16, 16, 16)) < 0.3
    aabbs = np.concatenate([AABB * 0.5, AABB], 0)
    o, d = _rays(1500, 5)
    oiv, osm, oterm = oracle.traverse_grids(o, d, occ, aabbs, None, None, 1e-2, 0.0)
    iv, sm, term = _traverse_gpu(cuda, o, d, occ, aabbs, step_size=1e-2, cone_angle=0.0)
    _cmp_segments(iv, sm, term, oiv, osm, oterm)
    assert np.array_equal(sm.vals.cpu().numpy(), osm["vals"])


def _ragged(n_rays, seed, max_len=200):
    rng = np.random.default_rng(seed)
    cnt = rng.integers(0, max_len, size=n_rays).astype(np.int64)
    cnt[rng.uniform(size=n_rays) < 0.3] = 0
    cnt[0] = 33; cnt[1] = 32; cnt[2] = 1; cnt[3] = 64; cnt[4] = 1040
    starts = (np.cumsum(cnt) - cnt).astype(np.int64)
    x = rng.uniform(0.01, 1.0, size=int(cnt.sum())).astype(np.float32)
    return starts, cnt, x


@pytest.mark.parametrize("exclusive", [False, True])
@pytest.mark.parametrize("backward", [False, True])
@pytest.mark.parametrize("normalize", [False, True])
def test_segmented_sums_bit_exact(cuda, oracle, exclusive, backward, normalize):
    from cnc_amd.backends import nerfacc_cuda as nc
    if backward and normalize:
        pytest.skip("reference: backward does not support normalize (scan.cu:25-26)")
    starts, cnt, x = _ragged(999, 3)
    want = oracle.segmented_scan(x, starts, cnt, exclusive, prod=False, reverse=backward, normalize=normalize)
    fn = nc.exclusive_sum if exclusive else nc.inclusive_sum
    got = fn(torch.as_tensor(starts, device=cuda), torch.as_tensor(cnt, device=cuda),
             torch.as_tensor(x, device=cuda), normalize, backward).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("exclusive", [False, True])
def test_segmented_prods_bit_exact(cuda, oracle, exclusive):
    from cnc_amd.backends import nerfacc_cuda as nc
    starts, cnt, x = _ragged(500, 8, max_len=60)
    x = (0.5 + x).astype(np.float32)
    t = lambda a: torch.as_tensor(a, device=cuda)
    want = oracle.segmented_scan(x, starts, cnt, exclusive, prod=True)
    fwd = nc.exclusive_prod_forward if exclusive else nc.inclusive_prod_forward
    got = fwd(t(starts), t(cnt), t(x)).cpu().numpy()
    assert np.array_equal(got, want)
    g = np.random.default_rng(1).normal(size=x.shape).astype(np.float32)
    want_b = oracle.prod_backward(x, want, g, starts, cnt, exclusive)
    bwd = nc.exclusive_prod_backward if exclusive else nc.inclusive_prod_backward
    got_b = bwd(t(starts), t(cnt), t(x), t(want), t(g)).cpu().numpy()
    assert np.array_equal(got_b, want_b)


def test_scan_docstring_examples(cuda):
    """Known answers from the reference docstrings (nerfacc/scan.py:36-39,78-81,127-130,170-173)."""
    from cnc_amd.nerfacc import exclusive_prod, exclusive_sum, inclusive_prod, inclusive_sum
    x = torch.tensor([1., 2., 3., 4., 5., 6., 7., 8., 9.], device=cuda)
    pk = torch.tensor([[0, 2], [2, 3], [5, 4]], device=cuda)
    assert inclusive_sum(x, pk).tolist() == [1., 3., 3., 7., 12., 6., 13., 21., 30.]
    assert exclusive_sum(x, pk).tolist() == [0., 1., 0., 3., 7., 0., 6., 13., 21.]
    assert inclusive_prod(x, pk).tolist() == [1., 2., 3., 12., 60., 6., 42., 336., 3024.]
    assert exclusive_prod(x, pk).tolist() == [1., 1., 1., 3., 12., 1., 6., 42., 336.]
    assert exclusive_sum(torch.empty(0, device=cuda), torch.zeros((3, 2), dtype=torch.long, device=cuda)).shape == (0,)


def test_march_full_frame_properties(cuda):
    """BASELINE size: 800x800 rays through the 128^3 ball occupancy (step 5e-3) — properties that
    need no oracle: count pass == fill pass, samples ordered by (ray, t), every sample inside its
    ray's box interval and inside an occupied cell, interval masks consistent."""
    from cnc_amd import synthetic
    from cnc_amd.nerfacc.grid import ray_aabb_intersect, traverse_grids
    o, d = synthetic.pinhole_rays(800, 800, 0.6911, 4.0, 0.7, 0.5, device=cuda)
    binaries = synthetic.ball_binaries(128, radius=1.0, device=cuda)
    aabbs = torch.tensor([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], device=cuda)
    iv, sm, term = traverse_grids(o, d, binaries, aabbs, step_size=5e-3, cone_angle=0.0)
    cnt = sm.packed_info[:, 1]
    S = int(cnt.sum())
    assert S == sm.vals.shape[0] and 6.0e7 < S < 7.5e7
    assert torch.equal(sm.packed_info[:, 0], torch.cumsum(cnt, 0) - cnt)
    ri = sm.ray_indices
    assert torch.all(ri[1:] >= ri[:-1])
    same = ri[1:] == ri[:-1]
    assert torch.all((sm.vals[1:] - sm.vals[:-1])[same] > 0)
    assert torch.equal(torch.bincount(ri, minlength=o.shape[0]), cnt)
    t0, t1, hit = ray_aabb_intersect(o, d, aabbs)
    assert torch.all(sm.vals >= t0[ri, 0]) and torch.all(sm.vals <= t1[ri, 0])
    assert not torch.any(cnt[~hit[:, 0]] > 0)
    # interval edges: #left == #right == #samples, and mid-points are the samples
    assert int(iv.is_left.sum()) == S and int(iv.is_right.sum()) == S
    mids = (iv.vals[iv.is_left] + iv.vals[iv.is_right]) * 0.5
    assert torch.equal(mids, sm.vals)
    # sampled cells are occupied (up to float rounding at cell faces: allow a sliver)
    pos = o[ri] + d[ri] * sm.vals[:, None]
    cell = ((pos + 1.5) / 3.0 * 128).long().clamp(0, 127)
    occ = binaries[0, cell[:, 0], cell[:, 1], cell[:, 2]]
    assert occ.float().mean() > 0.999


@pytest.mark.parametrize("n_rays", [0, 1, 7])
def test_march_degenerate_inputs(cuda, oracle, n_rays):
    """Empty ray sets, rays that all miss, and an all-False rays_mask."""
    from cnc_amd.nerfacc.grid import traverse_grids
    binaries = torch.ones((1, 8, 8, 8), dtype=torch.bool, device=cuda)
    aabbs = torch.tensor([[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]], device=cuda)
    o = torch.full((n_rays, 3), 5.0, device=cuda)
    d = torch.nn.functional.normalize(torch.ones((n_rays, 3), device=cuda), dim=-1) if n_rays else torch.zeros((0, 3), device=cuda)
    iv, sm, term = traverse_grids(o, d, binaries, aabbs, step_size=1e-2)   # pointing away: all miss
    assert sm.vals.shape == (0,) and iv.vals.shape == (0,)
    assert sm.packed_info.shape == (n_rays, 2) and int(sm.packed_info.sum()) == 0
    if n_rays:
        d2 = -d
        mask = torch.zeros(n_rays, dtype=torch.bool, device=cuda)
        iv, sm, term = traverse_grids(o, d2, binaries, aabbs, step_size=1e-2, traverse_steps_limit=4,
                                      over_allocate=True, rays_mask=mask)
        assert int(sm.packed_info[:, 1].sum()) == 0 and sm.vals.shape == (0,)
        iv, sm, term = traverse_grids(o, d2, binaries, aabbs, step_size=1e-2)
        want = oracle.traverse_grids(o.cpu().numpy(), d2.cpu().numpy(), binaries.cpu().numpy(), aabbs.cpu().numpy(),
                                     None, None, 1e-2, 0.0)
        assert np.array_equal(sm.vals.cpu().numpy(), want[1]["vals"]) and sm.vals.shape[0] > 0


def test_sample_positions_equals_torch_expression(cuda):
    """cnc_sample_positions vs the reference's rgb_sigma_fn expression (examples/utils.py:251-262) and
    the field's aabb mapping (ngp.py:518-519): bit-exact, both modes."""
    from cnc_amd.backends import nerfacc_cuda as C
    g = torch.Generator(device=cuda).manual_seed(3)
    n, S = 1000, 50001
    o = torch.randn(n, 3, device=cuda, generator=g)
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=cuda, generator=g), dim=-1)
    ri = torch.randint(0, n, (S,), device=cuda, generator=g)
    ts = torch.rand(S, device=cuda, generator=g) * 5
    te = ts + torch.rand(S, device=cuda, generator=g) * 0.01
    pos, dirs = C.sample_positions(o, d, ri, ts, te, want_dirs=True)
    assert torch.equal(pos, o[ri] + d[ri] * (ts + te)[:, None] / 2.0)
    assert torch.equal(dirs, d[ri])
    aabb = torch.tensor([-1.5, -1.4, -1.3, 1.5, 1.6, 1.7], device=cuda)
    x = C.sample_positions(o, d, ri, ts, None, aabb)
    p = o[ri] + d[ri] * ts[:, None]
    assert torch.equal(x, (p - aabb[:3]) / (aabb[3:] - aabb[:3]))
    assert C.sample_positions(o, d, ri[:0], ts[:0]).shape == (0, 3)


@pytest.mark.parametrize("k", [0, 1])
def test_hip_marcher_against_the_reference_lookup_golden(cuda, k):
    """tests/golden/march_query.npz: the HIP marcher emits exactly the candidate mid points that the
    reference's own `nerfacc.grid._query` calls occupied (cell-face rounding cases excepted)."""
    from cnc_amd.nerfacc import grid as ngrid
    from test_np_twins import check_against_reference_lookup, march_query_case
    o, d, b, aabb, step, cand_t, cand_ray, ref_occ = march_query_case(k)
    t = lambda a: torch.as_tensor(a, device=cuda)
    iv, sm, term = ngrid.traverse_grids(t(o), t(d), t(b), t(aabb), step_size=step, cone_angle=0.0)
    check_against_reference_lookup(o, d, sm.vals.cpu().numpy(), sm.ray_indices.cpu().numpy(), cand_t, cand_ray,
                                   ref_occ, b.shape[-1])


@pytest.mark.parametrize("cone", [0.0, 4e-3], ids=["constant_step", "cone_step"])
@pytest.mark.parametrize("resume", ["whole_ray", "resume_at_first_sample"])
@pytest.mark.parametrize("fill", ["staged", "direct"])
@pytest.mark.parametrize("limit,masked", [(-1, False), (9, True)])
@pytest.mark.parametrize("res,step", [(32, 2e-2), (128, 5e-3)])
def test_march_samples_equals_the_interval_edges(cuda, oracle, res, step, limit, masked, fill, resume, cone, monkeypatch):
    """cnc_march_samples (extension: (ray, t_start, t_end) per sample straight from the march) against the
    oracle's traverse_grids: t_starts == intervals.vals[is_left], t_ends == intervals.vals[is_right], same
    rays, counts and termination planes — unlimited two-pass and step-limited with dead rays.  `fill`: the LDS-staged
    fill pass of the big frames and the direct-store one small batches take (same march, same values)."""
    from cnc_amd import synthetic
    monkeypatch.setenv("CNC_MARCH_DIRECT_MAX", "0" if fill == "staged" else str(1 << 17))
    # the fill pass continuing from the state the count pass left at each ray's first sample, and stopping at its last
    monkeypatch.setenv("CNC_MARCH_RESUME", "0" if resume == "whole_ray" else "1")
    from cnc_amd.backends import nerfacc_cuda as C
    o, d = synthetic.pinhole_rays(36, 36, 0.6911, 4.0, 0.3, 0.4)
    binaries = synthetic.ball_binaries(res, radius=1.0)
    binaries ^= torch.rand(binaries.shape, generator=torch.Generator().manual_seed(2)) < 0.04
    aabbs = torch.tensor([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]])
    n = o.shape[0]
    near = torch.rand(n, generator=torch.Generator().manual_seed(3)) * step
    far = torch.full((n,), 1e10)
    mask = torch.ones(n, dtype=torch.bool)
    if masked:
        mask[::4] = False
    # `cone` = 0: the kernels specialised for a constant step (every CNC configuration); > 0: dt = clamp(t * cone, step, 1e10)
    oiv, osm, oterm = oracle.traverse_grids(o.numpy(), d.numpy(), binaries.numpy(), aabbs.numpy(), near.numpy(),
                                            far.numpy(), step, cone, traverse_steps_limit=limit if limit > 0 else None,
                                            over_allocate=limit > 0, rays_mask=mask.numpy())
    T = lambda a: a.to(cuda)
    t0, t1, hit = C.ray_aabb_intersect(T(o), T(d), T(aabbs), -float("inf"), float("inf"), float("inf"))
    order = torch.arange(2, device=cuda).expand(n, 2).contiguous()
    ri, ts, te, starts, counts, term = C.march_samples(T(o), T(d), T(mask) if masked else None, T(binaries), T(aabbs),
                                                        torch.cat([t0, t1], -1), order, hit, T(near), T(far), step, cone,
                                                        traverse_steps_limit=limit, want_terminate_planes=True)
    want_cnt = np.asarray(osm["chunk_cnts"])
    assert np.array_equal(counts.cpu().numpy(), want_cnt) and want_cnt.sum() > 2000
    assert np.array_equal(starts.cpu().numpy(), np.cumsum(want_cnt) - want_cnt)
    vals = np.asarray(oiv["vals"])
    assert np.array_equal(ts.cpu().numpy(), vals[np.asarray(oiv["is_left"]).astype(bool)])
    assert np.array_equal(te.cpu().numpy(), vals[np.asarray(oiv["is_right"]).astype(bool)])
    assert np.array_equal(ri.cpu().numpy(), np.asarray(osm["ray_indices"])[np.asarray(osm["is_valid"]).astype(bool)])
    live = mask.numpy() if limit > 0 else want_cnt > 0
    assert np.array_equal(term.cpu().numpy()[live], np.asarray(oterm)[live])
    if masked:
        assert np.all(counts.cpu().numpy()[~mask.numpy()] == 0)
    # fill-pass extras (ABI v24): positions / directions / int32 ray ids from the marching lane == the separate
    # cnc_sample_positions pass on (ray, t_start, t_end), bit for bit, with and without the unit-cube mapping
    for box in (None, torch.tensor([-1.5, -1.4, -1.3, 1.5, 1.6, 1.7], device=cuda)):
        for kind in ("int32", None, "int64"):
            ex = {"positions": True, "dirs": True, "ray_indices": kind, "aabb": box}
            ri2, ts2, te2, starts2, counts2, _ = C.march_samples(
                T(o), T(d), T(mask) if masked else None, T(binaries), T(aabbs), torch.cat([t0, t1], -1), order, hit,
                T(near), T(far), step, cone, traverse_steps_limit=limit, extras=ex)
            assert torch.equal(ts2, ts) and torch.equal(te2, te) and torch.equal(counts2, counts)
            if kind is None:
                assert ri2 is None
            else:
                assert ri2.dtype == (torch.int32 if kind == "int32" else torch.int64) and torch.equal(ri2.long(), ri)
            pos, dirs = C.sample_positions(T(o), T(d), ri, ts, te, box, want_dirs=True)
            assert torch.equal(ex["positions"], pos) and torch.equal(ex["dirs"], dirs)
    with pytest.raises(RuntimeError, match="at least emit positions"):
        C.march_samples(T(o), T(d), None, T(binaries), T(aabbs), torch.cat([t0, t1], -1), order, hit, T(near), T(far),
                        step, cone, extras={"ray_indices": None})


def test_march_positions_equal_the_separate_pass_on_the_bench_frame(cuda):
    """bench.py's 800x800 frame (640 k rays, ~68 M samples, staged fill pass with 16-entry rows): the positions the
    fill pass emits — unit-cube normalised, what the encoder is fed — equal cnc_sample_positions on the same samples."""
    import bench
    from cnc_amd.backends import nerfacc_cuda as C
    w = bench.build_workload(cuda, 0)
    t_lo, t_hi, hit = C.ray_aabb_intersect(w["rays_o"], w["rays_d"], w["aabbs"], -float("inf"), float("inf"), float("inf"))
    args = (w["rays_o"], w["rays_d"], None, w["binaries"], w["aabbs"], torch.cat([t_lo, t_hi], -1), w["t_order"], hit,
            w["near"], w["far"], bench.STEP_SIZE, 0.0)
    ri, ts, te, starts, counts, _ = C.march_samples(*args)
    ex = {"positions": True, "aabb": w["aabb0"], "ray_indices": "int32"}
    ri32, ts2, te2, _, counts2, _ = C.march_samples(*args, extras=ex)
    assert ts.shape[0] > 6e7 and torch.equal(ts, ts2) and torch.equal(te, te2) and torch.equal(counts, counts2)
    assert ri32.dtype == torch.int32 and torch.equal(ri32.long(), ri)
    want = C.sample_positions(w["rays_o"], w["rays_d"], ri, ts, te, w["aabb0"])
    assert torch.equal(ex["positions"], want)
    assert float(want.min()) >= 0.0 and float(want.max()) <= 1.0


def test_march_samples_resume_equals_whole_ray_march_on_the_bench_frame(cuda, monkeypatch):
    """Full size, no oracle (the small cases above pin both forms to it): bench.py's own 800x800 frame, 640 k rays and
    ~68 M samples.  The fill pass that resumes at each ray's first sample and stops at its last (constant-step kernels,
    16-entry staging) must write exactly what the fill pass that marches every ray end to end writes."""
    import bench
    from cnc_amd.backends import nerfacc_cuda as C
    w = bench.build_workload(cuda, 0)
    t_lo, t_hi, hit = C.ray_aabb_intersect(w["rays_o"], w["rays_d"], w["aabbs"], -float("inf"), float("inf"), float("inf"))
    outs = []
    for resume in ("1", "0"):
        monkeypatch.setenv("CNC_MARCH_RESUME", resume)
        ri, ts, te, starts, counts, _ = C.march_samples(w["rays_o"], w["rays_d"], None, w["binaries"], w["aabbs"],
                                                        torch.cat([t_lo, t_hi], -1), w["t_order"], hit, w["near"], w["far"],
                                                        bench.STEP_SIZE, 0.0)
        outs.append((ri, ts, te, starts, counts))
    assert outs[0][1].shape[0] > 6e7
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # sortedness / packing: samples are grouped by ray in ray order, t ascending inside a ray
    ri, ts, te, starts, counts = outs[0]
    assert bool((ri[1:] >= ri[:-1]).all()) and bool((te > ts).all())
    assert int(counts.sum()) == ts.shape[0] and bool((starts == torch.cumsum(counts, 0) - counts).all())


@pytest.mark.parametrize("shape", [(1, 128, 128, 128), (2, 32, 32, 32), (1, 16, 8, 12), (3, 64, 64, 64)])
def test_coarse_occupancy_bits_and_the_march_through_them(cuda, shape, monkeypatch):
    """cnc_occupancy_coarse_bits: bit (((g cx + x) cy + y) cz + z) = any cell of the 4 x 4 x 4 block — against torch; and
    the march that consults it (cnc_march_samples_coarse: a step through an empty block is decided from LDS) returns
    exactly what the march without it returns.  A shape the kernels take no coarse grid for gives None."""
    from cnc_amd import synthetic
    from cnc_amd.backends import nerfacc_cuda as C
    g = torch.Generator().manual_seed(7)
    binaries = torch.rand(shape, generator=g) < 0.003                                 # sparse: most blocks empty, some not
    binaries[:, : shape[1] // 2, : shape[2] // 2, : shape[3] // 4] |= torch.rand((shape[0], shape[1] // 2, shape[2] // 2, shape[3] // 4), generator=g) < 0.3
    b = binaries.to(cuda)
    words = C.occupancy_coarse_bits(b)
    n, rx, ry, rz = shape
    blocks = b.view(n, rx // 4, 4, ry // 4, 4, rz // 4, 4).permute(0, 1, 3, 5, 2, 4, 6).reshape(n, rx // 4, ry // 4, rz // 4, 64).any(-1)
    want = blocks.reshape(-1).cpu().numpy()
    got = words.cpu().numpy().view(np.uint32)
    bits = ((got[np.arange(want.size) // 32] >> (np.arange(want.size) % 32).astype(np.uint32)) & 1).astype(bool)
    assert np.array_equal(bits, want) and 0 < want.sum() < want.size
    assert C.occupancy_coarse_bits(torch.zeros((1, 30, 32, 32), dtype=torch.bool, device=cuda)) is None       # 30 % 4 != 0
    assert C.occupancy_coarse_bits(torch.zeros((2, 256, 256, 128), dtype=torch.bool, device=cuda)) is None    # > 2048 words
    o, d = synthetic.pinhole_rays(40, 40, 0.6911, 4.0, 0.3, 0.4)
    aabbs = torch.tensor([[-1.5 * 2 ** k] * 3 + [1.5 * 2 ** k] * 3 for k in range(n)], device=cuda)
    o, d = o.to(cuda), d.to(cuda)
    t0, t1, hit = C.ray_aabb_intersect(o, d, aabbs, -float("inf"), float("inf"), float("inf"))
    t_sorted, t_indices = torch.sort(torch.cat([t0, t1], -1), dim=-1)
    near, far = torch.zeros(o.shape[0], device=cuda), torch.full((o.shape[0],), 1e10, device=cuda)
    out = {}
    for on in (True, False):
        monkeypatch.setattr(C, "_COARSE_ON", on)
        ex = {"positions": True}
        out[on] = C.march_samples(o, d, None, b, aabbs, t_sorted, t_indices, hit, near, far, 1e-2, 0.0,
                                  want_terminate_planes=True, extras=ex) + (ex["positions"],)
    assert int(out[True][4].sum()) > 1000
    for a, c in zip(out[True], out[False]):
        assert torch.equal(a, c)


def test_premarched_sampling_equals_the_call_that_marches_itself(cuda):
    """OccGridEstimator.premarch (extension: the march of a later `sampling` call made ahead of time, on another stream)
    hands `sampling` exactly what it would have marched itself — same samples for the same jitter — and is dropped when
    the call is not the one it was made for (other rays, another grid)."""
    from cnc_amd import synthetic
    from cnc_amd.nerfacc import OccGridEstimator
    est = OccGridEstimator(roi_aabb=[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], resolution=64).to(cuda)
    est.binaries = synthetic.ball_binaries(64, radius=1.0).to(cuda).view(1, 64, 64, 64)
    est.occs = est.binaries.reshape(-1).float()
    o, d = synthetic.pinhole_rays(48, 48, 0.6911, 4.0, 0.3, 0.4)
    o, d = o.to(cuda), d.to(cuda)
    sigma = lambda t0, t1, ri: torch.full_like(t0, 3.0)
    kw = dict(near_plane=0.1, render_step_size=1e-2, stratified=True, cone_angle=0.0, alpha_thre=0.0)
    torch.manual_seed(11)
    want = est.sampling(o, d, sigma_fn=sigma, **kw)
    side = torch.cuda.Stream(device=cuda)
    torch.manual_seed(11)
    with torch.cuda.stream(side):
        est.premarch(o, d, near_plane=0.1, render_step_size=1e-2, stratified=True, cone_angle=0.0)
    assert est._premarched is not None
    torch.manual_seed(99)                                   # the jitter was drawn by premarch: this draw is not used
    got = est.sampling(o, d, sigma_fn=sigma, **kw)
    assert est._premarched is None
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    assert want[0].numel() > 1000
    # made for other rays / another grid: dropped, the call marches itself
    with torch.cuda.stream(side):
        est.premarch(o.clone(), d, near_plane=0.1, render_step_size=1e-2, stratified=True, cone_angle=0.0)
    torch.manual_seed(11)
    got = est.sampling(o, d, sigma_fn=sigma, **kw)
    assert all(torch.equal(a, b) for a, b in zip(want, got))
    with torch.cuda.stream(side):
        est.premarch(o, d, near_plane=0.1, render_step_size=1e-2, stratified=True, cone_angle=0.0)
    torch.cuda.synchronize()
    est.binaries = est.binaries.clone()
    est.binaries[0, 30:34, 30:34, 30:34] = False
    torch.manual_seed(11)
    fresh = est.sampling(o, d, sigma_fn=sigma, **kw)
    assert fresh[0].numel() != want[0].numel()


def test_window_positions_and_counted_scatter(cuda):
    """cnc_ray_window_positions / cnc_scatter_counted (the sampler's depth windows with their sample count left on the device):
    positions = o + (d (t0 + t1)) / 2 of samples [first, first + n) of every ray, packed by the running sum of n, with the
    samples' own indices — bit-equal to cnc_sample_positions on the compacted window — and the scatter back stops at the count."""
    from cnc_amd.backends import nerfacc_cuda as C
    from cnc_amd.backends import volrend_backend as K
    g = torch.Generator(device=cuda).manual_seed(4)
    n_rays = 777
    counts = torch.randint(0, 40, (n_rays,), device=cuda, generator=g)
    counts[5] = 0
    starts = torch.cumsum(counts, 0) - counts
    n = int(counts.sum())
    t0 = torch.rand(n, device=cuda, generator=g) * 4
    t1 = t0 + torch.rand(n, device=cuda, generator=g) * 0.01
    o = torch.randn(n_rays, 3, device=cuda, generator=g)
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=cuda, generator=g), dim=-1)
    first = (torch.rand(n_rays, device=cuda, generator=g) * (counts + 1)).long().clamp_max(counts)
    take = torch.minimum(torch.full_like(counts, 9), counts - first)
    ends = torch.cumsum(take, 0)
    total = int(ends[-1])
    cap = total + 123
    pos, src = K.window_positions(starts, first, take, ends, t0, t1, o, d, cap)
    ri, ts, te, src_ref = K.window_samples(starts, first, take, t0, t1, total, ends=ends)
    want, _ = C.sample_positions(o, d, ri, ts, te, want_dirs=True)
    assert torch.equal(pos[:total], want) and torch.equal(src[:total], src_ref)
    assert torch.equal(src_ref, torch.cat([torch.arange(int(starts[r] + first[r]), int(starts[r] + first[r] + take[r]),
                                                        device=cuda) for r in range(n_rays)]))
    values = torch.rand(cap, device=cuda, generator=g)
    for count in (0, 17, total, total + 1000):
        out = torch.full((n,), -1.0, device=cuda)
        src_all = torch.cat([src[:total], torch.zeros(cap - total, dtype=torch.int64, device=cuda)])   # garbage behind the count
        K.scatter_counted(out, src_all, values, torch.tensor([count], dtype=torch.int64, device=cuda))
        k = min(count, cap)                                      # a count beyond the capacity is the capacity
        ref = torch.full((n,), -1.0, device=cuda)
        ref[src_all[:min(k, total)]] = values[:min(k, total)]
        if k > total:               # rows [total, k) all point at sample 0 here: which of their values lands there is open
            assert float(out[0]) in set(values[total:k].tolist()) | {float(ref[0])}
            out[0] = ref[0]
        assert torch.equal(out, ref)
