"""Binned (atomic-free) embedding-gradient scatter for the finest levels
(cnc_grid_encode_backward_binned) vs the CPU oracle's float64 shadow, and vs the atomic kernel."""
import numpy as np
import pytest
import torch

from conftest import make_grid
from test_gpu_encoder import _bwd_gpu, _check_bwd, _points

pytestmark = pytest.mark.gpu

RES = [6, 14, 31, 44]      # log2_T = 10: level 0 dense with 216 rows (a partial slab), 1-3 hashed, 1024 rows


def _bwd_binned(dev, g, x, emb, offs, res, n_binned, level_rows, ste=False, ws_bytes=None, clip=None,
                ld=0, col=0, g_dev=None):
    from cnc_amd import _lib
    t = lambda a: None if a is None else torch.as_tensor(a, device=dev)
    L = len(res)
    N, D = x.shape
    F = emb.shape[1]
    lib = _lib.lib()
    if ws_bytes is None:
        ws_bytes = int(lib.cnc_grid_encode_backward_binned_workspace(N, n_binned, level_rows))
    ws = torch.full((max(ws_bytes, 4),), 0xAB, dtype=torch.uint8, device=dev)   # library must clear it
    ge = torch.zeros(emb.shape, dtype=torch.float32, device=dev)
    gd = t(g) if g_dev is None else g_dev
    xd, ed, od, rd, cd = t(x), t(emb), t(offs), t(res), t(clip)
    rc = lib.cnc_grid_encode_backward_binned(
        gd.data_ptr(), xd.data_ptr(), ed.data_ptr(), od.data_ptr(), rd.data_ptr(), ge.data_ptr(),
        N, D, F, L, _lib.CNC_FLAG_STE_BINARY if ste else 0, _lib.ptr(cd), ld, col,
        n_binned, level_rows, ws.data_ptr(), ws_bytes, _lib.stream())
    _lib.check(rc, "binned")
    torch.cuda.synchronize()
    return ge.cpu().numpy()


@pytest.mark.parametrize("F", [2, 4, 8])
@pytest.mark.parametrize("ste", [False, True])
@pytest.mark.parametrize("n_binned", [1, 3, 4])
def test_binned_backward_against_float64_shadow(cuda, oracle, F, ste, n_binned):
    offs, resl, emb = make_grid(RES, 10, 3, F, seed=41)
    x = _points(9001, 3, seed=42)
    g = np.random.default_rng(43).normal(size=(len(RES), x.shape[0], F)).astype(np.float32)
    want32, acc64 = oracle.grid_encode_backward(g, x, emb, offs, resl, ste_binary=ste, want_acc64=True)
    _, abs64 = oracle.grid_encode_backward(np.abs(g), x, emb, offs, resl, ste_binary=ste, want_acc64=True)
    got = _bwd_binned(cuda, g, x, emb, offs, resl, n_binned, 1024, ste=ste)
    _check_bwd(got, want32, acc64, abs64, n_terms_max=x.shape[0] * 8)
    if ste:
        assert np.all(got[np.abs(emb) > 1] == 0)


@pytest.mark.parametrize("case", ["tiny_workspace", "level_rows_too_small", "clip_count_zero"])
def test_binned_backward_spill_and_fallback_paths(cuda, oracle, case):
    """A full bin spills its excess items to atomics; a level with more rows than the bins were sized
    for is scattered with atomics entirely; the clip-count hint skips the STE mask.  Same gradient."""
    F = 8
    offs, resl, emb = make_grid(RES, 10, 3, F, seed=51, binary=(case == "clip_count_zero"))
    x = _points(6000, 3, seed=52)
    g = np.random.default_rng(53).normal(size=(len(RES), x.shape[0], F)).astype(np.float32)
    ste = case == "clip_count_zero"
    want32, acc64 = oracle.grid_encode_backward(g, x, emb, offs, resl, ste_binary=ste, want_acc64=True)
    _, abs64 = oracle.grid_encode_backward(np.abs(g), x, emb, offs, resl, ste_binary=ste, want_acc64=True)
    kw = {}
    if case == "tiny_workspace":
        kw = dict(level_rows=1024, ws_bytes=3 * 4 * (16 + 64 * 16))   # 64 item slots per bin, ~6000 wanted
    elif case == "level_rows_too_small":
        kw = dict(level_rows=512)
    else:
        kw = dict(level_rows=1024, clip=np.zeros(1, np.int32))
    got = _bwd_binned(cuda, g, x, emb, offs, resl, 3, ste=ste, **kw)
    _check_bwd(got, want32, acc64, abs64, n_terms_max=x.shape[0] * 8)


@pytest.mark.parametrize("N", [0, 1, 63, 4096, 4097])
def test_binned_backward_ragged_sizes_match_atomic_kernel(cuda, N):
    offs, resl, emb = make_grid(RES, 10, 3, 8, seed=61)
    x = _points(N, 3, seed=62)
    g = np.random.default_rng(63).normal(size=(len(RES), N, 8)).astype(np.float32)
    a = _bwd_gpu(cuda, g, x, emb, offs, resl)
    b = _bwd_binned(cuda, g, x, emb, offs, resl, 2, 1024)
    scale = max(np.abs(a).max(), 1e-6) if N else 1.0
    assert np.abs(a - b).max() <= 1e-5 * scale
    # entries no sample touches stay exactly 0 (touched ones may cancel to 0 in one summation order only)
    touched = _bwd_gpu(cuda, np.abs(g), x, emb, offs, resl) != 0
    assert np.all(b[~touched] == 0) and np.all(a[~touched] == 0)


def test_binned_backward_point_major_gradient(cuda):
    """grad_ld / grad_col: the gradient read in place from a wider [N, ld] matrix."""
    F, N, ld, col = 4, 5000, 40, 8
    offs, resl, emb = make_grid(RES, 10, 3, F, seed=71)
    x = _points(N, 3, seed=72)
    g = np.random.default_rng(73).normal(size=(len(RES), N, F)).astype(np.float32)
    wide = torch.randn(N, ld, device=cuda)
    wide[:, col:col + len(RES) * F] = torch.as_tensor(g, device=cuda).permute(1, 0, 2).reshape(N, -1)
    a = _bwd_binned(cuda, g, x, emb, offs, resl, 3, 1024)
    b = _bwd_binned(cuda, None, x, emb, offs, resl, 3, 1024, ld=ld, col=col, g_dev=wide.contiguous())
    assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max()


def test_plan_and_mirror_route(cuda):
    """`plan_binned_levels` picks the finest big levels; the `_gridencoder` mirror routes to the
    binned entry and gives the atomic kernel's gradient."""
    from cnc_amd.backends import gridencoder_backend as be
    from cnc_amd.synthetic import RES_16L, level_offsets
    offs16 = level_offsets(RES_16L, 19, 3)
    assert be.plan_binned_levels(RES_16L, offs16, 3, 8, 1 << 20) == (6, 1 << 19)
    assert be.plan_binned_levels(RES_16L, offs16, 3, 8, 1000) is None
    assert be.plan_binned_levels(RES_16L, offs16, 2, 8, 1 << 20) is None
    res = [20, 40, 90, 200]
    offs, resl, emb = make_grid(res, 16, 3, 8, seed=81)
    N = 1 << 16
    x = torch.rand(N, 3, device=cuda)
    g = torch.randn(len(res), N, 8, device=cuda)
    plan = be.plan_binned_levels(res, offs, 3, 8, N, min_resolution=64, min_work=0)
    assert plan == (2, 1 << 16)
    assert be.plan_binned_levels(res, offs, 3, 8, N, min_resolution=64) is None      # too little work for the bin pass
    t = lambda a: torch.as_tensor(a, device=cuda)
    outs = []
    for binned in (None, plan):
        ge = torch.zeros(emb.shape, device=cuda)
        be.grid_encode_backward(g, x, t(emb), t(offs), t(resl), ge, N, 3, 8, len(res), 0, 128, None, None,
                                None, None, ste_binary=True, binned=binned)
        outs.append(ge)
    torch.cuda.synchronize()
    assert (outs[0] - outs[1]).abs().max() <= 2e-5 * outs[0].abs().max()
    ge = torch.zeros(emb.shape, device=cuda)
    be.grid_encode_backward(g.abs(), x, t(emb), t(offs), t(resl), ge, N, 3, 8, len(res), 0, 128, None, None,
                            None, None, ste_binary=False)
    assert (outs[1][ge == 0] == 0).all()          # rows no sample touches stay exactly 0


@pytest.mark.parametrize("N", [4096, 9001])
def test_binned_backward_coherent_points_with_spills(cuda, N):
    """Sorted points (long same-cell runs, one hot bin) with a workspace too small for them: part of
    every bin goes through the owner wave, the rest through the atomic spill; gradients carry
    distinct values per sample so a mixed-up sample index cannot hide."""
    F = 8
    offs, resl, emb = make_grid(RES, 10, 3, F, seed=91)
    rng = np.random.default_rng(92)
    x = rng.uniform(0.05, 0.95, size=(N, 3)).astype(np.float32)
    x = x[np.lexsort((x[:, 0], x[:, 1], x[:, 2]))]
    g = rng.normal(size=(len(RES), N, F)).astype(np.float32)
    a = _bwd_gpu(cuda, g, x, emb, offs, resl)
    for ws_bytes in (4 * 4 * (16 + 64 * 16), 4 * 4 * (16 + (N // 8) * 16), None):
        b = _bwd_binned(cuda, g, x, emb, offs, resl, 4, 1024, ws_bytes=ws_bytes)
        assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max()


def test_binned_backward_full_size_equals_atomic_kernel(cuda):
    """BASELINE size (16L x 2^19 x F8, 2^20 ray-like samples, plan of the product path): the binned
    call and the all-atomic call give the same table gradient; the sum of the gradient equals the
    sum of the incoming gradient over in-range samples (the corner weights sum to 1: linearity)."""
    from cnc_amd.backends import gridencoder_backend as be
    from cnc_amd.synthetic import RES_16L, level_offsets
    F, L, N = 8, 16, 1 << 20
    offs = level_offsets(RES_16L, 19, 3)
    o_t = torch.as_tensor(offs, device=cuda)
    r_t = torch.tensor(RES_16L, dtype=torch.int32, device=cuda)
    g = torch.Generator(device=cuda).manual_seed(7)
    emb = torch.sign(torch.rand((int(offs[-1]), F), device=cuda, generator=g) * 2 - 1)
    # 8192 rays of 128 samples, step 1/600 of the unit cube, inside the interior of every level
    o = torch.rand((N // 128, 1, 3), device=cuda, generator=g) * 0.3 + 0.2
    d = torch.nn.functional.normalize(torch.randn((N // 128, 1, 3), device=cuda, generator=g), dim=-1)
    x = (o + d * (torch.arange(128, device=cuda).view(1, 128, 1) / 600.0)).clamp(0.02, 0.98).reshape(-1, 3).contiguous()
    grad = torch.randn((L, N, F), device=cuda, generator=g)
    plan = be.plan_binned_levels(RES_16L, offs, 3, F, N)
    assert plan is not None and plan[0] >= 5
    outs = []
    for binned in (None, plan):
        ge = torch.zeros_like(emb)
        be.grid_encode_backward(grad, x, emb, o_t, r_t, ge, N, 3, F, L, 0, 128, None, None, None, None,
                                ste_binary=True, binned=binned)
        outs.append(ge)
    a, b = outs
    assert (a - b).abs().max() <= 2e-5 * a.abs().max()
    # (no exact-zero comparison here: with ~16 terms per entry a sum can cancel to 0 in one order only)
    # per level: sum over the table rows of the level == sum over samples of the incoming gradient
    for l in range(L):
        want = grad[l].double().sum(0)
        got = b[int(offs[l]):int(offs[l + 1])].double().sum(0)
        assert (got - want).abs().max() <= 1e-4 * grad[l].abs().double().sum(0).max()


@pytest.mark.parametrize("D,F", [(3, 8), (3, 2), (2, 8)])
@pytest.mark.parametrize("N", [1, 300, 70001])
def test_interleaved_level_schedule_gives_the_same_gradient(cuda, oracle, D, F, N):
    """CNC_FLAG_LEVELS_FINEST_FIRST (mirror kwarg `interleave_levels`) only changes which block works
    on which (chunk, level) — 1-D grid, level slot as the fast index, slots walked last to first — so
    the plain entry must return the same sums (fp32 reordering aside) for ragged sizes and every
    lane mapping (64, 16 and 32 lanes per run)."""
    from cnc_amd.backends import gridencoder_backend as be
    res = RES if D == 3 else [10, 18, 34, 66]
    offs, resl, emb = make_grid(res, 10, D, F, seed=61)
    x = _points(N, D, seed=62)
    g = np.random.default_rng(63).normal(size=(len(res), N, F)).astype(np.float32)
    _, acc64 = oracle.grid_encode_backward(g, x, emb, offs, resl, ste_binary=True, want_acc64=True)
    t = lambda a: torch.as_tensor(a, device=cuda)
    outs = []
    for il in (False, True):
        ge = torch.zeros(emb.shape, dtype=torch.float32, device=cuda)
        be.grid_encode_backward(t(g), t(x), t(emb), t(offs), t(resl), ge, N, D, F, len(res), 0, 128, None, None,
                                None, None, ste_binary=True, interleave_levels=il)
        torch.cuda.synchronize()
        outs.append(ge.cpu().numpy())
    scale = max(np.abs(acc64).max(), 1e-30)
    for got in outs:
        assert np.abs(got - acc64).max() <= 1e-5 * scale
    assert np.array_equal(outs[0] == 0, outs[1] == 0) or np.abs(outs[0] - outs[1]).max() <= 1e-5 * scale


@pytest.mark.parametrize("point_major", [False, True])
def test_two_stream_split_equals_single_call(cuda, point_major):
    """From 2^16 points (every binned call) the mirror runs the coarse (atomic) levels on a side stream next to the bin /
    owner passes of the finest levels (disjoint table rows).  Same gradient as the one-stream call, for
    the level-major and the point-major gradient layout, and correctly ordered against work queued
    before and after on the caller's stream."""
    from cnc_amd.backends import gridencoder_backend as be
    from cnc_amd.synthetic import RES_16L, level_offsets
    F, L, N = 8, 16, (1 << 19) + 1234
    offs = level_offsets(RES_16L, 19, 3)
    o_t, r_t = torch.as_tensor(offs, device=cuda), torch.tensor(RES_16L, dtype=torch.int32, device=cuda)
    gen = torch.Generator(device="cpu").manual_seed(5)
    emb = torch.sign(torch.rand((int(offs[-1]), F), generator=gen) * 2 - 1).to(cuda)
    x = torch.rand((N, 3), generator=gen).to(cuda)
    plan = be.plan_binned_levels(RES_16L, [int(v) for v in offs], 3, F, N)
    assert plan is not None and 0 < plan[0] < L
    g_lm = torch.randn((L, N, F), generator=gen).to(cuda)
    outs = []
    for overlap in (False, True):
        if point_major:
            g = torch.zeros((N, L * F + 8), device=cuda)
            g[:, 4:4 + L * F] = g_lm.permute(1, 0, 2).reshape(N, L * F)       # queued just before the call
            kw = dict(grad_ld=L * F + 8, grad_col=4)
        else:
            g = g_lm.clone()
            kw = {}
        ge = torch.empty_like(emb)
        ge.zero_()                       # queued on the caller's stream right before the call
        be.grid_encode_backward(g, x, emb, o_t, r_t, ge, N, 3, F, L, 0, 128, None, None, None, None,
                                ste_binary=True, binned=plan, overlap_streams=overlap, **kw)
        outs.append(ge.clone())          # queued right after: must see both halves
    torch.cuda.synchronize()
    scale = float(outs[0].abs().max())
    assert scale > 0
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-5 * scale
    # both halves contributed: coarsest and finest level rows are non-zero
    assert float(outs[1][: int(offs[1])].abs().max()) > 0 and float(outs[1][int(offs[-2]):].abs().max()) > 0


@pytest.mark.parametrize("overlap,which", [(False, "middle"), (True, "middle"), (True, "first")])
def test_bench_chunk_backward_against_oracle(cuda, oracle, overlap, which):
    """THE call bench.py times, oracle-checked at bench size: one chunk (bench.CHUNK samples) from the middle of
    bench.py's own marched 800x800 frame (and the FIRST chunk: 36k rays grazing the top of the ball, a thin
    slice of space whose rows hash unevenly onto the owner slabs — bins up to 7.7x the mean, shared by several
    owner waves), 16L x 2^19 x F8, raw U(-1e-4, 1e-4) table with ste_binary, the
    product's binned plan (coarse levels on k_grid_encode_bwd_merge with its cross-ray LDS hash chains,
    finest levels on k_bwd_bin + k_bwd_owner), two-stream overlap off and on.  Every table entry must lie
    within the float32 summation bound (n_e + 2) * eps * sum|terms| of the oracle's float64 sum, with n_e
    the entry's own term count from the independent NumPy restatement (tests/np_twins.py)."""
    import bench
    import np_twins as tw
    from cnc_amd.backends import gridencoder_backend as be
    from cnc_amd.backends import nerfacc_cuda as ncu
    from cnc_amd.nerfacc import grid as ngrid
    w = bench.build_workload(cuda, 0)
    box = {}
    # bench.step's own march + positions
    t_lo, t_hi, hit = ncu.ray_aabb_intersect(w["rays_o"], w["rays_d"], w["aabbs"], -float("inf"), float("inf"), float("inf"))
    ri, ts, te = ncu.march_samples(w["rays_o"], w["rays_d"], None, w["binaries"], w["aabbs"], torch.cat([t_lo, t_hi], -1),
                                   w["t_order"], hit, w["near"], w["far"], bench.STEP_SIZE, 0.0)[:3]
    x = ncu.sample_positions(w["rays_o"], w["rays_d"], ri, ts, te, w["aabbs"][0])
    S, N, L, F = x.shape[0], bench.CHUNK, bench.L, bench.F
    assert S > 8 * N            # (CNC_BENCH_CHUNK may enlarge the call)
    c = (S // N) // 2 if which == "middle" else 0
    xs = x[c * N:(c + 1) * N].contiguous()
    be.pack_sign_bits(w["table"], w["bits"], w["clip"])
    out = w["out"]
    be.grid_encode_forward_bits(xs, w["bits"], w["offsets"], w["resolutions"], out, N, 3, F, L, 128)
    plan = be.plan_binned_levels(bench.synthetic.RES_16L, w["offsets_host"], 3, F, N)
    assert plan is not None and 0 < plan[0] < L
    gt = torch.zeros_like(w["table"])
    be.grid_encode_backward(out, xs, w["table"], w["offsets"], w["resolutions"], gt, N, 3, F, L, 0, 128, None, None,
                            None, None, ste_binary=True, ste_clip_count=w["clip"], binned=plan,
                            overlap_streams=overlap)
    torch.cuda.synchronize()
    got = gt.cpu().numpy()
    xn, g, table = xs.cpu().numpy(), out.cpu().numpy(), w["table"].cpu().numpy()
    offs, res = w["offsets"].cpu().numpy(), w["resolutions"].cpu().numpy()
    # forward of the same chunk, bit-exact (the gradient fed back is the oracle's own output too)
    threads = oracle.max_threads()
    assert np.array_equal(g, oracle.grid_encode_forward(xn, table, offs, res, ste_binary=True, threads=threads))
    want32, acc64 = oracle.grid_encode_backward(g, xn, table, offs, res, ste_binary=True, want_acc64=True,
                                                threads=threads)
    _, abs64 = oracle.grid_encode_backward(np.abs(g), xn, table, offs, res, ste_binary=True, want_acc64=True,
                                           threads=threads)
    n_e = tw.grid_entry_counts(xn, offs, res)
    assert n_e.max() > 10000          # the coarse levels really are many-to-one here
    eps = np.finfo(np.float32).eps
    bound = (n_e[:, None] + 2) * eps * abs64 + 1e-30
    err = np.abs(got.astype(np.float64) - acc64)
    assert np.all(err <= bound), f"worst entry: {np.max(err / bound):.3f} x its bound"
    assert np.all(got[abs64 == 0] == 0)
    # and globally far tighter than the worst case (errors do not line up)
    assert err.max() <= 1e-5 * np.abs(acc64).max()


def test_overlapped_entry_straight_through_the_c_abi(cuda):
    """cnc_grid_encode_backward_overlapped as a C host would call it (ABI v21): create a plan, one call on a
    non-default stream with the caller's own scratch, destroy the plan.  Same gradient as the serial binned entry;
    the work is ordered on the caller's stream when the call returns (no synchronisation in between)."""
    import ctypes as C

    from cnc_amd import _lib
    from cnc_amd.backends import gridencoder_backend as be
    from cnc_amd.synthetic import RES_16L, level_offsets
    F, L, N = 8, 16, (1 << 18) + 77
    offs = level_offsets(RES_16L, 19, 3)
    o_t, r_t = torch.as_tensor(offs, device=cuda), torch.tensor(RES_16L, dtype=torch.int32, device=cuda)
    gen = torch.Generator(device="cpu").manual_seed(9)
    emb = torch.sign(torch.rand((int(offs[-1]), F), generator=gen) * 2 - 1).to(cuda)
    x = torch.rand((N, 3), generator=gen).to(cuda)
    g = torch.randn((L, N, F), generator=gen).to(cuda)
    n_binned, level_rows = be.plan_binned_levels(RES_16L, [int(v) for v in offs], 3, F, N)
    lib = _lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    want = torch.zeros_like(emb)
    ws0 = torch.empty(int(lib.cnc_grid_encode_backward_binned_workspace(N, n_binned, level_rows)), dtype=torch.uint8, device=cuda)
    rc = lib.cnc_grid_encode_backward_binned(p(g), p(x), p(emb), p(o_t), p(r_t), p(want), N, 3, F, L, _lib.CNC_FLAG_STE_BINARY, None,
                                             0, 0, n_binned, level_rows, p(ws0), ws0.numel(), _lib.stream(cuda))
    assert rc == 0
    torch.cuda.synchronize()
    plan = C.c_void_p()
    assert lib.cnc_backward_plan_create(C.byref(plan)) == 0 and plan.value
    nbytes = int(lib.cnc_grid_encode_backward_overlapped_workspace(N, n_binned, level_rows))
    assert nbytes >= ws0.numel()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=cuda)
    side = torch.cuda.Stream(device=cuda)
    got = torch.empty_like(emb)
    with torch.cuda.stream(side):
        got.zero_()                                  # queued on the caller's stream right before
        for _ in range(2):                           # a plan is reusable call after call
            got.zero_()
            rc = lib.cnc_grid_encode_backward_overlapped(plan, p(g), p(x), p(emb), p(o_t), p(r_t), p(got), N, 3, F, L,
                                                         _lib.CNC_FLAG_STE_BINARY, None, 0, 0, n_binned, level_rows, p(ws),
                                                         ws.numel(), C.c_void_p(side.cuda_stream))
            assert rc == 0
        snap = got.clone()                           # queued right after: must see both halves
    side.synchronize()
    assert lib.cnc_backward_plan_destroy(plan) == 0
    scale = float(want.abs().max())
    assert float((snap - want).abs().max()) <= 1e-5 * scale
    assert float(snap[: int(offs[1])].abs().max()) > 0 and float(snap[int(offs[-2]):].abs().max()) > 0
    # NULL plan / tiny N: the serial path, same entry
    got2 = torch.zeros_like(emb)
    rc = lib.cnc_grid_encode_backward_overlapped(None, p(g), p(x), p(emb), p(o_t), p(r_t), p(got2), N, 3, F, L,
                                                 _lib.CNC_FLAG_STE_BINARY, None, 0, 0, n_binned, level_rows, p(ws), ws.numel(),
                                                 _lib.stream(cuda))
    assert rc == 0
    torch.cuda.synchronize()
    assert float((got2 - want).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("lane_stores", [False, True])
def test_sorted_bin_pass_overflow_and_round2_switch(cuda, oracle, lane_stores):
    """k_bwd_bin_sorted keeps a 16-bit source id per item in LDS, 5 per sample; when a block has more (here: every
    corner pair straddles a slab edge — all points sit in cell x = 255 of a resolution-300 level, so the x and the x+1
    rows differ in bit 8 — 8 items per sample) it stores lane by lane like the round-2 kernel.  Same gradient, also
    with the round-2 bin pass selected by CNC_FLAG_BIN_LANE_STORES, and for points that do not straddle."""
    from cnc_amd import _lib
    F = 8
    res = [20, 300]
    offs, resl, emb = make_grid(res, 12, 3, F, seed=61)
    rng = np.random.default_rng(62)
    N = 9000
    x = rng.uniform(0.02, 0.98, size=(N, 3)).astype(np.float32)
    x[:6000, 0] = (255.2 / 298.0)                       # floor(x * (R - 2) + 0.5) = 255 for R = 300
    g = rng.normal(size=(len(res), N, F)).astype(np.float32)
    want32, acc64 = oracle.grid_encode_backward(g, x, emb, offs, resl, ste_binary=True, want_acc64=True)
    _, abs64 = oracle.grid_encode_backward(np.abs(g), x, emb, offs, resl, ste_binary=True, want_acc64=True)
    t = lambda a: torch.as_tensor(a, device=cuda)
    lib = _lib.lib()
    ws_bytes = int(lib.cnc_grid_encode_backward_binned_workspace(N, 1, 4096))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=cuda)
    ge = torch.zeros(emb.shape, dtype=torch.float32, device=cuda)
    gd, xd, ed, od, rd = t(g), t(x), t(emb), t(offs), t(resl)
    flags = _lib.CNC_FLAG_STE_BINARY | (_lib.CNC_FLAG_BIN_LANE_STORES if lane_stores else 0)
    rc = lib.cnc_grid_encode_backward_binned(gd.data_ptr(), xd.data_ptr(), ed.data_ptr(), od.data_ptr(), rd.data_ptr(),
                                             ge.data_ptr(), N, 3, F, 2, flags, None, 0, 0, 1, 4096, ws.data_ptr(), ws_bytes,
                                             _lib.stream())
    _lib.check(rc, "binned")
    torch.cuda.synchronize()
    _check_bwd(ge.cpu().numpy(), want32, acc64, abs64, n_terms_max=N * 8)
    # the binned level really received its gradient through the bins: most items landed there
    counts = ws[: 16 * 4].view(torch.int32).cpu().numpy()                   # the 16 bin counters lead the workspace
    assert counts.sum() > 6000 * 7
