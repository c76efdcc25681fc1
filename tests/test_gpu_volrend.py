"""Fused volume-rendering kernels (cnc_amd/csrc/volrend.hip) through the C ABI mirror, against the oracle
(oracle.render_weight_from_density / composite: the reference's op chain restated, pinned to the reference's
own functions by tests/golden/render.npz) and against a float64 torch autograd of the same chain."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ragged(n_rays, seed, max_len=150):
    rng = np.random.default_rng(seed)
    cnts = rng.integers(0, max_len, size=n_rays).astype(np.int64)
    cnts[::11] = 0                                   # rays without samples
    cnts[5] = 1
    cnts[7] = 32
    cnts[9] = 33
    cnts[13] = 64
    cnts[15] = 257
    starts = np.cumsum(cnts) - cnts
    S = int(cnts.sum())
    ri = np.repeat(np.arange(n_rays, dtype=np.int64), cnts)
    dt = rng.uniform(1e-3, 2e-2, size=S).astype(np.float32)
    t0 = np.zeros(S, np.float32)
    for r in range(n_rays):                           # increasing t along each ray, gaps between intervals
        a, b = starts[r], starts[r] + cnts[r]
        t0[a:b] = 2.0 + np.cumsum(dt[a:b] * rng.uniform(1.0, 2.0, size=b - a))
    t1 = (t0 + dt).astype(np.float32)
    sig = (rng.uniform(size=S) ** 4 * 80).astype(np.float32)
    sig[rng.uniform(size=S) < 0.1] = 0
    rgb = rng.uniform(size=(S, 3)).astype(np.float32)
    return starts, cnts, ri, t0, t1, sig, rgb


def _dev(cuda, *arrs):
    return [None if a is None else torch.as_tensor(a, device=cuda) for a in arrs]


@pytest.mark.parametrize("n_rays", [1, 40, 777])
@pytest.mark.parametrize("with_prefix", [False, True])
def test_forward_against_oracle(cuda, oracle, n_rays, with_prefix):
    from cnc_amd.backends import volrend_backend as K
    starts, cnts, ri, t0, t1, sig, rgb = _ragged(max(n_rays, 16), seed=n_rays)
    starts, cnts = starts[:n_rays], cnts[:n_rays]
    S = int(cnts.sum())
    ri, t0, t1, sig, rgb = ri[:S], t0[:S], t1[:S], sig[:S], rgb[:S]
    op_in = np.random.default_rng(3).uniform(0, 0.9, size=n_rays).astype(np.float32) if with_prefix else None
    prefix = None if op_in is None else (np.float32(1) - op_in)[ri]
    w0, tr0, al0 = oracle.render_weight_from_density(t0, t1, sig, starts, cnts, prefix_trans=prefix)
    col0, op0, d0 = oracle.composite(w0, rgb, t0, t1, ri, n_rays, finalize=False)
    s_, c_, t0_, t1_, sig_, rgb_, op_in_ = _dev(cuda, starts, cnts, t0, t1, sig, rgb, op_in)
    w, tr, al, col, op, dep = K.volrend_forward(s_, c_, t0_, t1_, sig_, rgb_, opacity_in=op_in_)
    # per-sample: the same tile-tree sum; only exp() may differ by an ulp
    assert np.allclose(al.cpu().numpy(), al0, rtol=2e-6, atol=2.5e-7)
    assert np.allclose(tr.cpu().numpy(), tr0, rtol=4e-6, atol=1e-30)
    assert np.allclose(w.cpu().numpy(), w0, rtol=6e-6, atol=2.5e-7)
    assert np.allclose(col.cpu().numpy(), col0, rtol=1e-5, atol=1e-6)
    assert np.allclose(op.cpu().numpy(), op0, rtol=1e-5, atol=1e-6)
    assert np.allclose(dep.cpu().numpy(), d0, rtol=1e-5, atol=1e-6)
    assert np.all(op.cpu().numpy()[cnts == 0] == 0)              # rays without samples are written too
    # finalised outputs (depth / opacity, background)
    bk = np.array([0.2, 0.4, 0.9], np.float32)
    colf, opf, depf = oracle.composite(w0, rgb, t0, t1, ri, n_rays, render_bkgd=bk, finalize=True)
    _, _, _, col2, op2, dep2 = K.volrend_forward(s_, c_, t0_, t1_, sig_, rgb_, opacity_in=op_in_,
                                                 render_bkgd=torch.as_tensor(bk, device=cuda), want_samples=False,
                                                 finalize=True)
    assert np.allclose(col2.cpu().numpy(), colf, rtol=1e-5, atol=1e-6)
    assert np.allclose(dep2.cpu().numpy(), depf, rtol=2e-5, atol=1e-6)
    # in-place accumulation on top of existing sums
    base = [torch.full((n_rays, 3), 0.5, device=cuda), torch.full((n_rays, 1), 0.25, device=cuda),
            torch.full((n_rays, 1), 2.0, device=cuda)]
    K.volrend_forward(s_, c_, t0_, t1_, sig_, rgb_, opacity_in=op_in_, want_samples=False, accumulate_into=base)
    assert np.allclose(base[0].cpu().numpy(), 0.5 + col0, rtol=1e-5, atol=1e-6)
    assert np.allclose(base[1].cpu().numpy(), 0.25 + op0, rtol=1e-5, atol=1e-6)
    assert np.allclose(base[2].cpu().numpy(), 2.0 + d0, rtol=1e-5, atol=1e-6)


def test_forward_against_reference_golden(cuda):
    """tests/golden/render.npz (the reference's own functions) straight against the HIP kernel."""
    from cnc_amd.backends import volrend_backend as K
    g = np.load(os.path.join(GOLD, "render.npz"))
    R, M = g["sigmas"].shape
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a.reshape(R * M, *a.shape[2:])), device=cuda)
    starts = torch.arange(R, device=cuda) * M
    cnts = torch.full((R,), M, dtype=torch.int64, device=cuda)
    for case in ("plain", "prefix"):
        pt = t(g["prefix"]) if case == "prefix" else None
        w, tr, al, col, op, dep = K.volrend_forward(starts, cnts, t(g["t_starts"]), t(g["t_ends"]), t(g["sigmas"]),
                                                    t(g["rgbs"]), prefix_trans=pt,
                                                    render_bkgd=torch.as_tensor(g["bkgd"], device=cuda), finalize=True)
        assert np.allclose(w.cpu().numpy().reshape(R, M), g[f"{case}_weights"], rtol=3e-5, atol=2.5e-7)
        assert np.allclose(col.cpu().numpy(), g[f"{case}_colors_bkgd"], atol=1e-5)      # north_star: rgb within 1e-4
        assert np.allclose(op.cpu().numpy(), g[f"{case}_opacity"], atol=1e-5)
        assert np.allclose(dep.cpu().numpy(), g[f"{case}_depth"], rtol=2e-5, atol=1e-5)


def _chain64(t0, t1, sig, rgb, starts, cnts, bk):
    """The reference's op chain in float64 torch (CPU, autograd)."""
    tau = sig * (t1 - t0)
    before = torch.cat([torch.cumsum(torch.nn.functional.pad(tau[a:a + n][:-1], (1, 0)), 0)
                        for a, n in zip(starts.tolist(), cnts.tolist()) if n > 0])
    trans, alpha = torch.exp(-before), 1 - torch.exp(-tau)
    w = trans * alpha
    ri = torch.repeat_interleave(torch.arange(len(cnts)), torch.as_tensor(cnts))
    R = len(cnts)
    col = torch.zeros(R, 3, dtype=torch.float64).index_add(0, ri, w[:, None] * rgb)
    op = torch.zeros(R, 1, dtype=torch.float64).index_add(0, ri, w[:, None])
    dsum = torch.zeros(R, 1, dtype=torch.float64).index_add(0, ri, (w * (t0 + t1) / 2)[:, None])
    dep = dsum / op.clamp_min(torch.finfo(torch.float32).eps)
    return col + bk * (1 - op), op, dep, w, trans, alpha


@pytest.mark.parametrize("n_rays", [3, 300])
def test_autograd_against_float64_chain(cuda, n_rays):
    """rendering() forward + backward (one kernel each) vs float64 autograd of the op chain: gradients of a
    loss that touches colour, opacity, depth AND the per-sample weights."""
    import cnc_amd.nerfacc as n
    starts, cnts, ri, t0, t1, sig, rgb = _ragged(max(n_rays, 16), seed=100 + n_rays, max_len=90)
    starts, cnts = starts[:n_rays], cnts[:n_rays]
    S = int(cnts.sum())
    ri, t0, t1, sig, rgb = ri[:S], t0[:S], t1[:S], np.minimum(sig[:S], 30), rgb[:S]
    rng = np.random.default_rng(9)
    gc, go, gd, gw = rng.normal(size=(n_rays, 3)), rng.normal(size=(n_rays, 1)), rng.normal(size=(n_rays, 1)), rng.normal(size=S)
    bk = np.array([0.3, 0.6, 0.1])

    s64 = torch.tensor(sig, dtype=torch.float64, requires_grad=True)
    r64 = torch.tensor(rgb, dtype=torch.float64, requires_grad=True)
    col, op, dep, w, _, _ = _chain64(torch.tensor(t0, dtype=torch.float64), torch.tensor(t1, dtype=torch.float64), s64, r64,
                                     starts, cnts, torch.tensor(bk))
    loss = (col * torch.tensor(gc)).sum() + (op * torch.tensor(go)).sum() + (dep * torch.tensor(gd)).sum() + (w * torch.tensor(gw)).sum()
    loss.backward()

    sg = torch.tensor(sig, device=cuda, requires_grad=True)
    rg = torch.tensor(rgb, device=cuda, requires_grad=True)
    T = lambda a: torch.as_tensor(a, device=cuda)
    col_g, op_g, dep_g, extras = n.rendering(T(t0), T(t1), T(ri), n_rays=n_rays,
                                             rgb_sigma_fn=lambda a, b, c: (rg, sg, None),
                                             render_bkgd=T(bk.astype(np.float32)))
    assert torch.allclose(col_g.double().cpu(), col.detach(), atol=2e-6)
    assert torch.allclose(dep_g.double().cpu(), dep.detach(), rtol=2e-5, atol=2e-6)
    loss_g = (col_g * T(gc).float()).sum() + (op_g * T(go).float()).sum() + (dep_g * T(gd).float()).sum() \
        + (extras["weights"] * T(gw).float()).sum()
    loss_g.backward()
    scale = s64.grad.abs().max().item()
    assert (sg.grad.double().cpu() - s64.grad).abs().max().item() <= 2e-5 * scale
    assert torch.allclose(rg.grad.double().cpu(), r64.grad, rtol=1e-5, atol=1e-6)
    # the op-level entry point too, with gradients through trans and alphas
    sg2 = torch.tensor(sig, device=cuda, requires_grad=True)
    w2, tr2, al2 = n.render_weight_from_density(T(t0), T(t1), sg2, ray_indices=T(ri), n_rays=n_rays)
    ga, gt = rng.normal(size=S), rng.normal(size=S)
    ((w2 * T(gw).float()).sum() + (tr2 * T(gt).float()).sum() + (al2 * T(ga).float()).sum()).backward()
    s3 = torch.tensor(sig, dtype=torch.float64, requires_grad=True)
    _, _, _, w3, tr3, al3 = _chain64(torch.tensor(t0, dtype=torch.float64), torch.tensor(t1, dtype=torch.float64), s3,
                                     r64.detach(), starts, cnts, torch.tensor(bk))
    ((w3 * torch.tensor(gw)).sum() + (tr3 * torch.tensor(gt)).sum() + (al3 * torch.tensor(ga)).sum()).backward()
    assert (sg2.grad.double().cpu() - s3.grad).abs().max().item() <= 2e-5 * s3.grad.abs().max().item()


@pytest.mark.parametrize("alpha_thre", [0.0, 0.02])
def test_visibility_and_compaction(cuda, oracle, alpha_thre):
    from cnc_amd.backends import volrend_backend as K
    n_rays = 500
    starts, cnts, ri, t0, t1, sig, rgb = _ragged(n_rays, seed=21)
    sig = sig * 3
    w0, tr0, al0 = oracle.render_weight_from_density(t0, t1, sig, starts, cnts)
    cap = np.float32(0.015)
    thre = min(alpha_thre, float(cap))
    eps = 1e-2
    s_, c_, t0_, t1_, sig_ = _dev(cuda, starts, cnts, t0, t1, sig)
    mask, kept = K.render_visibility(s_, c_, sig_, t0_, t1_, early_stop_eps=eps, alpha_thre=alpha_thre,
                                     alpha_thre_cap=torch.tensor([cap], device=cuda))
    want = oracle.render_visibility(tr0, al0, eps, thre)
    got = mask.cpu().numpy().astype(bool)
    # a transmittance within an ulp of the threshold may fall on either side
    edge = np.abs(tr0 - np.float32(eps)) <= 4e-6 * eps
    if thre > 0:
        edge |= np.abs(al0 - np.float32(thre)) <= 3e-7
    assert np.array_equal(got[~edge], want[~edge]) and edge.sum() < 5
    assert 0.05 < got.mean() < 0.95
    assert np.array_equal(kept.cpu().numpy(), np.bincount(ri[got], minlength=n_rays))
    r2, a2, b2, ns, k2 = K.compact_samples(s_, c_, mask, kept, t0_, t1_)
    assert np.array_equal(r2.cpu().numpy(), ri[got])
    assert np.array_equal(a2.cpu().numpy(), t0[got]) and np.array_equal(b2.cpu().numpy(), t1[got])
    assert np.array_equal(ns.cpu().numpy(), np.cumsum(k2.cpu().numpy()) - k2.cpu().numpy())
    # the alpha route: same mask from alphas
    m2, _ = K.render_visibility(s_, c_, torch.as_tensor(al0, device=cuda), from_alpha=True, early_stop_eps=eps,
                                alpha_thre=alpha_thre, alpha_thre_cap=torch.tensor([cap], device=cuda))
    tr_a = oracle.segmented_scan(1 - al0, starts, cnts, exclusive=True, prod=True)
    want_a = oracle.render_visibility(tr_a, al0, eps, thre)
    edge_a = np.abs(tr_a - np.float32(eps)) <= 4e-6 * eps
    assert np.array_equal(m2.cpu().numpy().astype(bool)[~edge_a], want_a[~edge_a])


def test_ray_window_next_equals_the_tensor_expression(cuda):
    """cnc_ray_window_next (one step of the front-to-back sampler) against the op chain it replaced — done += take;
    alive = (exp(-sum sigma dt over the prefix) >= thr) & (done < counts); take = where(alive, min(left, w), 0) — over
    three windows, rays without samples and rays that finish early included."""
    from cnc_amd.backends import volrend_backend as K
    starts, cnts, ri, t0, t1, sig, rgb = _ragged(700, seed=33)
    sig = sig * 40                                              # many rays die inside the first windows
    s_, c_, t0_, t1_, sig_ = _dev(cuda, starts, cnts, t0, t1, sig)
    assert int((c_ == 0).sum()) > 0
    thr = 1e-2 * (1 - 1e-3)
    done = torch.zeros_like(c_)
    take = torch.empty_like(c_)
    done_ref, alive = torch.zeros_like(c_), c_ > 0
    died = 0
    for i, w in enumerate((4, 9, None)):
        K.ray_window_next(s_, c_, t0_, t1_, sig_, done, take, w, thr, first=i == 0)
        left = c_ - done_ref
        take_ref = torch.where(alive, left if w is None else left.clamp(max=w), torch.zeros_like(left))
        assert torch.equal(done, done_ref) and torch.equal(take, take_ref), i
        done_ref = done_ref + take_ref
        trans = K.ray_transmittance(s_, done_ref, t0_, t1_, sig_)
        now = (trans >= thr) & (done_ref < c_)
        died += int((alive & ~now & (done_ref < c_)).sum())
        alive = now
    assert died > 20 and int(take.sum()) > 0                    # both outcomes are exercised


def test_pack_bounds_and_pack_info(cuda):
    import cnc_amd.nerfacc as n
    ri = torch.tensor([0, 0, 1, 1, 1, 2, 2, 2, 2], device=cuda)
    assert n.pack_info(ri, n_rays=3).tolist() == [[0, 2], [2, 3], [5, 4]]             # pack.py docstring
    assert n.pack_info(ri, n_rays=5).tolist() == [[0, 2], [2, 3], [5, 4], [9, 0], [9, 0]]
    starts, cnts, ri_np, *_ = _ragged(300, seed=4)
    got = n.pack_info(torch.as_tensor(ri_np, device=cuda), n_rays=300).cpu().numpy()
    assert np.array_equal(got[:, 1], cnts) and np.array_equal(got[:, 0], starts)
    assert n.pack_info(torch.zeros(0, dtype=torch.int64, device=cuda), n_rays=2).tolist() == [[0, 0], [0, 0]]


def test_samples_from_intervals_both_layouts(cuda, oracle):
    """t_starts / t_ends / ray ids per sample from the interval edges == the reference's boolean indexing
    (vals[is_left], vals[is_right], ray_indices[is_valid]) for the two-pass layout and for the
    over-allocated layout with dead rays."""
    from cnc_amd import synthetic
    from cnc_amd.backends import nerfacc_cuda as C
    from cnc_amd.backends import volrend_backend as K
    o, d = synthetic.pinhole_rays(40, 40, 0.6911, 4.0, 0.7, 0.5)
    o, d = o.to(cuda), d.to(cuda)
    binaries = synthetic.ball_binaries(32, radius=1.0).to(cuda)
    binaries ^= torch.rand(binaries.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(1)) < 0.05
    aabbs = torch.tensor([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], device=cuda)
    n = o.shape[0]
    t0, t1, hit = C.ray_aabb_intersect(o, d, aabbs, -float("inf"), float("inf"), float("inf"))
    ts = torch.cat([t0, t1], -1).contiguous()
    ti = torch.arange(2, device=cuda).expand(n, 2).contiguous()
    near, far = torch.zeros(n, device=cuda), torch.full((n,), 1e10, device=cuda)
    for limit, over in ((-1, False), (7, True)):
        mask = torch.ones(n, dtype=torch.bool, device=cuda)
        if over:
            mask[::3] = False
        iv, sm, _ = C.traverse_grids(o, d, mask, binaries, aabbs, ts, ti, hit.contiguous(), near, far, 2e-2, 0.0,
                                     True, True, True, limit, over)
        ri, a, b, starts = K.samples_from_intervals(iv, sm.chunk_cnts)
        assert torch.equal(a, iv.vals[iv.is_left]) and torch.equal(b, iv.vals[iv.is_right])
        assert torch.equal(ri, sm.ray_indices[sm.is_valid])
        assert torch.equal(starts, torch.cumsum(sm.chunk_cnts, 0) - sm.chunk_cnts)
        assert a.shape[0] > 1000


def test_full_frame_identities(cuda):
    """BASELINE size (the bench's 800x800 frame, 68 M samples): size-independent identities of the fused
    compositing — per ray sum(w) = 1 - exp(-sum(sigma dt)) (telescoping), weights in [0, 1], transmittance
    non-increasing along a ray, colours of a constant-colour field = colour * opacity, and the backward's
    sum over a ray of d(opacity)/d(sigma_k) * ... matches the closed form d(opacity)/d(sigma_k) = dt_k * T_end."""
    import bench
    from cnc_amd.backends import nerfacc_cuda as C
    from cnc_amd.backends import volrend_backend as K
    w = bench.build_workload(cuda, 0)
    t_lo, t_hi, hit = C.ray_aabb_intersect(w["rays_o"], w["rays_d"], w["aabbs"], -float("inf"), float("inf"), float("inf"))
    ri, ts, te, starts, counts, _ = C.march_samples(w["rays_o"], w["rays_d"], None, w["binaries"], w["aabbs"],
                                                     torch.cat([t_lo, t_hi], -1), w["t_order"], hit, w["near"], w["far"],
                                                     bench.STEP_SIZE, 0.0)
    S, R = ts.shape[0], counts.shape[0]
    assert S > 6e7
    g = torch.Generator(device=cuda).manual_seed(1)
    sig = torch.rand(S, device=cuda, generator=g) * 8.0
    colour = torch.tensor([0.25, 0.5, 0.75], device=cuda)
    rgb = colour.expand(S, 3).contiguous()
    wts, tr, al, col, op, dep = K.volrend_forward(starts, counts, ts, te, sig, rgb)
    tau = torch.zeros(R, device=cuda, dtype=torch.float64).index_add_(0, ri, (sig * (te - ts)).double())
    want_op = 1.0 - torch.exp(-tau)
    assert float((op.view(-1).double() - want_op).abs().max()) < 2e-5
    assert float(wts.min()) >= 0.0 and float(wts.max()) <= 1.0 and float(al.max()) <= 1.0
    assert torch.allclose(col, op * colour, atol=2e-6)
    same_ray = ri[1:] == ri[:-1]
    # a parallel prefix sum associates differently per element: monotone up to a few ulps of the running sum
    assert bool((tr[1:][same_ray] <= tr[:-1][same_ray] * (1 + 4e-6)).all())
    first = torch.ones(S, dtype=torch.bool, device=cuda)
    first[1:] = ~same_ray
    assert bool((tr[first] == 1.0).all())
    # backward: d opacity / d sigma_k = dt_k * exp(-tau_ray)  (every sample of a ray shares T_end)
    g_sig, _ = K.volrend_backward(starts, counts, ts, te, None, wts, tr, al,
                                  grad_opacity=torch.ones(R, 1, device=cuda), want_grad_rgbs=False)
    want = (te - ts).double() * torch.exp(-tau)[ri]
    assert float((g_sig.double() - want).abs().max()) < 1e-7 + 2e-5 * float(want.max())
