"""End-to-end slice on the GPU: train a toy field on the procedural scene with the entropy loss,
evaluate, encode to .b files, wipe + decode, evaluate again (the reference's protocol,
train_CNC_nerf_synthetic.py:302-506)."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(tmp_path, **kw):
    from cnc_amd.trainer import TrainConfig
    base = dict(lmbda=2e-3, Pg_level=5, Pg_level_2D=3, log2_hashmap_size=12, log2_hashmap_size_2D=9,
                sample_num=3000, max_context_layer_num=3, n_features=2, n_neurons=32,
                resolutions_list=(10, 14, 18, 26, 34), resolutions_list_2D=(18, 34, 66),
                skip_levels_3D=(0, 1, 2), skip_levels_2D=(0,), max_steps=150, init_batch_size=512,
                target_sample_batch_size=1 << 14, grid_resolution=16, render_step_size=2e-2,
                milestones=(100, 130), warmup_iters=20, test_views=2, image_size=48,
                out_dir=str(tmp_path / "bits"), log_every=50)
    base.update(kw)
    return TrainConfig(**base)


def test_train_encode_decode_roundtrip(cuda, tmp_path):
    from cnc_amd.trainer import Trainer
    tr = Trainer(_cfg(tmp_path), device=cuda)
    first = tr.train_step(0)
    assert first is not None and first["n_rendering_samples"] > 0 and first["bpp"] > 0
    last = tr.train(steps=150, log=None)
    assert last["mse"] < first["mse"] * 0.5                      # it learns
    # no table went NaN on the way (a level whose entries all share a sign has Pg = 0 or 1: its zero-order
    # bit count must stay finite, and so must its gradient)
    assert all(torch.isfinite(p).all() for p in list(tr.field.parameters()) + list(tr.context.parameters()))
    assert last["num_rays"] != 512                                # adaptive ray budget kicked in
    psnr_before = tr.evaluate()
    assert psnr_before > 12.0
    Pgs, est_MB, coded_MB, prefix = tr.encode()
    files = sorted(f for f in os.listdir(os.path.dirname(prefix)) if f.endswith(".b"))
    # 3 skip + 2 coded 3-D levels, 3 planes x (1 skip + 2 coded) levels
    assert len(files) == 5 + 9
    on_disk = sum(os.path.getsize(os.path.join(os.path.dirname(prefix), f)) for f in files)
    assert abs(on_disk / 1024 / 1024 - coded_MB) < 1e-9
    assert coded_MB < 1.05 * est_MB + 2e-4                        # coded size tracks the entropy estimate
    e = tr.field.mlp_base
    orig = [torch.where(t.params.data >= 0, 1.0, -1.0) for t in (e.encoding_xyz, e.encoding_xy, e.encoding_xz, e.encoding_yz)]
    tr.decode_into_field(Pgs, prefix)
    for t, o in zip((e.encoding_xyz, e.encoding_xy, e.encoding_xz, e.encoding_yz), orig):
        dec = t.params.data
        assert torch.all(dec.abs() == 1)
        uncoded = (dec == 1).all(dim=1)                           # never-written rows keep the init value
        assert torch.equal(dec[~uncoded], o[~uncoded])            # every coded row decodes exactly
    psnr_after = tr.evaluate()
    assert abs(psnr_after - psnr_before) < 0.3, (psnr_before, psnr_after)
    sizes = tr.sizes_MB(coded_MB)
    assert sizes["total"] > sizes["embeddings"] > 0


def test_single_file_container_roundtrip(cuda, tmp_path):
    """train a little -> one .cnc file -> a FRESH trainer decodes it -> same render quality as the
    sender's field with a 13-bit MLP; the file size is the real size(KB)."""
    from cnc_amd.trainer import Trainer
    a = Trainer(_cfg(tmp_path), device=cuda)
    a.train(steps=120, log=None)
    psnr_a = a.evaluate()
    info = a.save_container(str(tmp_path / "scene.cnc"))
    assert os.path.getsize(tmp_path / "scene.cnc") == int(round(info["file_KB"] * 1024))
    assert info["file_KB"] > info["embeddings_KB"] > 0
    b = Trainer(_cfg(tmp_path, seed=7), device=cuda)          # different init: nothing shared but the config
    assert abs(b.evaluate() - psnr_a) > 3.0
    b.load_container(str(tmp_path / "scene.cnc"))
    assert torch.equal(b.estimator.binaries, a.estimator.binaries)
    psnr_b = b.evaluate()
    assert abs(psnr_b - psnr_a) < 0.5, (psnr_a, psnr_b)


def test_field_shapes_and_sh(cuda):
    from cnc_amd.field import NGPRadianceField_mygrid_2D3D, SHEncoding
    f = NGPRadianceField_mygrid_2D3D(aabb=[-1.5] * 3 + [1.5] * 3, n_features_per_level=8, n_neurons=160,
                                     resolutions_list=(18, 24, 33), log2_hashmap_size=12,
                                     resolutions_list_2D=(130, 258), log2_hashmap_size_2D=10).to(cuda)
    assert f.geo_feat_dim == 79
    assert f.mlp_base.network[0].in_features == 3 * 8 + 3 * 2 * 8 + 63
    assert f.mlp_head[0].in_features == 16 + 79
    p = torch.rand(100, 3, device=cuda) * 3 - 1.5
    d = torch.nn.functional.normalize(torch.randn(100, 3, device=cuda), dim=-1)
    rgb, sigma = f(p, d)
    assert rgb.shape == (100, 3) and sigma.shape == (100, 1)
    assert (rgb >= 0).all() and (rgb <= 1).all() and (sigma >= 0).all()
    # SH: orthonormal basis -> Monte-Carlo Gram matrix ~ identity / (4 pi) * 4 pi
    sh = SHEncoding()
    v = torch.nn.functional.normalize(torch.randn(400000, 3, device=cuda), dim=-1)
    Y = sh((v + 1) / 2)
    gram = (Y.T @ Y) / v.shape[0] * 4 * math.pi
    assert torch.allclose(gram, torch.eye(16, device=cuda), atol=0.03)


@pytest.mark.parametrize("F", [1, 2, 4, 8])
def test_fused_field_features_match_cat(cuda, F):
    """Encoders writing point-major into the MLP input (cnc_grid_encode_*'s out_ld/out_col) give the
    same feature matrix as permute + cat (ngp.py:111,631-642), and the same parameter gradients."""
    from cnc_amd.field import NGPRadianceField_mygrid_2D3D
    torch.manual_seed(3)
    kw = dict(aabb=[-1.5] * 3 + [1.5] * 3, n_features_per_level=F, n_neurons=32,
              resolutions_list=(10, 18, 33, 70), log2_hashmap_size=12,
              resolutions_list_2D=(34, 130, 258), log2_hashmap_size_2D=10)
    a = NGPRadianceField_mygrid_2D3D(fused_features=True, **kw).to(cuda)
    b = NGPRadianceField_mygrid_2D3D(fused_features=False, **kw).to(cuda)
    with torch.no_grad():
        for p in a.parameters():
            p.uniform_(-1.5, 1.5)       # some |p| > 1: STE mask in play
    b.load_state_dict(a.state_dict())
    x = torch.rand(5000, 3, device=cuda)
    assert a.mlp_base._can_fuse(x) and not b.mlp_base._can_fuse(x)
    fa, fb = a.mlp_base.features_fused(x), b.mlp_base.features(x)
    w = fb.shape[1]
    assert fa.shape[1] % 4 == 0 and torch.equal(fa[:, :w], fb) and (fa[:, w:] == 0).all()
    ya, yb = a.mlp_base(x), b.mlp_base(x)
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None:
            assert pb.grad is None
            continue
        scale = pb.grad.abs().max().clamp_min(1e-6)
        assert (pa.grad - pb.grad).abs().max() <= 2e-4 * scale, n


def test_no_grad_field_uses_fused_epilogue_and_matches(cuda):
    """Evaluation path (bias + ReLU in the GEMM epilogue) vs the training path (separate ReLU):
    same rgb / density within the north-star's 1e-4."""
    from cnc_amd.field import NGPRadianceField_mygrid_2D3D
    torch.manual_seed(9)
    f = NGPRadianceField_mygrid_2D3D(aabb=[-1.5] * 3 + [1.5] * 3, n_features_per_level=8, n_neurons=160,
                                     resolutions_list=(18, 24, 33, 70), log2_hashmap_size=14,
                                     resolutions_list_2D=(130, 258), log2_hashmap_size_2D=12).to(cuda)
    with torch.no_grad():
        for p in f.parameters():
            if p.dim() == 2 and p.shape[1] <= 8:
                p.uniform_(-1, 1)
    x = torch.rand(20000, 3, device=cuda) * 2.6 - 1.3
    d = torch.nn.functional.normalize(torch.randn(20000, 3, device=cuda), dim=-1)
    rgb_a, sig_a = f(x, d)
    with torch.no_grad():
        rgb_b, sig_b = f(x, d)
    assert (rgb_a - rgb_b).abs().max() <= 1e-4
    assert ((sig_a - sig_b).abs() <= 1e-4 * (1 + sig_a.abs())).all()
    # and against plain nn.Sequential modules (separate ReLU), forward and parameter gradients
    import copy
    ref_base, ref_head = copy.deepcopy(f.mlp_base.network), copy.deepcopy(f.mlp_head)
    feat = f.mlp_base.features(((x + 1.5) / 3.0).clamp(0, 1)).detach()
    from cnc_amd.mlp import run_layers
    ya = run_layers(f.mlp_base.network, feat)
    yb = ref_base(feat)
    assert (ya - yb).abs().max() <= 1e-4 * (1 + yb.abs().max())
    g = torch.randn_like(ya)
    ya.backward(g); yb.backward(g)
    for pa, pb in zip(f.mlp_base.network.parameters(), ref_base.parameters()):
        assert (pa.grad - pb.grad).abs().max() <= 1e-4 * (1 + pb.grad.abs().max())


def test_density_only_query_matches_the_full_head(cuda):
    """`query_density` without gradients evaluates unit 0 of the base MLP's last layer only (the sampler's visibility
    pass is 6-8x the samples of the gradient pass): same densities as the full 1 + geo_feat_dim outputs."""
    from cnc_amd.field import NGPRadianceField_mygrid_2D3D
    torch.manual_seed(11)
    f = NGPRadianceField_mygrid_2D3D(aabb=[-1.5] * 3 + [1.5] * 3, n_features_per_level=8, n_neurons=160).to(cuda)
    with torch.no_grad():
        for p in f.parameters():
            if p.dim() == 2 and p.shape[1] == 8:
                p.uniform_(-1, 1)
    x = (torch.rand(200000, 3, device=cuda) * 3.4 - 1.7)          # some outside the box: selector
    with torch.no_grad():
        only = f.query_density(x)
        full, _ = f.query_density(x, return_feat=True)
    assert only.shape == full.shape == (200000, 1)
    assert torch.allclose(only, full, rtol=2e-5, atol=1e-7)
    assert bool((only[(x.abs() > 1.5).any(-1)] == 0).all())


def test_front_to_back_density_gives_the_same_samples(cuda, tmp_path):
    """The sampler's depth-window evaluation of the density (a ray leaves once its transmittance is below
    early_stop_eps) returns the samples of the one-shot evaluation, and evaluates far fewer."""
    from cnc_amd.nerfacc.estimators.occ_grid import OccGridEstimator
    from cnc_amd.render import _FieldOnRays
    from cnc_amd.trainer import Trainer
    tr = Trainer(_cfg(tmp_path, lmbda=0.0, n_neurons=64), device=cuda)      # (64: a width the one-kernel evaluator has)
    tr.train(steps=150, log=None)                      # a surface has formed
    tr.field.eval(); tr.estimator.eval()
    view = tr.dataset.view(0)
    o, d = view["rays"].origins.reshape(-1, 3), view["rays"].viewdirs.reshape(-1, 3)
    seen = []

    class Counting(_FieldOnRays):
        def density(self, t_starts, t_ends, ray_indices):
            seen.append(t_starts.shape[0])
            return super().density(t_starts, t_ends, ray_indices)
    fn = Counting(tr.field, o, d, with_positions=False)
    kw = dict(sigma_fn=fn.density, near_plane=tr.cfg.near_plane, render_step_size=tr.cfg.render_step_size,
              stratified=False, cone_angle=tr.cfg.cone_angle, alpha_thre=tr.cfg.alpha_thre)
    outs = []
    tr.estimator._COUNTED_WINDOWS = False              # windows through the `sigma_fn` callback (counted below)
    for mode in (False, True):                         # the public switch of `sampling`
        seen.clear()
        with torch.no_grad():
            ri, ts, te = tr.estimator.sampling(o, d, front_to_back=mode, **kw)
        outs.append((ri.clone(), ts.clone(), te.clone(), sum(seen), len(seen)))
    (ri_a, ts_a, te_a, n_a, calls_a), (ri_b, ts_b, te_b, n_b, calls_b) = outs
    # the windows with their sample counts left on the device (no host round trip per window): the callback is not used,
    # the field evaluates buffers of a bound's size up to a count it reads on the device — the SAME samples as the windows
    # through the callback, exactly (a row's density does not depend on the batch it is evaluated in)
    tr.estimator._COUNTED_WINDOWS = True
    seen.clear()
    with torch.no_grad():
        assert fn.density_windows() is not None
        ri_c, ts_c, te_c = tr.estimator.sampling(o, d, front_to_back=True, **kw)
    assert not seen
    assert ri_c.shape == ri_b.shape and torch.equal(ri_c, ri_b) and torch.equal(ts_c, ts_b) and torch.equal(te_c, te_b)
    assert calls_a == 1 and 1 < calls_b <= 3
    assert n_b < 0.7 * n_a                              # most of the marched samples are never evaluated
    # a GEMM row may round differently in a different batch: allow a handful of samples at the threshold to differ
    assert abs(ts_a.shape[0] - ts_b.shape[0]) <= 1e-4 * ts_a.shape[0] + 2
    if ts_a.shape[0] == ts_b.shape[0]:
        same = (ri_a == ri_b) & (ts_a == ts_b) & (te_a == te_b)
        assert float(same.float().mean()) > 0.9999


def test_threaded_context_pass_equals_the_sequential_schedule(cuda, tmp_path):
    """The default single-process schedule (context forward + backward on a side stream, issued from a second host
    thread next to the render pass) against the sequential one and against one stream / one thread: same loss, same
    rate, same sample counts, same parameters after several steps — up to the order of float atomics.  The context
    pass draws from its own generator here, so the values do not depend on which thread draws first."""
    from cnc_amd.trainer import Trainer

    def run(mode):
        tr = Trainer(_cfg(tmp_path, seed=3), device=cuda)
        g = torch.Generator(device=cuda)
        g.manual_seed(77)
        tr.context.rand_like = lambda t: torch.rand(t.shape, generator=g, device=t.device, dtype=t.dtype)
        if mode == "plain":
            tr.ctx_stream, tr.ctx_thread = None, False
        elif mode == "stream":
            tr.ctx_thread = False
        else:
            assert tr.ctx_thread and tr.ctx_stream is not None            # the default
        out = [tr.train_step(s) for s in range(20)]
        torch.cuda.synchronize()
        return out, [p.detach().clone() for p in list(tr.field.parameters()) + list(tr.context.parameters())]

    ref, ref_p = run("plain")
    for mode in ("stream", "thread"):
        got, got_p = run(mode)
        for a, b in zip(ref[:4], got[:4]):             # before the chaotic regime of binarised tables
            assert a["n_rendering_samples"] == b["n_rendering_samples"] and a["num_rays"] == b["num_rays"]
            assert abs(a["mse"] - b["mse"]) <= 1e-5 * max(a["mse"], 1e-6) + 1e-9, mode
            assert abs(a["bpp"] - b["bpp"]) <= 1e-5 * a["bpp"], mode
        assert abs(ref[-1]["bpp"] - got[-1]["bpp"]) <= 0.05 * ref[-1]["bpp"], mode
        assert abs(ref[-1]["mse"] - got[-1]["mse"]) <= 0.2 * ref[-1]["mse"], mode
        assert all(torch.isfinite(p).all() for p in got_p)


@pytest.mark.parametrize("mode", ["plain", "stream", "thread"])
def test_gradient_sinks_give_the_gradients_autograd_gives(cuda, tmp_path, mode):
    """The per-step gradient sinks (cnc_amd._gradsink: encoder scatters and context-head weight gradients add into one
    buffer per parameter and pass, `.grad` gets them once) against plain autograd (a fresh zero-filled tensor per
    call, summed by the engine): the same gradient of every parameter after one step's backward passes, up to the
    order of float atomics — in all three schedules."""
    from cnc_amd.trainer import Trainer
    grads = {}
    for sinks in (True, False):
        tr = Trainer(_cfg(tmp_path, seed=5), device=cuda)
        g = torch.Generator(device=cuda)
        g.manual_seed(78)
        tr.context.rand_like = lambda t: torch.rand(t.shape, generator=g, device=t.device, dtype=t.dtype)
        if mode == "plain":
            tr.ctx_stream, tr.ctx_thread = None, False
        elif mode == "stream":
            tr.ctx_thread = False
        assert tr.sink_render is not None and tr.sink_ctx is not None
        if not sinks:
            tr.sink_render = tr.sink_ctx = None
        tr.fused_table_adam = False           # the tables' pieces into `.grad` (what this test reads), library step
        tr.train_step(0)                      # same initial state, same draws: the gradients of this one step
        torch.cuda.synchronize()
        named = list(tr.field.named_parameters()) + [("ctx." + n, p) for n, p in tr.context.named_parameters()]
        grads[sinks] = {n: p.grad.detach().clone() for n, p in named if p.grad is not None}
        assert len(grads[sinks]) == len(named)
    for n, want in grads[False].items():
        got = grads[True][n]
        scale = float(want.abs().max())
        assert scale > 0, n
        assert float((got - want).abs().max()) <= 2e-4 * scale, (n, float((got - want).abs().max()), scale)


def _entropy_pass_gradients(tr, step, params):
    """One entropy pass (forward + backward through `Trainer._context_pass`) on the trainer's current state: (bits per
    parameter, MB, the gradient of every parameter)."""
    for p in params:
        p.grad = None
    for s in (tr.sink_render, tr.sink_ctx):
        s.zero()
    for enc in tr.field.mlp_base._encoders():
        enc._bit_plane(enc.params)
    torch.manual_seed(5)                                  # the 3-D half's window draw
    main = torch.cuda.current_stream()
    tr._ensure_planes_graph(step, None)
    bpp, mb, done, _ = tr._context_pass(step, main.record_event())
    main.wait_event(done)
    tr.sink_ctx.flush()
    if tr._planes_replayed:
        tr.planes_graph.flush()
    torch.cuda.synchronize()
    return float(bpp), float(mb), [None if p.grad is None else p.grad.clone() for p in params]


def test_planes_graph_and_planes_stream_equal_the_one_stream_entropy_pass(cuda, tmp_path):
    """The planes' half of the entropy pass on its own stream (`stream_2D`), and as a captured HIP graph replayed across
    parameter updates (`cnc_amd._planes_graph`), against the op-by-op pass on one stream — on the same state: the same
    bits and the same gradient of every parameter up to the order of float atomics.  Then training goes on through the
    graph: it is recorded once per occupancy refresh, replayed on every other step, and nothing goes non-finite."""
    from cnc_amd.trainer import Trainer
    tr = Trainer(_cfg(tmp_path, seed=11), device=cuda)
    pg = tr.planes_graph
    assert pg is not None and tr.ctx_stream_2D is not None
    for step in range(20):                                # the graph: recorded at step 17, replayed 17, 18, 19
        tr.train_step(step, want_stats=False)
    torch.cuda.synchronize()
    assert pg.captures == 1 and pg.replays == 3
    params = list(tr.field.parameters()) + list(tr.context.parameters())
    names = [n for n, _ in tr.field.named_parameters()] + ["ctx." + n for n, _ in tr.context.named_parameters()]
    s2 = tr.ctx_stream_2D
    tr.planes_graph, tr.ctx_stream_2D = None, None
    want = _entropy_pass_gradients(tr, 20, params)       # one stream, op by op
    tr.ctx_stream_2D = s2
    two = _entropy_pass_gradients(tr, 20, params)        # the planes' half on its own stream
    tr.planes_graph = pg
    graph = _entropy_pass_gradients(tr, 20, params)      # ... as the graph recorded three updates ago
    assert pg.captures == 1 and pg.replays == 4
    for what, got in (("stream", two), ("graph", graph)):
        assert abs(got[0] - want[0]) <= 1e-6 * want[0] and abs(got[1] - want[1]) <= 1e-6 * want[1], what
        for n, a, b in zip(names, want[2], got[2]):
            assert (a is None) == (b is None), (what, n)
            if a is not None:
                assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-30), (what, n)
    for step in range(20, 70):
        s = tr.train_step(step, want_stats=step % 10 == 0)
        if step % 10 == 0:
            assert math.isfinite(s["bpp"]) and math.isfinite(s["mse"])
    torch.cuda.synchronize()
    refreshes = tr.context.refresh_stats["refreshes"] - tr.context.refresh_stats["skipped"]
    assert 2 <= pg.captures <= refreshes and pg.replays >= 4 + 50 - 4            # every non-refresh step replayed
    assert all(bool(torch.isfinite(p).all()) for p in params)
