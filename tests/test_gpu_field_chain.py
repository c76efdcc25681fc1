"""The gradient pass of the radiance field with its input-gradient chain as ONE kernel (cnc_field_backward_chain,
cnc_amd/csrc/field_bwd.hip; `_FieldChain` in cnc_amd/field.py) against the layer-by-layer autograd path it replaces
(library GEMMs, ReLU-backward passes, the post kernel's backward: itself pinned to the reference class's gradients at 2e-4
in tests/test_gpu_field_golden.py, which now runs through the chain by default).  Same forward values bit for bit (the
forward ops are the same); every parameter gradient within 2e-5 of its largest entry (three fp16 products per term on
operands scaled per tile: ~5e-7 per term)."""
import pytest
import torch

from test_gpu_field_fused import CONFIGS, _field, _inputs

pytestmark = pytest.mark.gpu


def _grads(f, x, d, wr, wd, chain, train=False):
    """chain False: layer by layer; True: library forward + the chain kernel; + train: the saving fused forward too."""
    f.fused_chain, f.fused_train = chain, train
    f._chain_supported = None
    for p in f.parameters():
        p.grad = None
    rgb, den = f(x, d)
    ((rgb * wr).sum() + (den * wd).sum()).backward()
    return rgb.detach(), den.detach(), {n: p.grad.clone() for n, p in f.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("n", [1, 33, 5000, 70001])
@pytest.mark.parametrize("cfg", ["f8_full", "f4_default", "f2_toy", "f8_toy", "f4_h64"])
def test_chain_gradients_equal_the_layer_by_layer_path(cuda, cfg, n):
    f = _field(cuda, CONFIGS[cfg], seed=8)
    x, d = _inputs(cuda, n, seed=n + 1)
    g = torch.Generator(device=cuda).manual_seed(3)
    # gradients of very different sizes from sample to sample, as a rendering loss produces them (weights w_i T_i)
    scale = torch.exp(torch.randn(n, 1, device=cuda, generator=g) * 3.0 - 6.0)
    wr = torch.randn(n, 3, device=cuda, generator=g) * scale
    wd = torch.randn(n, 1, device=cuda, generator=g) * scale * 0.1
    rgb0, den0, g0 = _grads(f, x, d, wr, wd, chain=False)
    rgb1, den1, g1 = _grads(f, x, d, wr, wd, chain=True)
    assert f._chain_supported, "the chain did not run"
    assert torch.equal(rgb0, rgb1) and torch.equal(den0, den1)
    assert set(g0) == set(g1) and len(g0) == 14             # 4 tables + 5 weights + 5 biases
    for name in g0:
        a, b = g1[name].double(), g0[name].double()
        scale_ = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * max(scale_, 1e-30), (name, float((a - b).abs().max()), scale_)
        assert scale_ > 0 or name.endswith("params")


def test_chain_handles_missing_output_gradients(cuda):
    """Only the colours, or only the density, enter the loss: the other gradient arrives as None."""
    f = _field(cuda, CONFIGS["f8_toy"], seed=2)
    x, d = _inputs(cuda, 3000, seed=4)
    for which in ("rgb", "density"):
        res = {}
        for chain in (False, True):
            f.fused_chain, f.fused_train, f._chain_supported = chain, False, None
            for p in f.parameters():
                p.grad = None
            rgb, den = f(x, d)
            (rgb.sum() if which == "rgb" else (den * den).sum()).backward()
            res[chain] = {n: p.grad.clone() for n, p in f.named_parameters() if p.grad is not None}
        for name, b in res[False].items():
            a = res[True][name]
            assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-30), (which, name)


@pytest.mark.parametrize("n", [1, 33, 5000, 70001])
@pytest.mark.parametrize("cfg", ["f8_full", "f4_default", "f2_toy", "f8_toy", "f4_h64"])
def test_fused_training_forward_and_its_gradients(cuda, cfg, n):
    """`_FieldTrain`: the gradient pass's forward as the saving form of the fused evaluator (cnc_field_save_t) — the
    outputs within the fused kernel's 2e-5 of the op chain's, every parameter gradient within 1e-4 of its largest entry
    of the layer-by-layer autograd path (the forward's 5e-7 per term carried through the chain)."""
    f = _field(cuda, CONFIGS[cfg], seed=8)
    x, d = _inputs(cuda, n, seed=n + 1)
    g = torch.Generator(device=cuda).manual_seed(3)
    scale = torch.exp(torch.randn(n, 1, device=cuda, generator=g) * 3.0 - 6.0)
    wr = torch.randn(n, 3, device=cuda, generator=g) * scale
    wd = torch.randn(n, 1, device=cuda, generator=g) * scale * 0.1
    rgb0, den0, g0 = _grads(f, x, d, wr, wd, chain=False)
    rgb1, den1, g1 = _grads(f, x, d, wr, wd, chain=True, train=True)
    assert f._train_ok(x, d), "the fused training forward did not run"
    assert float((rgb1 - rgb0).abs().max()) <= 2e-5
    assert float((den1 - den0).abs().max()) <= 2e-5 * max(1.0, float(den0.abs().max()))
    assert set(g0) == set(g1) and len(g0) == 14
    for name in g0:
        a, b = g1[name].double(), g0[name].double()
        scale_ = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * max(scale_, 1e-30), (name, float((a - b).abs().max()), scale_)
    assert not f.check_range_guard()


def test_range_guard_of_the_training_forward_is_looked_at_every_step_without_a_wait(cuda):
    """`snapshot_range_guard` behind a saving forward + `poll_range_guard` at the next step: nothing for an ordinary model;
    with hidden activations driven past fp16's 65504 the poll that follows the snapshot's arrival reports it, once, and the
    gradient pass leaves the fused kernel."""
    f = _field(cuda, CONFIGS["f8_full"], seed=8)
    x, d = _inputs(cuda, 4096, seed=2)
    wr, wd = torch.randn(4096, 3, device=cuda), torch.randn(4096, 1, device=cuda)
    for _ in range(3):                                   # three "steps": poll what the step before sent, send this one's
        _grads(f, x, d, wr, wd, chain=True, train=True)
        assert not f.poll_range_guard()
        f.snapshot_range_guard()
    torch.cuda.synchronize()
    assert not f.poll_range_guard() and f.fused_train
    with torch.no_grad():
        f.mlp_base.network[0].weight.mul_(300.0)
        f.mlp_base.network[0].bias.fill_(7.0e4)
        f.mlp_base.network[2].weight.mul_(1.0e-5)
    _grads(f, x, d, wr, wd, chain=True, train=True)
    f.snapshot_range_guard()
    torch.cuda.synchronize()
    with pytest.warns(UserWarning, match="left fp16's range"):
        assert f.poll_range_guard()
    assert not f.fused_train and not f.poll_range_guard()
    assert not f._train_ok(x, d)                         # the next gradient pass runs the fp32 library forward


def test_fused_training_forward_row_padding_and_saved_tensors(cuda):
    """Rows of padding (`_bucket_rows`): the saved matrices are finite there, the outputs of the live rows do not depend
    on the padding, and what the kernel saved is what the op chain computes (features bit for bit)."""
    f = _field(cuda, CONFIGS["f8_full"], seed=5)
    x, d = _inputs(cuda, 5000, seed=9)
    f._train_ok(x, d)
    ff = f._field_fused
    rgb_a, den_a, kept_a = ff.save_forward(x, d, 5000)
    rgb_b, den_b, kept_b = ff.save_forward(x, d, 5000 + 777)
    assert torch.equal(rgb_a, rgb_b[:5000]) and torch.equal(den_a, den_b[:5000])
    for k, v in kept_b.items():
        if k != "clips":
            assert v.shape[0] == 5777 and bool(torch.isfinite(v.float()).all()), k
            assert torch.equal(kept_a[k], v[:5000]), k
    assert float(den_b[5000:].abs().max()) == 0.0 and int(kept_b["selector"][5000:].sum()) == 0
    x_unit, selector, _ = f._prepare(x)
    assert torch.equal(kept_a["xyz"], x_unit) and torch.equal(kept_a["selector"], selector[:5000])
    assert torch.equal(kept_a["xy"], x_unit[:, :2]) and torch.equal(kept_a["xz"], x_unit[:, ::2]) and torch.equal(kept_a["yz"], x_unit[:, 1:])
    with torch.no_grad():
        feat = f.mlp_base.features_fused(x_unit)
    n_enc = sum(e.n_output_dims for e in f.mlp_base._encoders())
    assert torch.equal(kept_a["feat"][:, :n_enc], feat[:, :n_enc])
    assert float((kept_a["feat"][:, n_enc:feat.shape[1]] - feat[:, n_enc:]).abs().max()) <= 1e-6      # fast sincos
    assert float(kept_a["head_in"][:, 16].abs().max()) == 0.0


@pytest.mark.parametrize("n", [1, 63, 4097, 100001])
@pytest.mark.parametrize("shape", ["f8_full", "h64"])
def test_weight_gradient_kernel_against_float64(cuda, shape, n):
    """cnc_field_weight_grads on its own: dW_l = G_l^T A_l of random matrices whose rows differ by orders of magnitude
    (a rendering loss's gradients do), against the float64 product: within 2e-6 of each dW's largest entry."""
    import ctypes
    from cnc_amd import _lib
    H, geo, ldf, K0 = (160, 79, 256, 251) if shape == "f8_full" else (64, 15, 96, 91)
    g = torch.Generator(device=cuda).manual_seed(n)
    rnd = lambda *s: torch.randn(*s, device=cuda, generator=g)
    row_scale = torch.exp(rnd(n, 1) * 3.0 - 8.0)
    ld2, ldh = (1 + geo + 3) // 4 * 4, (17 + geo + 31) // 32 * 32
    Gs = [rnd(n, H) * row_scale, rnd(n, ld2) * row_scale, rnd(n, H) * row_scale * 10, rnd(n, H) * row_scale * 0.01, rnd(n, 4) * row_scale]
    As = [rnd(n, ldf), rnd(n, H).relu() * 3, rnd(n, ldh), rnd(n, H).relu() * 30, rnd(n, H).relu()]
    As[2][:, 16] = 0
    n_out, n_in = [H, 1 + geo, H, H, 3], [K0, H, 16 + geo, H, H]
    gmax = torch.tensor([float(G[:, :o].abs().max()) for G, o in zip(Gs, n_out)] + [0.0] * 3, device=cuda).view(torch.int32)
    d = _lib.FieldWGrad()
    d.N, d.head_gap_col, d.g_max = n, 16, gmax.data_ptr()
    out = [torch.full((o, i), float("nan"), device=cuda) for o, i in zip(n_out, n_in)]
    for l in range(5):
        d.G[l], d.ldG[l], d.n_out[l] = Gs[l].data_ptr(), Gs[l].shape[1], n_out[l]
        d.A[l], d.ldA[l], d.n_in[l] = As[l].data_ptr(), As[l].shape[1], n_in[l]
        d.dW[l], d.ld_dW[l] = out[l].data_ptr(), n_in[l]
    nbytes = ctypes.c_uint64(0)
    _lib.check(_lib.lib().cnc_field_weight_grads_workspace(ctypes.byref(d), ctypes.byref(nbytes)), "workspace")
    ws = torch.empty(nbytes.value // 4, device=cuda)
    d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes.value
    _lib.check(_lib.lib().cnc_field_weight_grads(ctypes.byref(d), _lib.stream(cuda)), "field_weight_grads")
    for l in range(5):
        ref = Gs[l][:, :n_out[l]].double().t() @ As[l].double()
        ref = torch.cat([ref[:, :16], ref[:, 17:17 + geo]], 1) if l == 2 else ref[:, :n_in[l]]
        err, scale = float((out[l].double() - ref).abs().max()), float(ref.abs().max())
        assert err <= 2e-6 * scale, (l, err, scale)


def _sh4_f64(d):
    """tiny-cuda-nn's published degree-4 polynomial (spherical_harmonics.h) in float64 (tests/golden/make_golden_field.py
    holds it against scipy's harmonics)."""
    import numpy as np
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return np.stack([
        np.full_like(x, 0.28209479177387814), -0.48860251190291987 * y, 0.48860251190291987 * z,
        -0.48860251190291987 * x, 1.0925484305920792 * xy, -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2, 0.59004358992664352 * y * (-3.0 * x2 + y2),
        2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)], -1)


def _float64_field(f, feats, xu, dirs, wr, wd):
    """The two networks of ngp.py:506-566 on given first-layer input rows, forward and backward, in float64 NumPy:
    (rgb, density, {parameter name: gradient of sum(rgb * wr) + sum(density * wd)}, {parameter name: the same sums over the
    ABSOLUTE values of every term — what an entry's rounding error is proportional to})."""
    import numpy as np
    mb, mh = f.mlp_base.network, f.mlp_head
    P = {n: t.detach().cpu().numpy().astype(np.float64) for n, t in
         (("W1", mb[0].weight), ("b1", mb[0].bias), ("W2", mb[2].weight), ("b2", mb[2].bias), ("W3", mh[0].weight),
          ("b3", mh[0].bias), ("W4", mh[2].weight), ("b4", mh[2].bias), ("W5", mh[4].weight), ("b5", mh[4].bias))}
    X = feats[:, :P["W1"].shape[1]].astype(np.float64)
    sel = np.all((xu > 0) & (xu < 1), axis=1).astype(np.float64)
    h1 = np.maximum(X @ P["W1"].T + P["b1"], 0.0)
    o2 = h1 @ P["W2"].T + P["b2"]
    raw = o2[:, 0]
    den = sel * np.exp(raw - 1.0)                                       # trunc_exp(x - 1) * selector
    d32 = ((dirs.astype(np.float32) + np.float32(1.0)) / np.float32(2.0)) * np.float32(2.0) - np.float32(1.0)
    hin = np.concatenate([_sh4_f64(d32.astype(np.float64)), o2[:, 1:]], axis=1)
    h3 = np.maximum(hin @ P["W3"].T + P["b3"], 0.0)
    h4 = np.maximum(h3 @ P["W4"].T + P["b4"], 0.0)
    rgb = 1.0 / (1.0 + np.exp(-(h4 @ P["W5"].T + P["b5"])))
    g5 = wr * rgb * (1.0 - rgb)
    g4 = (g5 @ P["W5"]) * (h4 > 0)
    g3 = (g4 @ P["W4"]) * (h3 > 0)
    ghin = g3 @ P["W3"]
    g2 = np.concatenate([(wd[:, 0] * sel * np.exp(np.minimum(raw - 1.0, 15.0)))[:, None], ghin[:, 16:]], axis=1)
    g1 = (g2 @ P["W2"]) * (h1 > 0)
    G = {"W5": g5.T @ h4, "b5": g5.sum(0), "W4": g4.T @ h3, "b4": g4.sum(0), "W3": g3.T @ hin, "b3": g3.sum(0),
         "W2": g2.T @ h1, "b2": g2.sum(0), "W1": g1.T @ X, "b1": g1.sum(0)}
    # the scale every entry's rounding error is proportional to: the same chain on absolute values (sum of |terms|)
    a5 = np.abs(g5)
    a4 = (a5 @ np.abs(P["W5"])) * (h4 > 0)
    a3 = (a4 @ np.abs(P["W4"])) * (h3 > 0)
    ahin = a3 @ np.abs(P["W3"])
    a2 = np.concatenate([np.abs(g2[:, :1]), ahin[:, 16:]], axis=1)
    a1 = (a2 @ np.abs(P["W2"])) * (h1 > 0)
    A = {"W5": a5.T @ np.abs(h4), "b5": a5.sum(0), "W4": a4.T @ np.abs(h3), "b4": a4.sum(0), "W3": a3.T @ np.abs(hin),
         "b3": a3.sum(0), "W2": a2.T @ np.abs(h1), "b2": a2.sum(0), "W1": a1.T @ np.abs(X), "b1": a1.sum(0)}
    return rgb, den, G, A


@pytest.mark.parametrize("train", [False, True], ids=["chain", "fused_training_forward"])
def test_gradient_pass_against_float64_at_full_size(cuda, train):
    """The full-size anchor of the gradient pass that is NOT the repo's own layer-by-layer path: the reference composition
    (12 x 3-D T = 2^19 + 3 x 4 planes T = 2^17, F = 8, H = 160), 2^16 samples, gradients spread over six orders of magnitude.
    The first layer's input rows come from the two-wave kernel's dump (bit-equal to oracle.grid_encode_forward:
    tests/test_gpu_field_fused.py); on them a float64 NumPy forward and backward of both MLPs (closed-form SH) is what rgb,
    density and every MLP parameter gradient of `cnc_field_backward_chain` + `cnc_field_weight_grads` (and, `train`, the
    saving fused forward in front of them) are held to — rgb / density at north_star's 1e-4; every gradient ENTRY against its
    own scale, the sum of the absolute values of its ~6e4 terms through the chain (an entry that is a near-cancellation of
    large terms cannot be relatively exact in any fp32 implementation; one that is not, is): |error| <= 1e-4 of that sum,
    <= 3e-4 of the tensor's largest entry, and the entries down to 1e-5 of the largest that are not cancellations (|sum| >=
    1 % of the sum of |terms|) relatively exact to 3e-3."""
    import numpy as np
    f = _field(cuda, CONFIGS["f8_full"], seed=31, sh_fp16_round=False)       # (the half-rounded SH has its own golden)
    n = 1 << 16
    x, d = _inputs(cuda, n, seed=77)
    g = torch.Generator(device=cuda).manual_seed(9)
    scale = torch.exp(torch.randn(n, 1, device=cuda, generator=g) * 3.0 - 6.0)
    wr = torch.randn(n, 3, device=cuda, generator=g) * scale
    wd = torch.randn(n, 1, device=cuda, generator=g) * scale * 0.1
    feats = torch.full((n, 256), float("nan"), device=cuda)
    with torch.no_grad():
        f.fused_field, f.fused_field_precision, f.fused_field_kernel = True, "f16x3", "w2"
        f.query_density(x[:8])
        f._field_fused(x, debug_features=feats)
    xu = ((x - f.aabb[:3]) / (f.aabb[3:] - f.aabb[:3])).cpu().numpy().astype(np.float32)
    rgb64, den64, G64, A64 = _float64_field(f, feats.cpu().numpy(), xu, d.cpu().numpy(), wr.cpu().numpy().astype(np.float64),
                                       wd.cpu().numpy().astype(np.float64))
    rgb, den, grads = _grads(f, x, d, wr, wd, chain=True, train=train)
    assert f._chain_supported and (f.fused_train or not train)
    assert np.abs(rgb.cpu().numpy() - rgb64).max() <= 1e-4
    assert np.abs(den.cpu().numpy()[:, 0] - den64).max() <= 1e-4 * den64.max()
    names = {"W1": "mlp_base.network.0.weight", "b1": "mlp_base.network.0.bias", "W2": "mlp_base.network.2.weight",
             "b2": "mlp_base.network.2.bias", "W3": "mlp_head.0.weight", "b3": "mlp_head.0.bias", "W4": "mlp_head.2.weight",
             "b4": "mlp_head.2.bias", "W5": "mlp_head.4.weight", "b5": "mlp_head.4.bias"}
    worst = {}
    for k, name in names.items():
        got = grads[name].cpu().numpy().astype(np.float64)
        want = G64[k][:, :got.shape[1]] if got.ndim == 2 else G64[k]
        assert got.shape == want.shape, (k, got.shape, want.shape)
        big = float(np.abs(want).max())
        absmax = float(np.abs(got - want).max()) / big
        scale_ = A64[k][:, :got.shape[1]] if got.ndim == 2 else A64[k]
        ratio = np.abs(got - want) / (scale_ + 1e-2 * big)       # (+ 1 % of the largest entry: units that are dead for all but
                                                                 # a few samples have a scale of next to nothing)
        # entries that are not cancellations (|sum| >= 1 % of the sum of |terms|), down to 1e-5 of the tensor's largest
        m = (np.abs(want) >= 1e-5 * big) & (np.abs(want) >= 1e-2 * scale_)
        rel = np.abs(got - want)[m] / np.abs(want)[m]
        worst[k] = (absmax, float(ratio.max()), float(rel.max()) if m.any() else 0.0, int(m.sum()), int(m.size))
    print("MLP gradients against float64: (max |error| / largest entry, max |error| / sum|terms|, max relative error of the non-cancelling entries, their count, entries):", worst)
    for k, w in worst.items():
        # measured (round 6, both forms): at most 1.2e-4 of the tensor's largest entry, 3.6e-5 of an entry's own sum of |terms| (+ 1 % of the largest),
        # 1.0e-3 relative on the non-cancelling entries — ReLU units whose pre-activation is within rounding of zero switch
        # between the float64 and the product's forward, which is what the largest of these are made of, not the products
        assert w[0] <= 3e-4 and w[1] <= 1e-4 and w[2] <= 3e-3, (k, w)
