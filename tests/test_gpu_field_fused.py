"""The gradient-free radiance field as one kernel (cnc_field_fused_forward, cnc_amd/csrc/field_fused.hip) against the
chain it replaces — encoder launches into a [N, 255] matrix, library GEMMs, glue kernels (ngp.py:506-547) — which is
itself pinned to the reference class (tests/test_gpu_field_golden.py; that test's no-grad leg now runs the fused
kernel against the reference's numbers directly).  Tolerance: north_star's 1e-4 of the tensor's scale (the two paths
sum each layer's products in different orders; the encoder features inside the kernel are bit-identical)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
AABB = [-1.5] * 3 + [1.5] * 3

CONFIGS = {
    # the reference composition at F = 8 (12 x 3-D + 3 x 4 planes, H = 160, geo 79): K0 = 255
    "f8_full": dict(n_features_per_level=8, n_neurons=160, resolutions_list=(18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514),
                    log2_hashmap_size=19, resolutions_list_2D=(130, 258, 514, 1026), log2_hashmap_size_2D=17),
    # the drivers' default F = 4 (geo 39: two of the three second-layer tiles are padding)
    "f4_default": dict(n_features_per_level=4, n_neurons=160, resolutions_list=(18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514),
                       log2_hashmap_size=19, resolutions_list_2D=(130, 258, 514, 1026), log2_hashmap_size_2D=17),
    # the toy shapes of the goldens: unit columns that end inside a 32-column chunk (144 and 36), H = 64
    "f8_toy": dict(n_features_per_level=8, n_neurons=160, resolutions_list=(6, 9, 14, 20, 26, 34), log2_hashmap_size=10,
                   resolutions_list_2D=(10, 18, 34, 66), log2_hashmap_size_2D=9),
    "f2_toy": dict(n_features_per_level=2, n_neurons=64, resolutions_list=(6, 9, 14, 20, 26, 34), log2_hashmap_size=10,
                   resolutions_list_2D=(10, 18, 34, 66), log2_hashmap_size_2D=9),
    "f4_h64": dict(n_features_per_level=4, n_neurons=64, resolutions_list=(6, 9, 14, 20, 26), log2_hashmap_size=10,
                   resolutions_list_2D=(10, 18, 34), log2_hashmap_size_2D=9),
}


def _field(cuda, kw, seed=0, **extra):
    from cnc_amd.field import NGPRadianceField_mygrid_2D3D
    torch.manual_seed(seed)
    f = NGPRadianceField_mygrid_2D3D(aabb=AABB, **kw, **extra).to(cuda)
    with torch.no_grad():
        for e in f.mlp_base._encoders():
            e.params.uniform_(-1, 1)
        f.mlp_base.network[2].bias[0] = 1.5          # densities of order one, both sides of the ReLU in play
    return f


def _inputs(cuda, n, seed):
    g = torch.Generator(device=cuda).manual_seed(seed)
    x = torch.rand(n, 3, device=cuda, generator=g) * 3.2 - 1.6          # some points outside the box
    if n >= 8:
        x[0] = torch.tensor([-1.5, 0.0, 0.0])            # on the box: selector 0, features of the boundary
        x[1] = torch.tensor([1.5, 1.5, 1.5])
        x[2] = torch.tensor([0.0, 0.0, 1.7])             # z outside: the xy plane still contributes features
        x[3] = torch.tensor([0.0, 0.0, 0.0])
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=cuda, generator=g), dim=-1)
    return x, d


def _close(got, want, tol, what):
    scale = float(want.abs().max())
    err = float((got.double() - want.double()).abs().max())
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert err <= tol * max(scale, 1e-30), (what, err, scale)


# "f32": each layer an exact fp32 fmaf chain (v_mfma_f32_32x32x2_f32); "f16x3" (the default): three fp16 products per
# term — the bound below is the SAME 1e-4 for both, and the fp16 form is additionally held to 2e-5.  "w2" (the default):
# two cooperating waves per 32-sample tile (field_fused2.hip, v_mfma_f32_16x16x32_f16); "w1": one wave per tile.
PRECISIONS = pytest.mark.parametrize("precision", ["f16x3-w2", "f16x3-w1", "f32"])


def _select(f, precision):
    f.fused_field_precision = precision.split("-")[0]
    f.fused_field_kernel = precision.split("-")[1] if "-" in precision else "w1"


@PRECISIONS
@pytest.mark.parametrize("n", [1, 31, 32, 33, 1000, 70001])
@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_fused_field_equals_the_chain(cuda, cfg, n, precision):
    from cnc_amd.field import FusedFieldForward
    f = _field(cuda, CONFIGS[cfg], seed=3)
    _select(f, precision)
    assert FusedFieldForward.supported(f)
    x, d = _inputs(cuda, n, seed=n)
    with torch.no_grad():
        f.fused_field = False
        rgb0, sig0 = f(x, d)
        den0 = f.query_density(x)
        f.fused_field = True
        rgb1, sig1 = f(x, d)
        den1 = f.query_density(x)
    assert f._field_fused, "the fused kernel did not run"
    tol = 2e-5 if precision.startswith("f16x3") else 1e-4
    assert not (precision.startswith("f16x3") and f._field_fused.range_guard_fired())      # in range: the fp16 kernel's own numbers
    _close(den1, den0, tol, "density (density-only kernel)")
    _close(sig1, sig0, tol, "density (colour kernel)")
    _close(rgb1, rgb0, tol, "rgb")
    # outside the box the density is exactly zero in both (selector, ngp.py:524), inside it is positive
    assert torch.equal(den1 == 0, den0 == 0) and torch.equal(sig1 == 0, sig0 == 0)
    if n >= 1000:
        assert 0.05 < float((den0 > 0).float().mean()) < 0.95 and float(den0.max()) > 0.5
        assert float(rgb0.std()) > 1e-3


@PRECISIONS
def test_fused_field_at_full_size_and_after_a_weight_update(cuda, precision):
    """2^20 samples of the reference composition: the fused kernels against the chain; then an optimiser-style in-place
    update of every parameter — the packed weights and the sign planes must follow it."""
    f = _field(cuda, CONFIGS["f8_full"], seed=5)
    _select(f, precision)
    n = 1 << 20
    x, d = _inputs(cuda, n, seed=11)
    for round_ in range(2):
        with torch.no_grad():
            f.fused_field = False
            den0 = f.query_density(x)
            rgb0, _ = f(x[: 1 << 18], d[: 1 << 18])
            f.fused_field = True
            den1 = f.query_density(x)
            rgb1, sig1 = f(x[: 1 << 18], d[: 1 << 18])
        _close(den1, den0, 1e-4, f"density, round {round_}")
        _close(rgb1, rgb0, 1e-4, f"rgb, round {round_}")
        _close(sig1, den0[: 1 << 18], 1e-4, f"density from the colour kernel, round {round_}")
        with torch.no_grad():
            for p in f.parameters():
                p.mul_(-0.9).add_(0.01)              # bumps _version: caches keyed on it must refresh
    assert float((den1 - den0).abs().max()) >= 0.0


def test_fused_field_with_gradients_and_outside_its_shapes(cuda):
    from cnc_amd.field import FusedFieldForward
    f = _field(cuda, CONFIGS["f2_toy"], seed=1)
    x, d = _inputs(cuda, 500, seed=2)
    f.fused_train = False
    rgb, sig = f(x, d)                              # gradients enabled, the saving kernel switched off: the op chain
    assert rgb.requires_grad and not f._field_fused
    (rgb.sum() + sig.sum()).backward()
    f.fused_train = True
    rgb, sig = f(x, d)                              # ... switched on (the default): the fused kernel in its saving form
    assert rgb.requires_grad and f._field_fused and f._field_fused._train_calls
    (rgb.sum() + sig.sum()).backward()
    x.requires_grad_(True)                          # gradients with respect to positions: the op chain again
    assert not f._train_ok(x, d)
    x.requires_grad_(False)
    g = _field(cuda, dict(CONFIGS["f2_toy"], n_neurons=48), seed=1)
    assert not FusedFieldForward.supported(g)
    with torch.no_grad():
        den = g.query_density(x)                    # falls back to the chain, silently correct
    assert den.shape == (500, 1) and g._field_fused is False


def test_sh_half_rounding_reaches_the_fused_kernel(cuda):
    """sh_fp16_round on / off changes the colours of the fused kernel exactly as it changes the chain's."""
    a = _field(cuda, CONFIGS["f8_toy"], seed=7)
    b = _field(cuda, CONFIGS["f8_toy"], seed=7, sh_fp16_round=False)
    b.load_state_dict(a.state_dict())
    x, d = _inputs(cuda, 4000, seed=9)
    out = {}
    with torch.no_grad():
        for name, f in (("half", a), ("float", b)):
            for fused in (False, True):
                f.fused_field = fused
                out[name, fused] = f(x, d)[0]
    _close(out["half", True], out["half", False], 1e-4, "half")
    _close(out["float", True], out["float", False], 1e-4, "float")
    assert float((out["half", True] - out["float", True]).abs().max()) > 1e-6


@pytest.mark.parametrize("kernel", ["w2", "w1"])
@pytest.mark.parametrize("cfg", ["f8_full", "f2_toy"])
def test_fp16_range_guard(cuda, cfg, kernel):
    """The three-product kernels split operands into two halves: above fp16's 65504 that would be inf / NaN.  The guard
    detects it on the device and the exact-fp32 kernel enqueued behind recomputes the call: finite, chain-equal output
    (i) with hidden activations driven past 65504, (ii) with a weight whose 2^8 multiple does not fit; and it does NOT
    fire for an ordinary model, nor for samples far outside the box."""
    f = _field(cuda, CONFIGS[cfg], seed=4)
    f.fused_field_precision, f.fused_field_kernel = "f16x3", kernel
    x, d = _inputs(cuda, 5000, seed=5)
    x[7] = torch.tensor([3.0e5, -2.0e6, 1.0e9])          # selector 0, raw coordinates beyond fp16

    def both():
        with torch.no_grad():
            f.fused_field = False
            rgb0, sig0 = f(x, d)
            den0 = f.query_density(x)
            f.fused_field = True
            rgb1, sig1 = f(x, d)
            fired_rgb = f._field_fused.range_guard_fired()
            den1 = f.query_density(x)
            fired_den = f._field_fused.range_guard_fired()
        return (rgb0, sig0, den0), (rgb1, sig1, den1), (fired_rgb, fired_den)

    ref, got, fired = both()
    assert fired == (False, False)
    assert all(bool(torch.isfinite(t).all()) for t in got)
    keep = torch.ones(x.shape[0], dtype=torch.bool, device=cuda)
    keep[7] = False                                       # its colour comes from clamped coordinates (density 0 either way)
    for a, b, what in zip(got, ref, ("rgb", "density (colour kernel)", "density")):
        _close(a[keep], b[keep], 2e-5, what)
    assert float(got[1][7]) == 0.0 and float(got[2][7]) == 0.0

    # (i) hidden activations of the base network far beyond 65504, pulled back by a small second layer
    with torch.no_grad():
        f.mlp_base.network[0].weight.mul_(300.0)
        f.mlp_base.network[0].bias.fill_(7.0e4)
        f.mlp_base.network[2].weight.mul_(1.0e-5)
    ref, got, fired = both()
    assert fired == (True, False)       # the density-only kernel keeps h1 in fp32 registers: nothing of it is split
    # (the chain's own density of the far-away sample is exp(huge) * 0 = NaN; the kernels select 0)
    assert all(bool(torch.isfinite(t).all()) for t in got) and all(bool(torch.isfinite(t[keep]).all()) for t in ref)
    for a, b, what in zip(got, ref, ("rgb", "density (colour kernel)", "density")):
        _close(a[keep], b[keep], 1e-4, what + ", activations beyond fp16")
    assert float(ref[2][keep].max()) > 0.01

    # (ii) a first-layer weight of 300: 2^8 * 300 > 65504 — flagged by the packer, every call takes the exact kernel
    g = _field(cuda, CONFIGS[cfg], seed=4)
    g.fused_field_precision, g.fused_field_kernel = "f16x3", kernel
    with torch.no_grad():
        g.mlp_base.network[0].weight[3, 5] = 300.0
        g.mlp_head[2].weight[1, 2] = -400.0
    f = g
    ref, got, fired = both()
    assert fired == (True, True)
    for a, b, what in zip(got, ref, ("rgb", "density (colour kernel)", "density")):
        _close(a[keep], b[keep], 1e-4, what + ", weight beyond fp16")
    # ... and once the weights are back in range the fp16 kernels serve the calls again
    with torch.no_grad():
        g.mlp_base.network[0].weight[3, 5] = 0.25
        g.mlp_head[2].weight[1, 2] = -0.25
    ref, got, fired = both()
    assert fired == (False, False)
    for a, b, what in zip(got, ref, ("rgb", "density (colour kernel)", "density")):
        _close(a[keep], b[keep], 2e-5, what)


@pytest.mark.parametrize("kernel", ["w2", "w1"])
def test_fused_field_is_repeatable(cuda, kernel):
    """The kernels have no atomics and a fixed summation order: the same call returns the same bits.  (A race between
    the two waves of a tile shows up here as a handful of rows of one tile differing in one call out of a few — that is
    how a build with the direction load moved in front of layer 2 was caught.)"""
    f = _field(cuda, CONFIGS["f8_full"], seed=3)
    f.fused_field_precision, f.fused_field_kernel = "f16x3", kernel
    x, d = _inputs(cuda, 70001, seed=70001)
    with torch.no_grad():
        rgb0, sig0 = f(x, d)
        den0 = f.query_density(x)
        for rep in range(60):
            rgb, sig = f(x, d)
            den = f.query_density(x)
            assert torch.equal(rgb, rgb0) and torch.equal(sig, sig0) and torch.equal(den, den0), rep


def test_fused_field_features_and_density_against_the_oracle_at_full_size(cuda, oracle):
    """The full-size anchor: not the repo's own chain but the ORACLE.  The two-wave kernel dumps the first layer's input
    rows as it computed them (`debug_features`); on the reference composition (12 x 3-D levels at T = 2^19, 3 planes x 4
    levels at T = 2^17, F = 8) and 2^16 points they must equal, bit for bit, oracle.grid_encode_forward on the binarised
    tables (gridencoder.cu:114-316) for the four encoders; the raw-coordinate / sinusoid columns are held to float64
    NumPy (the kernel's v_sin / v_cos: 3e-7), and the density to a float64 NumPy MLP on those very features (1e-4:
    north_star's bound)."""
    f = _field(cuda, CONFIGS["f8_full"], seed=21)
    f.fused_field_precision, f.fused_field_kernel = "f16x3", "w2"
    n = 1 << 16
    g = torch.Generator(device=cuda).manual_seed(5)
    x = torch.rand(n, 3, device=cuda, generator=g) * 3.0 - 1.5            # inside the box (up to rounding at the faces)
    x[:64] = torch.rand(64, 3, device=cuda, generator=g) * 3.4 - 1.7      # and a few outside
    mb = f.mlp_base
    k0 = mb.network[0].in_features
    feats = torch.full((n, 256), float("nan"), device=cuda)
    with torch.no_grad():
        f.fused_field = True
        f.query_density(x[:8])                                           # builds the evaluator
        den = f._field_fused(x, debug_features=feats)
    feats = feats.cpu().numpy()
    xu = ((x - f.aabb[:3]) / (f.aabb[3:] - f.aabb[:3])).cpu().numpy().astype(np.float32)
    col = 0
    for e, dims in zip(mb._encoders(), ((0, 1, 2), (0, 1), (0, 2), (1, 2))):
        table = e.params.detach().cpu().numpy()
        signs = np.where(table >= 0, 1.0, -1.0).astype(np.float32)      # STE_binary, ngp.py:24-39
        want = oracle.grid_encode_forward(np.ascontiguousarray(xu[:, dims]), signs, e.offsets_list.cpu().numpy(),
                                          e.resolutions_list.cpu().numpy(), threads=8)          # [L, N, F]
        want = np.transpose(want, (1, 0, 2)).reshape(n, -1)
        got = feats[:, col:col + want.shape[1]]
        assert np.array_equal(got, want), (dims, float(np.abs(got - want).max()))
        assert float(np.abs(want).max()) > 0.5
        col += want.shape[1]
    assert col == 192
    freqs = mb._freqs.cpu().numpy().astype(np.float64)
    x64 = xu.astype(np.float64)
    # raw coordinates, then per frequency sin (3) and cos (3); padding column 255
    assert np.array_equal(feats[:, col:col + 3], xu)
    for k, fr in enumerate(freqs):
        arg = (xu * np.float32(fr)).astype(np.float64)                    # the kernel multiplies in float32
        assert np.abs(feats[:, col + 3 + 6 * k: col + 6 + 6 * k] - np.sin(arg)).max() < 1e-6
        assert np.abs(feats[:, col + 6 + 6 * k: col + 9 + 6 * k] - np.cos(arg)).max() < 1e-6
    assert k0 == col + 3 + 6 * len(freqs) == 255 and np.all(feats[:, 255] == 0)
    # density from those features in float64
    w1, b1 = (t.detach().cpu().numpy().astype(np.float64) for t in (mb.network[0].weight, mb.network[0].bias))
    w2, b2 = (t.detach().cpu().numpy().astype(np.float64) for t in (mb.network[2].weight, mb.network[2].bias))
    h1 = np.maximum(feats[:, :255].astype(np.float64) @ w1.T + b1, 0.0)
    raw = h1 @ w2[0] + b2[0]
    sel = np.all((xu > 0) & (xu < 1), axis=1)
    want_den = np.where(sel, np.exp(raw - 1.0), 0.0)
    got_den = den.cpu().numpy().reshape(-1).astype(np.float64)
    assert np.abs(got_den - want_den).max() <= 1e-4 * want_den.max()
    assert np.array_equal(got_den == 0, ~sel) and sel[64:].all() and not sel[:64].all()


def test_fused_field_rgb_against_float64_at_full_size(cuda):
    """The colour half of the full-size anchor: rgb of the two-wave colour kernel (reference composition, F = 8, 2^16 points)
    against a float64 NumPy evaluation of both MLPs (closed-form SH) on the kernel's own first-layer input rows — which
    `test_fused_field_features_and_density_against_the_oracle_at_full_size` holds bit-equal to the oracle's encoder."""
    from test_gpu_field_chain import _float64_field
    f = _field(cuda, CONFIGS["f8_full"], seed=22, sh_fp16_round=False)        # (the half-rounded SH has its own golden)
    f.fused_field_precision, f.fused_field_kernel = "f16x3", "w2"
    n = 1 << 16
    x, d = _inputs(cuda, n, seed=123)
    feats = torch.full((n, 256), float("nan"), device=cuda)
    with torch.no_grad():
        f.fused_field = True
        f.query_density(x[:8])
        f._field_fused(x, debug_features=feats)
        rgb, den = f(x, d)
    xu = ((x - f.aabb[:3]) / (f.aabb[3:] - f.aabb[:3])).cpu().numpy().astype(np.float32)
    zeros = np.zeros((n, 3))
    rgb64, den64, _, _ = _float64_field(f, feats.cpu().numpy(), xu, d.cpu().numpy(), zeros, zeros[:, :1])
    assert np.abs(rgb.cpu().numpy() - rgb64).max() <= 1e-4
    assert np.abs(den.cpu().numpy()[:, 0] - den64).max() <= 1e-4 * den64.max()
    assert rgb64.std() > 0.01


@pytest.mark.parametrize("kernel", ["w2", "w1"])
@pytest.mark.parametrize("cfg", ["f8_full", "f2_toy"])
def test_fused_field_with_a_row_count_on_the_device(cuda, cfg, kernel):
    """cnc_fused_field_t.n_rows_dev: the call works on min(N, count) rows, the count read on the device — the rows in front of
    it get the values of an exactly sized call bit for bit, nothing behind it is touched (positions there may be garbage), a
    count beyond the capacity is the capacity, a count of zero does nothing.  Density-only and colour calls, both kernels,
    and the exact-fp32 form."""
    from cnc_amd import _lib
    f = _field(cuda, CONFIGS[cfg], seed=6)
    f.fused_field_kernel = kernel
    cap = 7013
    x, d = _inputs(cuda, cap, seed=8)
    for precision in ("f16x3", "f32"):
        f.fused_field_precision = precision
        with torch.no_grad():
            ff = f._fused_forward(x)
            assert ff is not None
            for count in (0, 1, 31, 32, 4097, cap, cap + 50):
                n = min(count, cap)
                xg = x.clone()
                xg[n:] = float("nan")                           # what lies behind the count is never read
                n_dev = torch.tensor([count], dtype=torch.int64, device=cuda)
                want_den = ff(x[:n]) if n else x.new_zeros((0, 1))
                want_den2, want_rgb = ff(x[:n], d[:n]) if n else (x.new_zeros((0, 1)), x.new_zeros((0, 3)))
                # outputs are allocated by the call: poison the allocator's next blocks through a first, discarded call
                got_den = ff(xg, n_rows_dev=n_dev)
                got_den2, got_rgb = ff(xg, d, n_rows_dev=n_dev)
                assert got_den.shape == (cap, 1) and got_rgb.shape == (cap, 3)
                assert torch.equal(got_den[:n], want_den) and torch.equal(got_den2[:n], want_den2)
                assert torch.equal(got_rgb[:n], want_rgb)
        # rows behind the count keep what the buffers held: call the C entry on buffers of our own
        import ctypes
        with torch.no_grad():
            st, keep, _ = ff._descriptor(x.device, False)
            den = torch.full((cap, 1), -7.0, device=cuda)
            n_dev = torch.tensor([100], dtype=torch.int64, device=cuda)
            st.n_rows_dev = n_dev.data_ptr()
            _lib.check(_lib.lib().cnc_field_fused_forward(ctypes.byref(st), x.data_ptr(), None, cap, den.data_ptr(), None,
                                                          _lib.stream(x.device)), "field_fused_forward")
            torch.cuda.synchronize()
            assert bool((den[100:] == -7.0).all()) and torch.equal(den[:100], ff(x[:100]))
