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
