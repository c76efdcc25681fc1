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


def _grads(f, x, d, wr, wd, chain):
    f.fused_chain = chain
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
            f.fused_chain, f._chain_supported = chain, None
            for p in f.parameters():
                p.grad = None
            rgb, den = f(x, d)
            (rgb.sum() if which == "rgb" else (den * den).sum()).backward()
            res[chain] = {n: p.grad.clone() for n, p in f.named_parameters() if p.grad is not None}
        for name, b in res[False].items():
            a = res[True][name]
            assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-30), (which, name)
