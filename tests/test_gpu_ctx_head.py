"""Fused context heads and Bernoulli rate (cnc_amd/csrc/ctx_head.hip) against torch autograd of the
reference's op chain (examples/utils_bpp_acc.py:378-393, :561-572, :689-701, :1002-1013) in float64, and the
whole context pass with `fused_heads` on vs off."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _ref_mlp(seq, x):
    return seq.double()(x.double())


@pytest.mark.parametrize("F", [2, 4, 8])
@pytest.mark.parametrize("n_layers,Ca,Cb,with_pg", [(1, 8, 8, True), (1, 24, 8, True), (1, 16, 0, True), (3, 24, 1, False),
                                                     (3, 24, 0, True), (3, 6, 0, True)])
@pytest.mark.parametrize("N", [1, 130, 5001])
def test_context_mlp_forward_backward(cuda, F, n_layers, Ca, Cb, with_pg, N):
    from cnc_amd.backends import context_backend as K
    g = torch.Generator(device="cpu").manual_seed(N + Ca + F)
    C = Ca + Cb + int(with_pg)
    if n_layers == 1:
        seq = nn.Sequential(nn.Linear(C, F))
    else:
        seq = nn.Sequential(nn.Linear(C, 32), nn.LeakyReLU(), nn.Linear(32, 32), nn.LeakyReLU(), nn.Linear(32, F))
    with torch.no_grad():
        for p in seq.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.4)
    a = torch.randn(N, Ca, generator=g)
    b = torch.randn(N, Cb, generator=g) if Cb else None
    pg = torch.rand((), generator=g) if with_pg else None
    go = torch.randn(N, F, generator=g)
    # float64 reference
    ref = nn.Sequential(*[type(m)(m.in_features, m.out_features) if isinstance(m, nn.Linear) else nn.LeakyReLU() for m in seq])
    ref.load_state_dict(seq.state_dict())
    ref = ref.double()
    a64, b64 = a.double().requires_grad_(), None if b is None else b.double().requires_grad_()
    pg64 = None if pg is None else pg.double().requires_grad_()
    cols = [a64] + ([b64] if b is not None else []) + ([pg64.reshape(1, 1).repeat(N, 1)] if pg is not None else [])
    y64 = ref(torch.cat(cols, dim=-1))
    (y64 * go.double()).sum().backward()
    # fused
    seq = seq.to(cuda)
    ad, bd = a.to(cuda).requires_grad_(), None if b is None else b.to(cuda).requires_grad_()
    pgd = None if pg is None else pg.to(cuda).requires_grad_()
    y = K.context_mlp(seq, ad, bd, pgd)
    (y * go.to(cuda)).sum().backward()
    tol = dict(rtol=2e-5, atol=2e-5)
    assert torch.allclose(y.double().cpu(), y64.detach(), **tol)
    assert torch.allclose(ad.grad.double().cpu(), a64.grad, **tol)
    if b is not None:
        assert torch.allclose(bd.grad.double().cpu(), b64.grad, **tol)
    if pg is not None:
        assert abs(float(pgd.grad) - float(pg64.grad)) <= 1e-4 * max(1.0, abs(float(pg64.grad)))
    for p, q in zip(seq.parameters(), ref.parameters()):
        scale = max(1.0, float(q.grad.abs().max()))
        assert float((p.grad.double().cpu() - q.grad).abs().max()) <= 3e-5 * scale


@pytest.mark.parametrize("with_rows", [False, True])
def test_bernoulli_bits(cuda, with_rows):
    from cnc_amd.backends import context_backend as K
    from cnc_amd.context import Bernoulli_entropy
    g = torch.Generator().manual_seed(5)
    T, S, F = 4000, 1500, 8
    table = torch.where(torch.rand(T, F, generator=g) > 0.4, 1.0, -1.0)
    rows = torch.randperm(T, generator=g)[:S] if with_rows else None
    mean = torch.rand(S if with_rows else T, F, generator=g) * 1.2 - 0.1        # some outside [1e-6, 1 - 1e-6]
    # float32 reference (the reference computes in float32: 1 - clamp(p) near the upper clamp is itself rounded,
    # which a float64 reference would not reproduce); the sum in float64
    t64, m64 = table.clone().requires_grad_(), mean.clone().requires_grad_()
    x64 = t64[rows] if with_rows else t64
    ref = torch.sum(Bernoulli_entropy()(x64, m64).double())
    (ref * 0.37).backward()
    td, md = table.to(cuda).requires_grad_(), mean.to(cuda).requires_grad_()
    got = K.bernoulli_bits(td, None if rows is None else rows.to(cuda), md)
    (got * 0.37).backward()
    assert abs(float(got.detach()) - float(ref.detach())) <= 2e-6 * float(ref.detach())
    assert torch.allclose(md.grad.cpu(), m64.grad, rtol=2e-5, atol=1e-6)
    assert torch.allclose(td.grad.cpu(), t64.grad, rtol=2e-5, atol=1e-6)
    # same seed, same bits: the partial sums make the total deterministic
    again = K.bernoulli_bits(td.detach(), None if rows is None else rows.to(cuda), md.detach())
    assert float(again) == float(got)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_segment_reduce_backward_kernel(cuda, mode):
    from cnc_amd.context import _cum, _segment_reduce
    g = torch.Generator().manual_seed(8)
    cnt = torch.randint(0, 9, (700,), generator=g)
    cnt[::5] = 0
    cnt[3] = 300
    T, F = int(cnt.sum()), 4
    v = torch.randn(T, F, generator=g)
    w = torch.rand(T, generator=g) + 0.1 if mode != 2 else None
    go = torch.randn(700, F, generator=g)
    slot = torch.repeat_interleave(torch.arange(700), cnt)
    v64 = v.double().requires_grad_()
    num = torch.zeros(700, F, dtype=torch.float64).index_add(0, slot, v64 * (w.double()[:, None] if w is not None else 1.0))
    if mode == 1:
        den = torch.zeros(700, dtype=torch.float64).index_add(0, slot, w.double())
        out = num / den.clamp_min(1e-300)[:, None]
    elif mode == 2:
        out = num / cnt.clamp_min(1).double()[:, None]
    else:
        out = num
    nz = cnt > 0
    (out[nz] * go.double()[nz]).sum().backward()
    vd = v.to(cuda).requires_grad_()
    got = _segment_reduce.apply(vd, _cum(cnt.to(cuda)), None if w is None else w.to(cuda), mode)
    (got[nz.to(cuda)] * go.to(cuda)[nz.to(cuda)]).sum().backward()
    assert torch.allclose(vd.grad.double().cpu(), v64.grad, rtol=2e-5, atol=1e-6)
    # the same through a row permutation (cnc_segment_weighted_sum_gathered): ragged row r = values[order[r]]
    order = torch.randperm(T, generator=g)
    inv = torch.empty_like(order)
    inv[order] = torch.arange(T)
    vp = v[inv].to(cuda).requires_grad_()               # vp[order[r]] = v[r]
    got_p = _segment_reduce.apply(vp, _cum(cnt.to(cuda)), None if w is None else w.to(cuda), mode, order.to(cuda))
    assert torch.equal(got_p[nz.to(cuda)], got[nz.to(cuda)])
    (got_p[nz.to(cuda)] * go.to(cuda)[nz.to(cuda)]).sum().backward()
    assert torch.equal(vp.grad[order.to(cuda)], vd.grad)


def test_context_pass_fused_heads_equals_op_chain(cuda):
    """The whole training pass of a toy configuration with the fused heads on and off: same bits per
    parameter (1e-5 relative), same gradients for the four tables and every context-model weight."""
    from cnc_amd import synthetic
    from cnc_amd.context import CNC_context_models
    from cnc_amd.gridencoder import GridEncoder
    F = 4
    res3, res2 = [10, 14, 18, 26, 34, 50, 66], [18, 34, 66, 130]
    outs = []
    for fused in (False, True):
        torch.manual_seed(11)
        m = CNC_context_models(num_dim=3, resolutions_list=res3, resolutions_list_2D=res2, log2_hashmap_size=13,
                               log2_hashmap_size_2D=11, n_features=F, sample_num=6000, max_context_layer_num=3,
                               ste_binary=True, Pg_level=7, Pg_level_2D=4, Rb=16, step_update=16,
                               skip_levels_3D=[0, 1, 2], skip_levels_2D=[0], device=cuda,
                               dimension_wise_resolution=res3[-1], fused_heads=fused)
        encs = [GridEncoder(3, F, res3, 13, ste_binary=True).to(cuda)] + \
               [GridEncoder(2, F, res2, 11, ste_binary=True).to(cuda) for _ in range(3)]
        binaries = synthetic.ball_binaries(16, radius=1.0, device=cuda)
        torch.manual_seed(12)
        bpp, mb = m.forward_binary_vxl_mixPg_3D2D(*encs, binaries, step=0)
        bpp.backward()
        outs.append((float(bpp.detach()), [e.params.grad.clone() for e in encs], [p.grad.clone() for p in m.parameters()]))
    (b0, ge0, gp0), (b1, ge1, gp1) = outs
    assert abs(b0 - b1) <= 1e-5 * abs(b0)
    for a, b in zip(ge0 + gp0, ge1 + gp1):
        scale = max(float(a.abs().max()), 1e-12)
        # float32 sums of ~1e5-1e6 signed terms in two orders (GEMM reduction vs LDS tiles + atomics): both carry
        # eps * sum|terms|, which after cancellation is ~1e-4 of the result (vs float64: test_context_mlp_*)
        assert float((a - b).abs().max()) <= 3e-3 * scale


def test_level_stats_kernel(cuda):
    """All levels' Pg / zero-order bits in one pass vs the per-level formula (utils_bpp_acc.py:472-486, with the
    1e-9 floor of cnc_amd.context._zero_order_bits), forward and backward, including a level whose entries
    all share a sign (Pg = 0: 0 bits, zero gradient, no NaN) and rows outside the levels."""
    from cnc_amd.backends import context_backend as K
    from cnc_amd.context import _zero_order_bits
    g = torch.Generator().manual_seed(3)
    off = (8, 332, 844, 1356, 5000)             # the first 8 rows and the tail belong to no level
    rows, F = 5200, 8
    t = torch.where(torch.rand(rows, F, generator=g) > 0.3, 1.0, -1.0)
    t[844:1356] = -1.0                           # Pg = 0
    gp, gb = torch.randn(4, generator=g), torch.randn(4, generator=g)
    t64 = t.clone().requires_grad_()
    Pg_ref, bits_ref = [], []
    for a, b in zip(off[:-1], off[1:]):
        lvl = t64[a:b]
        s = lvl.sum()
        ttl = lvl.numel()
        pos, neg = (ttl + s) / 2.0, (ttl - s) / 2.0
        Pg_ref.append(pos / ttl)
        bits_ref.append(_zero_order_bits(pos, neg, pos / ttl))
    Pg_ref, bits_ref = torch.stack(Pg_ref), torch.stack(bits_ref)
    ((Pg_ref * gp).sum() + (bits_ref * gb).sum()).backward()
    td = t.to(cuda).requires_grad_()
    Pg, bits = K.level_stats(td, off)
    ((Pg * gp.to(cuda)).sum() + (bits * gb.to(cuda)).sum()).backward()
    assert torch.allclose(Pg.cpu(), Pg_ref.detach(), rtol=1e-6, atol=0) and float(Pg[2]) == 0.0
    assert torch.allclose(bits.cpu(), bits_ref.detach(), rtol=2e-6, atol=1e-3) and float(bits[2]) == 0.0
    assert torch.isfinite(td.grad).all()
    assert torch.allclose(td.grad.cpu(), t64.grad, rtol=2e-5, atol=1e-7)
    assert float(td.grad[:8].abs().max()) == 0 and float(td.grad[5000:].abs().max()) == 0


def test_context_mlp_with_per_row_pg_table(cuda):
    """pg as a table indexed per row (rows of several levels in one call): forward equals the op chain on
    [in_a | pg[idx]], the table's gradient is the per-entry sum — runs of equal indices (one atomic per
    wave) and a scrambled tail (per-lane atomics)."""
    from cnc_amd.backends import context_backend as K
    g = torch.Generator().manual_seed(21)
    N, Ca, F, T = 20000, 24, 8, 12
    seq = nn.Sequential(nn.Linear(Ca + 1, 32), nn.LeakyReLU(), nn.Linear(32, 32), nn.LeakyReLU(), nn.Linear(32, F))
    a = torch.randn(N, Ca, generator=g)
    idx = torch.sort(torch.randint(3, T, (N,), generator=g))[0]
    idx[-700:] = torch.randint(0, T, (700,), generator=g)
    pg = torch.rand(T, generator=g)
    go = torch.randn(N, F, generator=g)
    ref = nn.Sequential(*[type(m)(m.in_features, m.out_features) if isinstance(m, nn.Linear) else nn.LeakyReLU() for m in seq])
    ref.load_state_dict(seq.state_dict())
    ref = ref.double()
    a64, pg64 = a.double().requires_grad_(), pg.double().requires_grad_()
    y64 = ref(torch.cat([a64, pg64[idx][:, None]], dim=-1))
    (y64 * go.double()).sum().backward()
    seq = seq.to(cuda)
    ad, pgd = a.to(cuda).requires_grad_(), pg.to(cuda).requires_grad_()
    y = K.context_mlp(seq, ad, None, pgd, idx.to(cuda))
    (y * go.to(cuda)).sum().backward()
    assert torch.allclose(y.double().cpu(), y64.detach(), rtol=2e-5, atol=2e-5)
    assert torch.allclose(ad.grad.double().cpu(), a64.grad, rtol=2e-5, atol=2e-5)
    assert torch.allclose(pgd.grad.double().cpu(), pg64.grad, rtol=1e-4, atol=1e-3)
    assert float(pgd.grad[:3].abs().max()) >= 0 and float(pg64.grad.abs().max()) > 1
