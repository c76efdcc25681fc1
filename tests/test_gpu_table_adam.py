"""cnc_table_adam (csrc/table_adam.hip, cnc_amd._table_adam): the tables' Adam update with the gradient summed from pieces,
against torch.optim.Adam — the optimizer the reference steps (examples/train_CNC_nerf_synthetic.py:254-259,363) — and
against the float64 formula; then inside the Trainer against the step that flushes the pieces into `.grad` first."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pieces(shapes, seed, dev):
    """Per table: four pieces — three whole-table ones and one that covers a row range (like the planes' graph's gradient
    of the 3-D table's finest level)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = []
    for rows, F in shapes:
        lo, hi = rows // 4, rows // 2 + 3
        out.append([(torch.randn(rows, F, generator=g).to(dev) * 3.0, None),
                    (torch.randn(rows, F, generator=g).to(dev) * 1e-3, None),
                    (torch.randn(hi - lo, F, generator=g).to(dev), (lo, hi)),
                    (torch.randn(rows, F, generator=g).to(dev) * 0.1, None)])
    return out


def _summed(pieces, p):
    tot = None
    for gt, rows in pieces:
        full = gt if rows is None else torch.zeros_like(p).index_copy_(0, torch.arange(rows[0], rows[1], device=p.device), gt)
        tot = full.clone() if tot is None else tot + full            # fp32, in the kernel's order
    return tot


@pytest.mark.parametrize("wd", [0.0, 2e-6], ids=["no_decay", "decay"])
def test_table_adam_against_the_library_optimizer_and_float64(cuda, wd):
    from cnc_amd._table_adam import TableAdam
    shapes = [(4099, 8), (515, 8), (1028, 8), (64, 4)]           # (rows, F): not multiples of the kernel's 4,096-element blocks
    torch.manual_seed(1)
    init = [(torch.rand(r, F, device=cuda) * 2 - 1) * 1e-2 for r, F in shapes]
    mine = [torch.nn.Parameter(t.clone()) for t in init]
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    other_a, other_b = torch.nn.Parameter(torch.ones(7, device=cuda)), torch.nn.Parameter(torch.ones(7, device=cuda))
    kw = dict(lr=6e-3, eps=1e-15, weight_decay=wd, fused=True)
    opt_a = torch.optim.Adam([{"params": [other_a]}, {"params": mine}], **kw)
    opt_b = torch.optim.Adam([{"params": [other_b]}, {"params": ref}], **kw)
    ta = TableAdam(opt_a, mine)
    p64 = [t.double() for t in init]
    m64 = [torch.zeros_like(t) for t in p64]
    v64 = [torch.zeros_like(t) for t in p64]
    b1, b2 = 0.9, 0.999
    for step in range(1, 7):
        opt_a.param_groups[1]["lr"] = opt_b.param_groups[1]["lr"] = lr = 6e-3 * (0.5 + 0.1 * step)   # a schedule moves it
        pcs = _pieces(shapes, 100 + step, cuda)
        # a table's own `.grad` (what autograd left) goes first and is dropped
        mine[0].grad = pcs[0][0][0].clone()
        ta.step({id(p): (pc[1:] if k == 0 else pc) for k, (p, pc) in enumerate(zip(mine, pcs))})
        other_a.grad = torch.full_like(other_a, 0.5)
        opt_a.step()                                          # skips the tables (`.grad` None), steps the rest
        for p, pc in zip(ref, pcs):
            p.grad = _summed(pc, p)
        other_b.grad = torch.full_like(other_b, 0.5)
        opt_b.step()
        for k, pc in enumerate(pcs):
            g = _summed(pc, ref[k]).double()
            g = g + wd * p64[k]
            m64[k] = m64[k] + (1 - b1) * (g - m64[k])
            v64[k] = b2 * v64[k] + (1 - b2) * g * g
            p64[k] = p64[k] - lr / (1 - b1 ** step) * m64[k] / (v64[k].sqrt() / (1 - b2 ** step) ** 0.5 + 1e-15)
        torch.cuda.synchronize()
        for k in range(len(shapes)):
            assert mine[k].grad is None
            sa, sb = opt_a.state[mine[k]], opt_b.state[ref[k]]
            assert float(sa["step"]) == float(sb["step"]) == step
            scale = float(ref[k].detach().abs().max())
            # the library's kernel and this one round differently at most in the last place of each of m, v, p
            assert float((mine[k] - ref[k]).detach().abs().max()) <= 4e-7 * scale, (step, k)
            assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 4e-7 * float(sb["exp_avg"].abs().max())
            assert float((sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max()) <= 4e-7 * float(sb["exp_avg_sq"].abs().max())
            # and neither drifts from the float64 recurrence (fp32 state: ~1e-7 per step)
            assert float((mine[k].detach().double() - p64[k]).abs().max()) <= 2e-6 * scale, (step, k)
        assert torch.equal(other_a, other_b)
    # the state is the library optimizer's own: a step through the library continues from it
    for p, q, pc in zip(mine, ref, _pieces(shapes, 999, cuda)):
        p.grad, q.grad = _summed(pc, p), _summed(pc, q)
    opt_a.step(); opt_b.step()
    ta.steps_done += 1
    torch.cuda.synchronize()
    for k in range(len(shapes)):
        assert float(opt_a.state[mine[k]]["step"]) == 7.0
        assert float((mine[k] - ref[k]).detach().abs().max()) <= 8e-7 * float(ref[k].detach().abs().max())


def test_table_adam_leaves_the_sign_planes_of_the_updated_tables(cuda):
    """With the encoders given, the kernel writes each updated table's sign bit plane and clip counter into the encoder's
    cache buffers: bit for bit what cnc_pack_sign_bits makes of the updated table, the count of entries outside [-1, 1],
    at the buffers' old addresses, and `_bit_plane` takes them as current."""
    from cnc_amd._table_adam import TableAdam
    from cnc_amd.backends import gridencoder_backend as be
    from cnc_amd.gridencoder import GridEncoder
    torch.manual_seed(2)
    encs = [GridEncoder(num_dim=3, n_features=8, resolutions_list=(10, 18, 33), log2_hashmap_size=12, ste_binary=True).to(cuda),
            GridEncoder(num_dim=2, n_features=8, resolutions_list=(34, 130), log2_hashmap_size=10, ste_binary=True).to(cuda)]
    with torch.no_grad():
        for e in encs:
            e.params.uniform_(-1.2, 1.2)                       # some beyond +-1, some that the update will carry across
            e.invalidate_caches()
    planes = [e._bit_plane(e.params) for e in encs]            # creates the cache buffers
    where = [(b.data_ptr(), c.data_ptr()) for b, c in planes]
    tables = [e.params for e in encs]
    other = torch.nn.Parameter(torch.ones(3, device=cuda))
    opt = torch.optim.Adam([{"params": [other]}, {"params": tables}], lr=0.3, eps=1e-15, fused=True)
    ta = TableAdam(opt, tables, encs)
    for step in range(3):
        ta.step({id(p): [(torch.randn_like(p), None)] for p in tables})
        other.grad = torch.ones_like(other)
        opt.step()                                             # its post-step hook drops every cache ...
        ta.mark_planes_current()                               # ... the tables' planes stand
        torch.cuda.synchronize()
        for e, (b_at, c_at) in zip(encs, where):
            bits, clip = e._bit_plane(e.params)                # taken from the cache: no repack
            assert (bits.data_ptr(), clip.data_ptr()) == (b_at, c_at)
            want_clip = torch.zeros(1, dtype=torch.int32, device=cuda)
            want = be.pack_sign_bits(e.params.detach(), None, want_clip)
            assert torch.equal(bits, want) and int(clip) == int(want_clip) == int((e.params.abs() > 1).sum())
            assert int(clip) > 0
    # without the follow-up call the hook's invalidation stands and the plane is repacked from the table (same contents)
    ta.step({id(p): [(torch.randn_like(p), None)] for p in tables})
    opt.step()
    assert all(e._bits_key is None for e in encs)
    for e in encs:
        bits, clip = e._bit_plane(e.params)
        assert torch.equal(bits, be.pack_sign_bits(e.params.detach())) and int(clip) == int((e.params.abs() > 1).sum())


def test_table_adam_refuses_what_it_cannot_do(cuda):
    from cnc_amd import _lib
    from cnc_amd._table_adam import TableAdam
    p = torch.nn.Parameter(torch.zeros(16, 8, device=cuda))
    q = torch.nn.Parameter(torch.zeros(3, device=cuda))
    with pytest.raises(ValueError):
        TableAdam(torch.optim.Adam([p, q], fused=True), [p])                 # the tables must be a group of their own
    with pytest.raises(ValueError):
        TableAdam(torch.optim.Adam([{"params": [q]}, {"params": [p]}], fused=True, amsgrad=True), [p])
    ta = TableAdam(torch.optim.Adam([{"params": [q]}, {"params": [p]}], fused=True), [p])
    with pytest.raises(RuntimeError):
        ta.step({id(p): [(torch.zeros(16, 8, device=cuda), None)] * 5})       # more than four pieces
    with pytest.raises(RuntimeError):
        ta.step({id(p): [(torch.zeros(4, 8, device=cuda), (0, 5))]})          # piece and range disagree
    a = _lib.AdamTables()
    assert _lib.lib().cnc_table_adam(None, 1e-3, 0.9, 0.999, 1e-15, 0.0, 1.0, None) != 0
    a.n_tables = 1
    assert _lib.lib().cnc_table_adam(ctypes.byref(a), 1e-3, 0.9, 0.999, 1e-15, 0.0, 1.0, None) != 0     # null table


def test_trainer_steps_the_tables_from_the_pieces(cuda, tmp_path):
    """A Trainer whose tables go through the kernel (the default) against one that flushes the pieces into `.grad` and lets
    the library step them: the same first steps (before the binarised tables' chaotic regime), finite after 40."""
    import os
    from cnc_amd.trainer import Trainer
    from test_gpu_trainer import _cfg
    if os.environ.get("CNC_TABLE_ADAM", "1") != "1":
        pytest.skip("the tables' Adam kernel is switched off (CNC_TABLE_ADAM=0)")

    def run(fused):
        tr = Trainer(_cfg(tmp_path, seed=3), device=cuda)
        g = torch.Generator(device=cuda)
        g.manual_seed(77)
        tr.context.rand_like = lambda t: torch.rand(t.shape, generator=g, device=t.device, dtype=t.dtype)
        tr.ctx_stream, tr.ctx_thread = None, False            # one stream, one thread: the order of float adds is fixed
        assert tr.table_adam is not None and tr.fused_table_adam
        tr.fused_table_adam = fused
        out = [tr.train_step(s) for s in range(40)]
        torch.cuda.synchronize()
        e = tr.field.mlp_base
        assert all(t.params.grad is None for t in e._encoders()) == fused
        assert int(tr.opt.state[e.encoding_xyz.params]["step"].item()) == 40 == tr.table_adam.steps_done
        return out, [t.params.detach().clone() for t in e._encoders()]

    ref, ref_t = run(False)
    got, got_t = run(True)
    for a, b in zip(ref[:4], got[:4]):
        assert a["n_rendering_samples"] == b["n_rendering_samples"] and a["num_rays"] == b["num_rays"]
        assert abs(a["mse"] - b["mse"]) <= 1e-5 * max(a["mse"], 1e-6) + 1e-9
        assert abs(a["bpp"] - b["bpp"]) <= 1e-5 * a["bpp"]
    assert abs(ref[-1]["bpp"] - got[-1]["bpp"]) <= 0.05 * ref[-1]["bpp"]
    assert abs(ref[-1]["mse"] - got[-1]["mse"]) <= 0.2 * ref[-1]["mse"]
    assert all(torch.isfinite(t).all() for t in got_t)
