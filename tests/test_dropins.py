"""The zero-edit route: after `cnc_amd.install_dropins()` every name the reference's own Python takes from
`_gridencoder`, `pack_and_align`, `nerfacc`, `torchac` and `tinycudann` resolves to this package and every recorded
call site binds to the stand-in's signature (tests/golden/dropin_manifest.json, produced in the build container by an
AST walk over the reference's examples — make_dropin_manifest.py).  The GPU part pushes data through the aliases."""
import importlib
import inspect
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
MANIFEST = json.load(open(os.path.join(HERE, "golden", "dropin_manifest.json")))
REF = "/root/reference"


@pytest.fixture(scope="module")
def dropins():
    import cnc_amd
    cnc_amd.install_dropins()
    return cnc_amd


def _resolve(module, name):
    obj = importlib.import_module(module)
    for part in name.split("."):
        obj = getattr(obj, part)
    return obj


def _binds(fn, shape, is_method):
    sig = inspect.signature(fn)
    args = [None] * (shape["npos"] + (1 if is_method else 0))
    sig.bind(*args, **{k: None for k in shape["kw"]})        # raises TypeError when the call would not fit


def test_every_name_resolves_and_every_call_binds(dropins):
    assert sorted(MANIFEST["roots"]) == ["_gridencoder", "nerfacc", "pack_and_align", "tinycudann", "torchac"]
    seen = set()
    for u in MANIFEST["uses"]:
        where = f"{u['file']}:{u['line']}"
        mod = importlib.import_module(u["module"])           # `import X` / `from X import ...` works
        assert mod.__name__.startswith("cnc_amd."), (u["module"], mod.__name__, where)
        if "name" not in u:
            continue
        if u["kind"] == "member":
            continue                                          # a buffer / attribute: instance-level, checked below
        obj = _resolve(u["module"], u["name"])
        seen.add((u["module"], u["name"]))
        if "call" in u and not u["call"]["star"]:
            target = obj.__init__ if inspect.isclass(obj) else obj
            try:
                _binds(target, u["call"], is_method=inspect.isclass(obj) or u["kind"] == "method")
            except TypeError as e:
                raise AssertionError(f"{u['module']}.{u['name']} does not accept the call at {where}: {u['call']} ({e})")
    # what the CNC path needs is all there
    for need in (("_gridencoder", "grid_encode_forward"), ("_gridencoder", "grid_encode_backward"),
                 ("_gridencoder", "cnt_np_embed"), ("_gridencoder", "cnt_np_embed_backward"),
                 ("pack_and_align", "align_and_pack_forward"), ("pack_and_align", "align_and_pack_backward"),
                 ("pack_and_align", "query_mask_3D"), ("pack_and_align", "query_mask_3D_qlist"),
                 ("torchac", "encode_float_cdf"), ("torchac", "decode_float_cdf"), ("tinycudann", "Encoding"),
                 ("nerfacc.volrend", "rendering"), ("nerfacc.grid", "traverse_grids"),
                 ("nerfacc.estimators.occ_grid", "OccGridEstimator.sampling")):
        assert need in seen, need


def test_aliases_are_the_same_module_objects(dropins):
    """`from nerfacc.estimators.occ_grid import OccGridEstimator` must give THE class of cnc_amd.nerfacc, not a second
    copy loaded through the alias's __path__ (relative imports inside such a copy would not even resolve)."""
    import cnc_amd.nerfacc as mine
    from nerfacc.estimators.occ_grid import OccGridEstimator
    from nerfacc.volrend import rendering
    assert OccGridEstimator is mine.OccGridEstimator and rendering is mine.rendering
    assert sys.modules["nerfacc.grid"] is sys.modules["cnc_amd.nerfacc.grid"]
    est = OccGridEstimator(roi_aabb=[-1.5] * 3 + [1.5] * 3, resolution=8, levels=1)
    for u in MANIFEST["uses"]:
        if u["kind"] == "member":
            assert hasattr(est, u["name"].split(".")[1]), u


def test_torchac_standin_round_trip_and_binary_stream(dropins):
    """encode_float_cdf / decode_float_cdf as utils_bpp_acc.py:77-110 calls them; the binary stream equals the
    product's own ±1 entry point (same coder underneath), general alphabets round-trip near their entropy."""
    import torchac

    from cnc_amd import _codec
    g = torch.Generator().manual_seed(5)
    p = (torch.rand(4000, 8, generator=g) * 0.98 + 0.01)
    x = torch.where(torch.rand(4000, 8, generator=g) < p, 1.0, -1.0)
    p_u = 1 - p.unsqueeze(-1)
    cdf = torch.cat([torch.zeros_like(p_u), p_u, torch.ones_like(p_u)], dim=-1)
    sym = ((x + 1) // 2).to(torch.int16)
    stream = torchac.encode_float_cdf(cdf, sym, check_input_bounds=True)
    assert isinstance(stream, bytes)
    back = torchac.decode_float_cdf(cdf, stream)
    assert back.dtype == torch.int16 and back.shape == sym.shape and torch.equal(back, sym)
    L = _codec.lib()
    n = x.numel()
    buf = np.empty(int(L.cnc_rc_bound(n)), np.uint8)
    nb = L.cnc_rc_encode_pm1(p.contiguous().data_ptr(), x.contiguous().data_ptr(), n, buf.ctypes.data, buf.shape[0])
    assert buf[:nb].tobytes() == stream
    bits = float(-(torch.where(x > 0, p, 1 - p)).log2().sum())
    assert len(stream) * 8 <= bits * 1.002 + 64
    with pytest.raises(ValueError):
        torchac.encode_float_cdf(cdf * 1.5, sym, check_input_bounds=True)
    # a 5-symbol alphabet
    logits = torch.randn(3000, 5, generator=g)
    pmf = torch.softmax(logits, -1)
    cdf5 = torch.cat([torch.zeros(3000, 1), pmf.cumsum(-1)], -1).clamp(max=1.0)
    cdf5[:, -1] = 1.0
    s5 = torch.multinomial(pmf, 1, generator=g).squeeze(-1).to(torch.int16)
    st5 = torchac.encode_float_cdf(cdf5, s5, check_input_bounds=True)
    assert torch.equal(torchac.decode_float_cdf(cdf5, st5), s5)
    h = float(-pmf.gather(1, s5.long()[:, None]).log2().sum())
    assert len(st5) * 8 <= h * 1.01 + 64
    with pytest.raises(ValueError):
        torchac.encode_float_cdf(cdf5, s5 + 5)
    # edges: nothing to code, a one-symbol alphabet, probabilities at and next to 0 / 1 (the +arange(Lp) of the 16-bit
    # conversion keeps every symbol codable)
    e = torchac.encode_float_cdf(torch.zeros(0, 3), torch.zeros(0, dtype=torch.int16))
    assert torchac.decode_float_cdf(torch.zeros(0, 3), e).shape == (0,)
    one = torch.tensor([[0.0, 1.0]] * 10)
    z = torch.zeros(10, dtype=torch.int16)
    assert torch.equal(torchac.decode_float_cdf(one, torchac.encode_float_cdf(one, z)), z)
    pe = torch.tensor([1e-9, 1 - 1e-9, 0.5, 0.0, 1.0])
    pu = (1 - pe).unsqueeze(-1)
    ce = torch.cat([torch.zeros_like(pu), pu, torch.ones_like(pu)], -1)
    se = torch.tensor([0, 1, 1, 0, 1], dtype=torch.int16)
    assert torch.equal(torchac.decode_float_cdf(ce, torchac.encode_float_cdf(ce, se, check_input_bounds=True)), se)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")
def test_reference_modules_import_against_the_dropins(dropins):
    """In the build container: the reference's own files import UNCHANGED on top of the stand-ins (CPU-redirected,
    because examples/utils.py builds CUDA constants at import), and its field / context classes construct."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden_context import cpu_redirect
    keep = {k: sys.modules.get(k) for k in ("utils", "utils_bpp_acc", "radiance_fields", "radiance_fields.ngp", "datasets",
                                            "datasets.utils")}
    saved_path = list(sys.path)
    undo = {n: getattr(torch, n) for n in ("tensor", "arange", "zeros", "ones", "rand", "empty", "randperm", "full", "randn")}
    undo_t = (torch.Tensor.cuda, torch.nn.Module.cuda, torch.cuda.synchronize, torch.cuda.empty_cache)
    try:
        for k in keep:
            sys.modules.pop(k, None)
        cpu_redirect()
        sys.path.insert(0, os.path.join(REF, "examples"))
        ngp = importlib.import_module("radiance_fields.ngp")
        ex = importlib.import_module("utils")
        ub = importlib.import_module("utils_bpp_acc")
        assert ngp._backend.__name__ == "cnc_amd.backends.gridencoder_backend"
        assert ub.pack_and_align.__name__ == "cnc_amd.backends.pack_and_align" and ub.torchac.__name__ == "cnc_amd.backends.torchac"
        assert ex.OccGridEstimator.__module__ == "cnc_amd.nerfacc.estimators.occ_grid"
        f = ngp.NGPRadianceField_mygrid_2D3D(aabb=[-1.5] * 3 + [1.5] * 3, n_features_per_level=2, n_neurons=16,
                                             resolutions_list=[6, 9], log2_hashmap_size=8, resolutions_list_2D=[10], log2_hashmap_size_2D=7)
        assert type(f.direction_encoding).__module__ == "cnc_amd.backends.tinycudann" and f.direction_encoding.n_output_dims == 16
        m = ub.CNC_context_models(num_dim=3, resolutions_list=[6, 9, 14, 20], resolutions_list_2D=[10, 18], log2_hashmap_size=8,
                                  log2_hashmap_size_2D=7, n_features=2, sample_num=100, max_context_layer_num=3, ste_binary=True,
                                  Pg_level=4, Pg_level_2D=2, Rb=8, step_update=16, skip_levels_3D=[0, 1], skip_levels_2D=[0])
        assert len(list(m.parameters())) > 0
    finally:
        for n, fn in undo.items():
            setattr(torch, n, fn)
        torch.Tensor.cuda, torch.nn.Module.cuda, torch.cuda.synchronize, torch.cuda.empty_cache = undo_t
        sys.path[:] = saved_path
        for k, v in keep.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


@pytest.mark.gpu
def test_data_flows_through_the_aliases(cuda, dropins):
    """One call of each stand-in through the reference's import names, on the GPU."""
    import _gridencoder
    import pack_and_align
    import tinycudann as tcnn
    import torchac
    from nerfacc.estimators.occ_grid import OccGridEstimator
    from conftest import make_grid
    offs, res, emb = make_grid([6, 9, 14], 10, 3, 4, seed=1)
    t = lambda a: torch.as_tensor(a, device=cuda)
    x = torch.rand(300, 3, device=cuda)
    out = torch.empty(3, 300, 4, device=cuda)
    _gridencoder.grid_encode_forward(x, t(emb), t(offs), t(res), out, 300, 3, 4, 3, 0, 128, 0.0, None, None, None)
    import oracle
    assert np.array_equal(out.cpu().numpy(), oracle.grid_encode_forward(x.cpu().numpy(), emb, offs, res))
    enc = tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "Composite", "nested": [
        {"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": 4}]})
    d = torch.nn.functional.normalize(torch.randn(64, 3, device=cuda), dim=-1)
    sh = enc((d + 1) / 2)
    assert sh.shape == (64, 16) and sh.dtype == torch.float16          # tcnn's default output precision
    assert abs(float(sh[0, 0]) - 0.2820948) < 1e-3
    pts = torch.randint(0, 16, (500, 3), device=cuda, dtype=torch.int16)
    vxl = torch.rand(8, 8, 8, device=cuda) < 0.5
    mask = torch.zeros(500, dtype=torch.int16, device=cuda)
    overlap = torch.zeros(500, dtype=torch.int32, device=cuda)
    pack_and_align.query_mask_3D(pts, vxl, mask, overlap, 18, 500)
    m_o, o_o = oracle.query_mask(pts.cpu().numpy(), vxl.cpu().numpy(), resolution=18)
    assert np.array_equal(mask.cpu().numpy(), m_o) and np.array_equal(overlap.cpu().numpy(), o_o)
    p = torch.rand(1000, device=cuda) * 0.9 + 0.05
    xs = torch.where(torch.rand(1000, device=cuda) < p, 1.0, -1.0)
    pu = (1 - p).cpu().unsqueeze(-1)
    cdf = torch.cat([torch.zeros_like(pu), pu, torch.ones_like(pu)], -1)
    sym = ((xs.cpu() + 1) // 2).to(torch.int16)
    assert torch.equal(torchac.decode_float_cdf(cdf, torchac.encode_float_cdf(cdf, sym, check_input_bounds=True)), sym)
    est = OccGridEstimator(roi_aabb=torch.tensor([-1.5] * 3 + [1.5] * 3), resolution=16, levels=1).to(cuda)
    est.binaries = torch.ones_like(est.binaries)
    o = torch.tensor([[0.0, 0.0, -4.0]], device=cuda).repeat(8, 1)
    dd = torch.tensor([[0.0, 0.0, 1.0]], device=cuda).repeat(8, 1)
    ri, ts, te = est.sampling(o, dd, render_step_size=0.1)
    assert ri.shape[0] > 0 and ts.shape == te.shape
