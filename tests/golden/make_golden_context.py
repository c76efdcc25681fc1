"""Golden vectors for the context-model pass, produced by the REFERENCE's own
`CNC_context_models` (examples/utils_bpp_acc.py) running in this container on CPU:

  * `device='cuda'` is redirected to CPU (the module builds CUDA tensors at import, :19-20);
  * its compiled dependencies are bound to the CPU oracle: `_gridencoder`, `pack_and_align`,
    and `torchac` (the oracle's restatement of the torchac coder);
  * a toy grid keeps the fixture small.  The dimension-wise branch is hard-wired to a finest 3-D
    resolution of 514 and Rb=128 through *default arguments* (:397-398,489,498,515); for the toy
    grid those defaults are re-bound on the instance (34 / 8) — the arithmetic is untouched.

    python tests/golden/make_golden_context.py     # writes tests/golden/context_toy.npz
"""
import ast
import functools
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import oracle  # noqa: E402
from make_golden import _OracleGridencoder, ref_get_grid_index  # noqa: E402

TOY = dict(res3=[6, 9, 14, 20, 26, 34], res2=[10, 18, 34, 66], T3=10, T2=9, F=4, Rb=8, fine=34,
           sample_num=400, max_pts=20000)


def cpu_redirect():
    """Map device='cuda' to CPU for the factory functions the reference module uses."""
    def wrap(fn):
        @functools.wraps(fn)
        def inner(*a, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return inner
    for name in ("tensor", "arange", "zeros", "ones", "rand", "empty", "randperm", "full", "randn"):
        setattr(torch, name, wrap(getattr(torch, name)))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None


class _GE(_OracleGridencoder):
    @staticmethod
    def cnt_np_embed(inputs, emb, outputs, N, resolution, F, hashmap_size, axis):
        outputs.add_(torch.from_numpy(oracle.cnt_np_embed(inputs.numpy(), emb.detach().numpy(), resolution, hashmap_size, axis)))

    @staticmethod
    def cnt_np_embed_backward(inputs, emb, out_sum, grad, grad_emb, N, resolution, F, hashmap_size, axis):
        grad_emb.add_(torch.from_numpy(oracle.cnt_np_embed_backward(inputs.numpy(), emb.detach().numpy(), out_sum.numpy(),
                                                                    grad.contiguous().numpy(), resolution, hashmap_size, axis)))


def stub_modules():
    sys.modules["_gridencoder"] = _GE()
    sys.modules["tinycudann"] = types.ModuleType("tinycudann")
    pa = types.ModuleType("pack_and_align")

    def fwd(feat, cnt, cumsum, N, M, F, V, dim):
        return torch.from_numpy(oracle.align_and_pack_forward(feat.detach().numpy(), cnt.numpy(), float(V)))

    def bwd(dL, feat, cnt, cumsum, N, M, F, T, dim):
        return torch.from_numpy(oracle.align_and_pack_backward(dL.numpy(), cnt.numpy(), int(T)))

    def q(points, vxl, mask, overlap, resolution, N):
        m, o = oracle.query_mask(points.numpy(), vxl.numpy(), resolution=int(resolution))
        mask.copy_(torch.from_numpy(m)); overlap.copy_(torch.from_numpy(o))

    def ql(points, vxl, mask, overlap, rl, N):
        m, o = oracle.query_mask(points.numpy(), vxl.numpy(), resolution_list=rl.numpy())
        mask.copy_(torch.from_numpy(m)); overlap.copy_(torch.from_numpy(o))

    pa.align_and_pack_forward, pa.align_and_pack_backward, pa.query_mask_3D, pa.query_mask_3D_qlist = fwd, bwd, q, ql
    sys.modules["pack_and_align"] = pa
    ta = types.ModuleType("torchac")

    def enc(cdf, sym, check_input_bounds=False):
        return oracle.rc_encode(cdf[..., 1].contiguous().numpy(), sym.numpy(), prob_is_cdf1=True)

    def dec(cdf, stream):
        return torch.from_numpy(oracle.rc_decode(cdf[..., 1].contiguous().numpy(), stream, prob_is_cdf1=True))

    ta.encode_float_cdf, ta.decode_float_cdf = enc, dec
    sys.modules["torchac"] = ta
    u = types.ModuleType("utils")
    u.get_grid_index = ref_get_grid_index()
    sys.modules["utils"] = u


def main():
    oracle.build()
    cpu_redirect()
    stub_modules()
    sys.path.insert(0, os.path.join(REF, "examples"))
    import radiance_fields.ngp as ngp
    import utils_bpp_acc as ub

    c = TOY
    out = {}
    torch.manual_seed(11)
    model = ub.CNC_context_models(num_dim=3, resolutions_list=c["res3"], resolutions_list_2D=c["res2"],
                                  log2_hashmap_size=c["T3"], log2_hashmap_size_2D=c["T2"], n_features=c["F"],
                                  sample_num=c["sample_num"], max_context_layer_num=3, ste_binary=True,
                                  Pg_level=6, Pg_level_2D=4, Rb=c["Rb"], step_update=16,
                                  skip_levels_3D=[0, 1, 2], skip_levels_2D=[0])
    # re-bind the defaults that hard-wire 514 / 128
    model.binary_vxl_len = c["Rb"]
    model.init_binary_vxl_coords(scale=c["fine"] - 2)
    orig_idx, orig_pn = model.get_idx_coords2, model.get_pn_embed_frac
    model.get_idx_coords2 = lambda bv, resolution=c["fine"]: orig_idx(bv, resolution)
    model.get_pn_embed_frac = lambda e, i, resolution=c["fine"], axis="xy": orig_pn(e, i, resolution, axis)
    model.MAX_POINTS_NUM_TO_OOM = c["max_pts"]
    for k, v in model.state_dict().items():
        out["sd_" + k] = v.numpy().copy()
    out["utils_rand"] = model.utils_rand.numpy().copy()
    for n in range(6):
        out[f"uv_{n}"] = model.unique_value_list[n].numpy()
        out[f"pos_{n}"] = model.pos_grid_sorted_list[n].numpy()
    out["unique_count_list"] = model.unique_count_list.numpy()
    out["sample_num_levels"] = model.sample_num_levels.numpy()

    encs = {}
    torch.manual_seed(12)
    for name, D, res, T in (("xyz", 3, c["res3"], c["T3"]), ("xy", 2, c["res2"], c["T2"]),
                            ("xz", 2, c["res2"], c["T2"]), ("yz", 2, c["res2"], c["T2"])):
        e = ngp.GridEncoder(num_dim=D, n_features=c["F"], resolutions_list=res, log2_hashmap_size=T, ste_binary=True)
        with torch.no_grad():
            e.params.uniform_(-1.3, 1.3)
            # bias the signs so the coder has something to gain
            e.params.add_(0.35 * torch.sin(torch.arange(e.params.shape[0]).float() / 37.0)[:, None])
        encs[name] = e
        out[f"params_{name}"] = e.params.detach().numpy().copy()

    Rb = c["Rb"]
    ax = (torch.arange(Rb).float() + 0.5) / Rb - 0.5
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
    binary = ((gx ** 2 + gy ** 2 + 1.4 * gz ** 2) < 0.33 ** 2)[None]
    binary[0, 1, 6, 2] = True
    out["binary_vxl"] = binary.numpy()

    # ---- training pass (two steps: tables refreshed at step 0, reused at step 1) ----
    for step, seed in ((0, 77), (1, 78)):
        torch.manual_seed(seed)
        for e in encs.values():
            e.zero_grad()
        model.zero_grad()
        bpp, mb = model.forward_binary_vxl_mixPg_3D2D(encs["xyz"], encs["xy"], encs["xz"], encs["yz"], binary, step=step)
        bpp.backward()
        out[f"fwd{step}_bpp"] = np.float64(bpp.item()); out[f"fwd{step}_mb"] = np.float64(mb)
        for name, e in encs.items():
            out[f"fwd{step}_grad_{name}"] = e.params.grad.numpy().copy()
        out[f"fwd{step}_grad_ctx3d_w0"] = model.context_model_3D[0].weight.grad.numpy().copy()
        out[f"fwd{step}_grad_ctx2d_w0"] = model.context_model_2D[0][0].weight.grad.numpy().copy()

    # ---- encode / decode ----
    with tempfile.TemporaryDirectory() as td:
        prefix = os.path.join(td, "b")
        with torch.no_grad():
            Pgs, est_mb, coded_mb = model.encode_binary_vxl_mixPg_3D2D(encs["xyz"], encs["xy"], encs["xz"], encs["yz"], binary, filename_prefix=prefix)
        files = sorted(f for f in os.listdir(td) if f.endswith(".b"))
        out["enc_files"] = np.array(files)
        out["enc_sizes"] = np.array([os.path.getsize(os.path.join(td, f)) for f in files], np.int64)
        out["enc_est_mb"] = np.float64(est_mb); out["enc_coded_mb"] = np.float64(coded_mb)
        out["pg_keys"] = np.array(list(Pgs.keys())); out["pg_vals"] = np.array([float(v) for v in Pgs.values()], np.float64)
        recs = [torch.ones_like(encs[n].params.data) for n in ("xyz", "xy", "xz", "yz")]
        with torch.no_grad():
            r = model.decode_binary_vxl_mixPg_3D2D(encs["xyz"], encs["xy"], encs["xz"], encs["yz"], *recs, binary, Pgs, filename_prefix=prefix)
        for name, t in zip(("xyz", "xy", "xz", "yz"), r):
            out[f"dec_{name}"] = t.numpy().astype(np.int8)
    np.savez_compressed(os.path.join(HERE, "context_toy.npz"), **out)
    print("files:", list(out["enc_files"]))
    print("sizes:", out["enc_sizes"].tolist(), "est MB", est_mb, "coded MB", coded_mb)
    print("bpp", out["fwd0_bpp"], out["fwd1_bpp"], os.path.getsize(os.path.join(HERE, "context_toy.npz")))
    # sanity: decoded == STE(params) on coded rows
    for name in ("xyz", "xy", "xz", "yz"):
        q = np.where(out[f"params_{name}"] >= 0, 1, -1)
        d = out[f"dec_{name}"]
        print(name, "rows equal to sign(params):", (d == q).all(axis=1).mean())


if __name__ == "__main__":
    main()
