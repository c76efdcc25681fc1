"""Regenerate tests/golden/*.npz from the reference's own Python (THIS container only).

    python tests/golden/make_golden.py

/root/reference does not exist on the GPU box, so the vectors produced here are committed as
data.  Nothing of the reference's source is stored: only seeded inputs and the outputs its
functions returned.  What is importable:
  * examples/utils.py:get_grid_index (extracted by AST: the module itself builds CUDA tensors at
    import time) -> index/hash pins;
  * nerfacc.grid._ray_aabb_intersect (pure torch) -> slab-test pins;
  * radiance_fields.ngp {GridEncoder, _grid_encode, STE_binary} with `_gridencoder` bound to the
    CPU oracle and tinycudann stubbed -> pins the host glue (offset tables, level slicing,
    [L,N,F]->[N,L*F] permute, STE forward/backward) of cnc_amd.gridencoder;
  * nerfacc.estimators.occ_grid.OccGridEstimator._update on CPU -> pins the occupancy EMA;
  * utils_bpp_acc.Bernoulli_entropy / STE under a CPU redirect -> entropy pins.
"""
import ast
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import oracle  # noqa: E402


def ref_get_grid_index():
    src = open(os.path.join(REF, "examples", "utils.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_grid_index"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module([fn], []), "ref_get_grid_index", "exec"), ns)
    return ns["get_grid_index"]


def gen_index():
    f = ref_get_grid_index()
    rng = np.random.default_rng(1234)
    cases = {}
    # (D, resolution, hashmap_size): dense, hashed, dense-but-padded (ceil8), 2-D planes
    for k, (D, R, hs) in enumerate([(3, 18, 5832), (3, 24, 13824), (3, 33, 35944), (3, 148, 2 ** 19),
                                    (3, 514, 2 ** 19), (3, 2049, 2 ** 19), (2, 130, 16904),
                                    (2, 514, 2 ** 17), (2, 1026, 2 ** 17), (3, 80, 2 ** 19), (3, 81, 2 ** 19)]):
        pos = rng.integers(0, R, size=(4096, D)).astype(np.int64)
        pos[:8] = np.array([[0] * D, [R - 1] * D, [1] * D, [R - 2] * D, [0] * D, [R - 1] * D, [1] * D, [R // 2] * D])
        out = f(hs, R, torch.from_numpy(pos)).numpy()
        cases[f"c{k}_D"] = np.int64(D); cases[f"c{k}_R"] = np.int64(R); cases[f"c{k}_hs"] = np.int64(hs)
        cases[f"c{k}_pos"] = pos.astype(np.int32); cases[f"c{k}_rows"] = out.astype(np.int64)
    cases["n_cases"] = np.int64(11)
    np.savez_compressed(os.path.join(HERE, "grid_index.npz"), **cases)


def gen_slab():
    sys.path.insert(0, REF)
    import nerfacc.grid as rg   # reference nerfacc (its _C stays unresolved; only torch code is used)
    rng = np.random.default_rng(77)
    n = 2000
    o = rng.normal(size=(n, 3)); o = o / np.linalg.norm(o, axis=1, keepdims=True) * rng.uniform(0.2, 5.0, size=(n, 1))
    d = rng.normal(size=(n, 3)); d = d / np.linalg.norm(d, axis=1, keepdims=True)
    o = o.astype(np.float32); d = d.astype(np.float32)
    aabbs = np.array([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], [-3, -3, -3, 3, 3, 3], [0.1, 0.2, -0.4, 0.6, 0.9, 0.3]], np.float32)
    out = {}
    for k, (near, far, miss) in enumerate([(-np.inf, np.inf, np.inf), (0.0, 1e10, 1e10), (0.5, 4.0, -1.0)]):
        t0, t1, h = rg._ray_aabb_intersect(torch.from_numpy(o), torch.from_numpy(d), torch.from_numpy(aabbs), near, far, miss)
        out[f"t0_{k}"] = t0.numpy(); out[f"t1_{k}"] = t1.numpy(); out[f"hit_{k}"] = h.numpy()
        out[f"nfm_{k}"] = np.array([near, far, miss], np.float32)
    np.savez_compressed(os.path.join(HERE, "ray_aabb.npz"), rays_o=o, rays_d=d, aabbs=aabbs, **out)
    sys.path.remove(REF)
    for m in [m for m in sys.modules if m == "nerfacc" or m.startswith("nerfacc.")]:
        del sys.modules[m]


class _OracleGridencoder(types.ModuleType):
    """`_gridencoder` bound to the CPU oracle, so the reference's autograd glue can run here."""

    def __init__(self):
        super().__init__("_gridencoder")

    @staticmethod
    def grid_encode_forward(inputs, embeddings, offsets, resolutions, outputs, N, D, F, L, max_level, Rb, PV, dy_dx, binary_vxl, min_level_id):
        out = oracle.grid_encode_forward(inputs.detach().numpy(), embeddings.detach().numpy(), offsets.numpy(), resolutions.numpy(),
                                         n_levels_calc=L, binary_vxl=None if binary_vxl is None else binary_vxl.numpy(),
                                         min_level_id=None if min_level_id is None else min_level_id.numpy())
        outputs.copy_(torch.from_numpy(out))

    @staticmethod
    def grid_encode_backward(grad, inputs, embeddings, offsets, resolutions, grad_embeddings, N, D, F, L, max_level, Rb, dy_dx, grad_inputs, binary_vxl, min_level_id):
        g = oracle.grid_encode_backward(grad.numpy(), inputs.detach().numpy(), embeddings.detach().numpy(), offsets.numpy(), resolutions.numpy(),
                                        binary_vxl=None if binary_vxl is None else binary_vxl.numpy(),
                                        min_level_id=None if min_level_id is None else min_level_id.numpy())
        grad_embeddings.copy_(torch.from_numpy(g))


def import_ref_ngp():
    sys.modules["_gridencoder"] = _OracleGridencoder()
    tcnn = types.ModuleType("tinycudann")
    sys.modules["tinycudann"] = tcnn
    sys.path.insert(0, os.path.join(REF, "examples"))
    import radiance_fields.ngp as ngp
    return ngp


def gen_gridencoder_glue():
    ngp = import_ref_ngp()
    out = {}
    torch.manual_seed(7)
    cfgs = [dict(num_dim=3, n_features=4, resolutions_list=(6, 9, 14, 20, 31, 44), log2_hashmap_size=10, ste_binary=True),
            dict(num_dim=2, n_features=8, resolutions_list=(10, 18, 34, 66), log2_hashmap_size=9, ste_binary=True),
            dict(num_dim=3, n_features=2, resolutions_list=(6, 9, 14), log2_hashmap_size=12, ste_binary=False)]
    for k, cfg in enumerate(cfgs):
        enc = ngp.GridEncoder(**cfg)
        with torch.no_grad():
            enc.params.uniform_(-1.5, 1.5)      # so the STE mask |x|<=1 is exercised
        D = cfg["num_dim"]
        x = torch.rand(257, D)
        y = enc(x)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        out[f"g{k}_offsets"] = enc.offsets_list.numpy(); out[f"g{k}_res"] = enc.resolutions_list.numpy()
        out[f"g{k}_params"] = enc.params.detach().numpy().copy(); out[f"g{k}_x"] = x.numpy(); out[f"g{k}_w"] = w.numpy()
        out[f"g{k}_y"] = y.detach().numpy(); out[f"g{k}_grad"] = enc.params.grad.numpy().copy()
        # level window + occupancy mask + outspace params (context-model style call)
        Rb = 8
        vxl = torch.rand([Rb] * D) < 0.6
        lo, hi = 1, len(cfg["resolutions_list"])
        osp = torch.sign(torch.randn_like(enc.params))
        y2 = enc(x, lo, hi, outspace_params=osp, binary_vxl=vxl)
        out[f"g{k}_vxl"] = vxl.numpy(); out[f"g{k}_osp"] = osp.numpy(); out[f"g{k}_y_win"] = y2.detach().numpy()
        if D == 3:
            mli = torch.randint(0, len(cfg["resolutions_list"]) - 2, (257,), dtype=torch.int32)
            y3 = enc.forward_diff_levels(x, mli, 2, binary_vxl=vxl)
            out[f"g{k}_mli"] = mli.numpy(); out[f"g{k}_y_diff"] = y3.detach().numpy()
        else:
            R = 12
            tab = torch.rand(R * R, cfg["n_features"])
            y4 = enc.forward_given_params(x, torch.tensor([0, R * R], dtype=torch.int32), torch.tensor([R], dtype=torch.int32), tab, vxl)
            out[f"g{k}_tab"] = tab.numpy(); out[f"g{k}_y_given"] = y4.detach().numpy()
    # STE_binary forward/backward known answers
    v = torch.tensor([-2.0, -1.0, -0.5, -0.0, 0.0, 0.3, 1.0, 1.0001, 5.0], requires_grad=True)
    s = ngp.STE_binary.apply(v)
    s.backward(torch.arange(1.0, 10.0))
    out["ste_in"] = v.detach().numpy(); out["ste_out"] = s.detach().numpy(); out["ste_grad"] = v.grad.numpy()
    te = ngp.trunc_exp(torch.tensor([-3.0, 0.0, 2.5]))
    out["trunc_exp"] = te.numpy()
    np.savez_compressed(os.path.join(HERE, "gridencoder_glue.npz"), **out)


def gen_occgrid():
    sys.path.insert(0, REF)
    from nerfacc.estimators.occ_grid import OccGridEstimator
    out = {}
    est = OccGridEstimator([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], resolution=16, levels=1)
    est.train()

    def occ_fn(x):
        return (torch.exp(-4.0 * (x ** 2).sum(-1, keepdim=True)) * 0.05)

    torch.manual_seed(123)
    for step in (0, 16, 256, 272):
        est._update(step=step, occ_eval_fn=occ_fn, occ_thre=0.01, ema_decay=0.95, warmup_steps=256)
        out[f"occs_{step}"] = est.occs.numpy().copy()
        out[f"bin_{step}"] = est.binaries.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "occ_grid.npz"), **out)
    sys.path.remove(REF)
    for m in [m for m in sys.modules if m == "nerfacc" or m.startswith("nerfacc.")]:
        del sys.modules[m]


def gen_entropy():
    """Bernoulli_entropy and the per-level statistics, evaluated by the reference code under a
    CPU redirect (its module creates CUDA tensors at import)."""
    src = open(os.path.join(REF, "examples", "utils_bpp_acc.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("Bernoulli_entropy",)]
    ns = {"torch": torch, "nn": torch.nn}
    exec(compile(ast.Module(keep, []), "ref_entropy", "exec"), ns)
    ent = ns["Bernoulli_entropy"]()
    torch.manual_seed(3)
    x = torch.sign(torch.randn(1000, 8)); x[x == 0] = 1
    p = torch.rand(1000, 8) * 1.2 - 0.1      # includes values outside (0,1): clamped
    bits = ent(x, p)
    np.savez_compressed(os.path.join(HERE, "entropy.npz"), x=x.numpy(), p=p.numpy(), bits=bits.numpy())


if __name__ == "__main__":
    oracle.build()
    gen_index()
    gen_slab()
    gen_occgrid()
    gen_entropy()
    gen_gridencoder_glue()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
