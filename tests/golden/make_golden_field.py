"""Golden vectors for the radiance field, the render harness and a short training trajectory, produced by
the REFERENCE's own Python running in this container on CPU (the build container only: /root/reference does
not exist on the GPU box).  Nothing of the reference's source is stored — only seeded inputs, key names and
the numbers its functions returned.

    python tests/golden/make_golden_field.py            # writes field_toy{,_sh16}.npz, render_toy.npz, train_toy.npz
    python tests/golden/make_golden_field.py field_sh16  # one of: field, field_sh16, render, train

What runs:
  * `radiance_fields.ngp.NGPRadianceField_mygrid_2D3D` (ngp.py:365-566: feature order of
    `compose_3D_2D_embed` :620-645, `Embedder` :569-617, geo_feat_dim, selector, trunc_exp(x - 1), the
    state-dict key names) with `_gridencoder` bound to the CPU oracle and `tinycudann.Encoding` bound to a
    closed-form degree-4 spherical-harmonics module (tiny-cuda-nn is a third-party dependency that is not in
    the reference tree; its SH convention is checked below against scipy's complex harmonics);
  * `examples/utils.py` `render_image_with_occgrid` (:83-216) and `render_image_with_occgrid_test` (:317-489)
    with the reference's own nerfacc Python on top of an `nerfacc.csrc` module bound to the CPU oracle
    (ray_aabb_intersect, traverse_grids, the segmented scans);
  * the body of the training loop (train_CNC_nerf_synthetic.py:302-366 — occupancy refresh, render, adaptive
    num_rays, mse + lmbda * bpp, both Adam groups with their chained schedulers, the never-unscaled 2^10 loss
    scale) restated around the reference's classes (field, estimator, `CNC_context_models`), on a procedural
    scene whose batches come from NumPy (so that the GPU test can regenerate them bit for bit).

Random draws: every `torch.rand_like` of the reference lands on the CPU generator; the GPU tests replay the
same stream (they patch `torch.rand_like` to draw on the CPU), so the two runs see the same jitter, the same
occupancy-cell offsets and the same context windows.

The parameters of the state dicts are not stored: `fill_state` writes seeded NumPy values into both the
reference's module here and the module under test there (key names and shapes ARE stored and loaded with
strict=True).
"""
import os
import sys
import types
import zlib

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import oracle  # noqa: E402
from make_golden_context import _GE, cpu_redirect, stub_modules  # noqa: E402

FIELD_CASES = {   # name -> constructor arguments of the field
    "f8": dict(n_features_per_level=8, n_neurons=160, resolutions_list=[6, 9, 14, 20, 26, 34], log2_hashmap_size=10,
               resolutions_list_2D=[10, 18, 34, 66], log2_hashmap_size_2D=9),
    "f2": dict(n_features_per_level=2, n_neurons=64, resolutions_list=[6, 9, 14, 20, 26, 34], log2_hashmap_size=10,
               resolutions_list_2D=[10, 18, 34, 66], log2_hashmap_size_2D=9),
}
AABB = [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]
TRAIN = dict(res3=[6, 9, 14, 20, 26, 34], res2=[10, 18, 34, 66], T3=10, T2=9, F=4, Rb=8, fine=34, sample_num=400,
             max_pts=20000, n_neurons=160, render_step_size=1e-2, init_batch_size=256, target=1 << 15, lmbda=2e-3,
             step_update=16, lr=6e-3, weight_decay=2e-6, milestones=[9000, 12000, 15000, 17000, 19000])


# ------------------------------------------------------------------------------------------------ shared helpers
def fill_state(sd, seed):
    """Seeded values for every parameter (keys ending in .weight / .bias / .params), by key name: tables (keys ending in
    `.params`) ~ U(-1.3, 1.3) so that the STE mask |x| <= 1 matters, everything else ~ U(-1, 1) / sqrt(fan_in).
    The same function lives in tests/test_gpu_field_golden.py."""
    out = {}
    for k in sorted(sd.keys()):
        v = sd[k]
        if not (torch.is_floating_point(v) and k.endswith((".weight", ".bias", ".params"))):
            out[k] = v.clone()
            continue
        rng = np.random.default_rng([seed, zlib.crc32(k.encode())])
        if k.endswith(".params"):
            a = rng.uniform(-1.3, 1.3, size=tuple(v.shape))
        else:
            fan_in = v.shape[-1] if v.dim() > 1 else v.shape[0]
            a = rng.uniform(-1.0, 1.0, size=tuple(v.shape)) / np.sqrt(fan_in)
        out[k] = torch.from_numpy(a.astype(np.float32))
    return out


def ball_batch(step, n, seed=5):
    """Training batch of the procedural scene (an opaque shaded ball of radius 0.8, random background) from a NumPy
    stream keyed by the step: rays_o, rays_d, pixels, bkgd as float32 arrays.  Same function in the GPU test."""
    rng = np.random.default_rng([seed, step])
    az = rng.uniform(0, 2 * np.pi, n)
    el = (rng.uniform(0, 1, n) - 0.3) * 1.2
    eye = 4.0 * np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1)
    target = rng.uniform(-0.9, 0.9, (n, 3))
    d = target - eye
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o, d = eye.astype(np.float32), d.astype(np.float32)
    o64, d64 = o.astype(np.float64), d.astype(np.float64)
    b = (o64 * d64).sum(-1)
    c = (o64 * o64).sum(-1) - 0.8 ** 2
    disc = b * b - c
    hit = disc > 0
    t = -b - np.sqrt(np.maximum(disc, 0))
    p = o64 + d64 * t[:, None]
    nrm = p / 0.8
    tex = 0.5 + 0.5 * np.sin(p * 9.0 + np.array([0.0, 2.0, 4.0]))
    lam = 0.35 + 0.65 * np.clip((nrm * np.array([0.3, 0.5, 0.8])).sum(-1), 0, 1)
    rgb = np.clip(tex * lam[:, None], 0, 1)
    bkgd = rng.uniform(0, 1, 3)
    pix = np.where(hit[:, None], rgb, bkgd[None])
    return o, d, pix.astype(np.float32), bkgd.astype(np.float32)


def sh4_f64(d):
    """Real spherical harmonics up to degree 3 (16 values; tiny-cuda-nn calls this 'degree 4') of unit
    directions, from scipy's complex harmonics with the Condon-Shortley phase: index l*l + l + m holds
    Re Y_l^0, sqrt(2) Re Y_l^m (m > 0), sqrt(2) Im Y_l^|m| (m < 0)."""
    from scipy.special import sph_harm_y
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    theta, phi = np.arccos(np.clip(z, -1, 1)), np.arctan2(y, x)
    out = np.empty((d.shape[0], 16))
    for l in range(4):
        for m in range(-l, l + 1):
            Y = sph_harm_y(l, abs(m), theta, phi)
            out[:, l * l + l + m] = Y.real if m == 0 else np.sqrt(2) * (Y.real if m > 0 else Y.imag)
    return out


def _sh4_closed_form(d):
    """tiny-cuda-nn's published polynomial form (spherical_harmonics.h, degree 4) in float64 — used only to
    assert that the scipy construction above is the same function, also off the unit sphere's poles."""
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


class _SHStub(torch.nn.Module):
    """`tcnn.Encoding(n_input_dims=3, {"otype": "Composite", "nested": [{SphericalHarmonics, degree 4}]})`:
    input in [0, 1]^3 (the field feeds (dir + 1) / 2, ngp.py:540-541), mapped back to [-1, 1]; the polynomial
    form evaluated in float64 and rounded once to float32 (the directions used below are unit vectors, where it
    equals the scipy construction — asserted in main())."""

    def __init__(self, n_input_dims, encoding_config, **kw):
        super().__init__()
        nested = encoding_config["nested"] if encoding_config.get("otype") == "Composite" else [encoding_config]
        assert len(nested) == 1 and nested[0]["otype"] == "SphericalHarmonics" and nested[0]["degree"] == 4
        self.n_input_dims, self.n_output_dims = n_input_dims, 16

    # tcnn's output precision: None = the float32 stand-in of rounds 1-3 (float64 polynomial rounded once);
    # torch.float16 = what tiny-cuda-nn does on a GPU with fp16 support and the reference's defaults — the published
    # polynomial evaluated in float32 and stored to a HALF tensor (ngp.py:412-425 asks for no dtype; :540-547 then
    # `cat`s it with float32 features, which promotes the rounded values back)
    output_dtype = None

    def forward(self, x):
        if _SHStub.output_dtype is torch.float16:
            d = x.detach().float().numpy() * np.float32(2.0) - np.float32(1.0)
            v = _sh4_closed_form(d)
            assert v.dtype == np.float32
            return torch.from_numpy(v.astype(np.float16))
        d = x.detach().double().numpy() * 2.0 - 1.0
        return torch.from_numpy(_sh4_closed_form(d).astype(np.float32))


class _Spec:
    """Attribute bag standing in for the C++ `RaySegmentsSpec` (nerfacc.cpp:100-129)."""

    def __init__(self):
        self.vals = self.is_left = self.is_right = self.is_valid = None
        self.chunk_starts = self.chunk_cnts = self.ray_indices = None


def _nerfacc_csrc():
    """`nerfacc.csrc` bound to the CPU oracle (only what the occupancy-grid path calls)."""
    m = types.ModuleType("nerfacc.csrc")
    t = torch.from_numpy
    m.RaySegmentsSpec = _Spec

    def ray_aabb_intersect(o, d, aabbs, near, far, miss):
        t0, t1, h = oracle.ray_aabb_intersect(o.numpy(), d.numpy(), aabbs.numpy(), near, far, miss)
        return t(t0), t(t1), t(h)

    def traverse_grids(o, d, mask, binaries, aabbs, t_sorted, t_indices, hits, near, far, step, cone,
                       want_iv, want_sm, want_term, limit, over_allocate):
        iv, sm, term = oracle.traverse_grids(o.numpy(), d.numpy(), binaries.numpy(), aabbs.numpy(), near.numpy(), far.numpy(),
                                             step, cone, None if limit <= 0 else limit, over_allocate, mask.numpy(),
                                             t_sorted.numpy(), t_indices.numpy(), hits.numpy())
        specs = []
        for rec in (iv, sm):
            s = _Spec()
            for k, v in rec.items():
                setattr(s, k, t(np.ascontiguousarray(v)))
            specs.append(s)
        return specs[0], specs[1], t(term)

    def scan(exclusive, prod):
        def f(starts, cnts, x, normalize=False, backward=False):
            if x.numel() == 0:
                return torch.empty_like(x)
            return t(oracle.segmented_scan(x.detach().numpy(), starts.numpy(), cnts.numpy(), exclusive, prod=prod,
                                           reverse=backward, normalize=normalize))
        return f

    def prod_bwd(exclusive):
        def f(starts, cnts, x, y, g):
            return t(oracle.prod_backward(x.detach().numpy(), y.detach().numpy(), g.numpy(), starts.numpy(), cnts.numpy(), exclusive))
        return f

    m.ray_aabb_intersect, m.traverse_grids = ray_aabb_intersect, traverse_grids
    m.inclusive_sum, m.exclusive_sum = scan(False, False), scan(True, False)
    m.inclusive_prod_forward = lambda s, c, x: scan(False, True)(s, c, x)
    m.exclusive_prod_forward = lambda s, c, x: scan(True, True)(s, c, x)
    m.inclusive_prod_backward, m.exclusive_prod_backward = prod_bwd(False), prod_bwd(True)
    return m


def import_reference():
    """The reference's examples + nerfacc importable on CPU with the oracle underneath."""
    oracle.build()
    cpu_redirect()
    stub_modules()                                  # _gridencoder / pack_and_align / torchac -> oracle
    del sys.modules["utils"]                        # the real examples/utils.py is wanted here
    tcnn = types.ModuleType("tinycudann")
    tcnn.Encoding = _SHStub
    sys.modules["tinycudann"] = tcnn
    sys.modules["nerfacc.csrc"] = _nerfacc_csrc()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "examples"))
    import nerfacc
    nerfacc.csrc = sys.modules["nerfacc.csrc"]
    # pack_info (nerfacc/pack.py:38) runs its torch ops only for tensors that say `is_cuda`: let CPU tensors say so
    # for the duration of that one call (its body is device-agnostic)
    import nerfacc.pack
    import nerfacc.volrend

    def claiming_cuda(fn):
        def wrapped(*a, **k):
            torch.Tensor.is_cuda = property(lambda self: True)
            try:
                return fn(*a, **k)
            finally:
                del torch.Tensor.is_cuda
        return wrapped
    wrapped = claiming_cuda(nerfacc.pack.pack_info)
    for mod in (nerfacc, nerfacc.pack, nerfacc.volrend):
        if hasattr(mod, "pack_info"):
            mod.pack_info = wrapped
    import radiance_fields.ngp as ngp
    import utils as ex_utils
    import utils_bpp_acc as ub
    from datasets.utils import Rays
    return ngp, ex_utils, ub, nerfacc, Rays


def occupancy(res, radius, seed):
    c = (np.arange(res, dtype=np.float32) + 0.5) / res * 3.0 - 1.5
    gx, gy, gz = np.meshgrid(c, c, c, indexing="ij")
    b = ((gx * gx + gy * gy + gz * gz) < radius * radius)[None]
    b ^= (np.random.default_rng(seed).uniform(size=b.shape) < 0.03)
    return b


class RandTape:
    """Records the shapes of the reference's `torch.rand_like` draws (the GPU test asserts that the module under
    test asks for the same shapes in the same order)."""

    def __init__(self):
        self.shapes = []
        self._orig = torch.rand_like

    def __enter__(self):
        def rl(t, *a, **k):
            self.shapes.append(tuple(t.shape))
            return self._orig(t, *a, **k)
        torch.rand_like = rl
        return self

    def __exit__(self, *a):
        torch.rand_like = self._orig


# ------------------------------------------------------------------------------------------------------- field
def gen_field(ngp, sh16=False):
    """sh16: the same cases with `tinycudann.Encoding` returning half precision (field_toy_sh16.npz)."""
    out = {}
    _SHStub.output_dtype = torch.float16 if sh16 else None
    for name, kw in FIELD_CASES.items():
        torch.manual_seed(3)
        f = ngp.NGPRadianceField_mygrid_2D3D(aabb=torch.tensor(AABB), ste_binary=True, ste_multistep=False, add_noise=False,
                                             Q=10, **kw)
        sd = f.state_dict()
        out[f"{name}_keys"] = np.array(list(sd.keys()))
        out[f"{name}_shapes"] = np.array([",".join(str(s) for s in v.shape) for v in sd.values()])
        for k, v in sd.items():                      # what fill_state keeps (buffers), by value
            if not (torch.is_floating_point(v) and k.endswith((".weight", ".bias", ".params"))):
                out[f"{name}_sd_{k}"] = v.numpy().copy()
        f.load_state_dict(fill_state(sd, seed=17), strict=True)
        out[f"{name}_geo_feat_dim"] = np.int64(f.geo_feat_dim)
        rng = np.random.default_rng(21)
        n = 256
        pos = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
        pos[:6] = np.array([[-1.5, 0, 0], [1.5, 0.2, 0.1], [0, 1.6, 0], [0.3, 0.3, -1.7], [0, 0, 0], [1.4999, -1.4999, 0.5]], np.float32)
        dirs = rng.normal(size=(n, 3))
        dirs = (dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)).astype(np.float32)
        w_rgb = rng.normal(size=(n, 3)).astype(np.float32)
        w_sig = rng.normal(size=(n, 1)).astype(np.float32)
        x, v = torch.from_numpy(pos), torch.from_numpy(dirs)
        density, feat = f.query_density(x, return_feat=True)
        rgb, sigma = f(x, v)
        loss = (rgb * torch.from_numpy(w_rgb)).sum() + (sigma * torch.from_numpy(w_sig)).sum()
        f.zero_grad()
        loss.backward()
        out.update({f"{name}_pos": pos, f"{name}_dirs": dirs, f"{name}_w_rgb": w_rgb, f"{name}_w_sig": w_sig,
                    f"{name}_density": density.detach().numpy(), f"{name}_feat": feat.detach().numpy()[:96],
                    f"{name}_rgb": rgb.detach().numpy(), f"{name}_sigma": sigma.detach().numpy(),
                    f"{name}_loss": np.float64(loss.item())})
        for k, p in f.named_parameters():
            out[f"{name}_grad_{k}"] = p.grad.numpy().copy()
        # the 255-wide (here: narrower) input of the base MLP, as the reference composes it
        with torch.no_grad():
            mb = f.mlp_base
            xu = (x - f.aabb[:3]) / (f.aabb[3:] - f.aabb[:3])
            xs, ys, zs = torch.chunk(xu, 3, dim=-1)
            feat_in = torch.cat([mb.encoding_xyz(xu), mb.encoding_xy(torch.cat([xs, ys], -1)), mb.encoding_xz(torch.cat([xs, zs], -1)),
                                 mb.encoding_yz(torch.cat([ys, zs], -1)), mb.embed_fn(xu)], -1)
        out[f"{name}_mlp_in"] = feat_in.numpy()[:64]
        print(name, "keys", len(sd), "mlp_in", tuple(feat_in.shape), "geo", f.geo_feat_dim, "loss", loss.item())
    _SHStub.output_dtype = None
    np.savez_compressed(os.path.join(HERE, "field_toy_sh16.npz" if sh16 else "field_toy.npz"), **out)


# ------------------------------------------------------------------------------------------------------ render
def make_field(ngp, kw, seed=17, density_bias=None):
    f = ngp.NGPRadianceField_mygrid_2D3D(aabb=torch.tensor(AABB), ste_binary=True, ste_multistep=False, add_noise=False, Q=10, **kw)
    sd = fill_state(f.state_dict(), seed)
    if density_bias is not None:
        sd["mlp_base.network.2.bias"][0] = density_bias
    f.load_state_dict(sd, strict=True)
    return f


def gen_render(ngp, ex_utils, nerfacc, Rays):
    from cnc_amd import synthetic
    out = {}
    kw = FIELD_CASES["f8"]
    f = make_field(ngp, kw, seed=17, density_bias=2.5)     # exp(2.5 - 1 + ...) ~ 4.5: surfaces form, early stop is exercised
    out["density_bias"] = np.float64(2.5)
    est = nerfacc.OccGridEstimator(roi_aabb=torch.tensor(AABB), resolution=32, levels=1)
    b = occupancy(32, 1.0, 9)
    est.binaries = torch.from_numpy(b)
    est.occs = torch.from_numpy(b.reshape(-1).astype(np.float32) * 0.02)
    out["binaries"] = b
    o, d = synthetic.pinhole_rays(16, 16, 0.6911, 4.0, 0.7, 0.5)
    out["rays_o"], out["rays_d"] = o.numpy(), d.numpy()
    bk = torch.tensor([0.2, 0.6, 1.0])
    out["bkgd"] = bk.numpy()
    step = 2e-2
    out["render_step_size"] = np.float64(step)

    # --- training render: stratified, whole batch, gradients
    f.train(); est.train()
    torch.manual_seed(101)
    with RandTape() as tape:
        rgb, acc, depth, n, extra = ex_utils.render_image_with_occgrid(f, est, Rays(o, d), near_plane=0.0, render_step_size=step,
                                                                       render_bkgd=bk, cone_angle=0.0, alpha_thre=0.0, return_extra=True)
    assert tape.shapes == [(256,)], tape.shapes
    torch.manual_seed(101)
    out["train_jitter"] = torch.rand(256).numpy()
    rng = np.random.default_rng(4)
    pix = rng.uniform(0, 1, (256, 3)).astype(np.float32)
    loss = F.mse_loss(rgb, torch.from_numpy(pix))
    f.zero_grad()
    loss.backward()
    out.update(train_pixels=pix, train_rgb=rgb.detach().numpy(), train_opacity=acc.detach().numpy(), train_depth=depth.detach().numpy(),
               train_n=np.int64(n), train_loss=np.float64(loss.item()),
               train_extra_sigmas=extra["sigmas"].detach().numpy(),
               train_grad_w0=f.mlp_base.network[0].weight.grad.numpy().copy(),
               train_grad_head_w2=f.mlp_head[4].weight.grad.numpy().copy(),
               train_grad_xyz=f.mlp_base.encoding_xyz.params.grad.numpy().copy(),
               train_grad_xz=f.mlp_base.encoding_xz.params.grad.numpy().copy())
    # the samples the estimator handed over (same seed -> same jitter)
    torch.manual_seed(101)
    with torch.no_grad():
        ri, ts, te = est.sampling(o, d, sigma_fn=lambda a, b_, r: f.query_density(o[r] + d[r] * (a + b_)[:, None] / 2.0).squeeze(-1),
                                  near_plane=0.0, render_step_size=step, stratified=True)
    assert ri.shape[0] == n
    out.update(train_ray_indices=ri.numpy().astype(np.int32), train_t_starts=ts.numpy(), train_t_ends=te.numpy())
    print("train render: samples", n, "opacity mean", float(acc.mean()), "loss", loss.item())

    # --- evaluation render in chunks (no jitter), and the iterative whole-image render
    f.eval(); est.eval()
    with torch.no_grad():
        rgb, acc, depth, n = ex_utils.render_image_with_occgrid(f, est, Rays(o.view(16, 16, 3), d.view(16, 16, 3)), near_plane=0.0,
                                                                render_step_size=step, render_bkgd=bk, test_chunk_size=96)
        out.update(eval_rgb=rgb.numpy(), eval_opacity=acc.numpy(), eval_depth=depth.numpy(), eval_n=np.int64(n))
        print("eval render: samples", n)
        for tag, thre in (("t0", 0.0), ("t1", 9e-2)):
            rgb, acc, depth, n = ex_utils.render_image_with_occgrid_test(1024, f, est, Rays(o.view(16, 16, 3), d.view(16, 16, 3)),
                                                                         near_plane=0.0, render_step_size=step, render_bkgd=bk,
                                                                         alpha_thre=thre)
            out.update({f"test_{tag}_rgb": rgb.numpy(), f"test_{tag}_opacity": acc.numpy(), f"test_{tag}_depth": depth.numpy(),
                        f"test_{tag}_n": np.int64(n), f"test_{tag}_alpha_thre": np.float64(thre)})
            print("iterative render", tag, "samples", n, "opacity mean", float(acc.mean()))
    np.savez_compressed(os.path.join(HERE, "render_toy.npz"), **out)


# -------------------------------------------------------------------------------------------------- trajectory
class SummationNoise:
    """Stand-in for the CUDA reference's own run-to-run spread: its encoder backward sums with float32 atomicAdd in
    whatever order the hardware schedules (gridencoder.cu:575-583), so two runs of the reference differ by about
    one ulp of each table entry's sum of |terms|.  While active, the oracle's gradient gets exactly that: g += z *
    2^-24 * (sum of |terms|), z ~ N(0, 1) from a NumPy stream (the torch generator is left alone).  The trajectory
    goldens store the unperturbed run and two perturbed ones; the spread between them is the band the GPU run
    is held to."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)

    def __enter__(self):
        mod = sys.modules["_gridencoder"]
        self._orig = mod.grid_encode_backward
        orig, rng = self._orig, self.rng

        def noisy(grad, inputs, embeddings, offsets, resolutions, grad_embeddings, N, D, F_, L, max_level, Rb, dy_dx, grad_inputs,
                  binary_vxl, min_level_id):
            orig(grad, inputs, embeddings, offsets, resolutions, grad_embeddings, N, D, F_, L, max_level, Rb, dy_dx, grad_inputs,
                 binary_vxl, min_level_id)
            mag = oracle.grid_encode_backward(np.abs(grad.numpy()), inputs.detach().numpy(), embeddings.detach().numpy(), offsets.numpy(),
                                              resolutions.numpy(), binary_vxl=None if binary_vxl is None else binary_vxl.numpy(),
                                              min_level_id=None if min_level_id is None else min_level_id.numpy())
            z = rng.standard_normal(mag.shape).astype(np.float32)
            grad_embeddings.add_(torch.from_numpy(z * mag * np.float32(2.0 ** -24)))
        mod.grid_encode_backward = noisy
        return self

    def __exit__(self, *a):
        sys.modules["_gridencoder"].grid_encode_backward = self._orig


class _NoNoise:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass


class _LossScale:
    """`torch.cuda.amp.GradScaler(2**10)` as the reference uses it on a GPU: `scale(loss)` multiplies by 2^10 and
    the optimisers are stepped directly, never unscaled (train_CNC_nerf_synthetic.py:211,361-364).  On this
    CPU-only container the real class disables itself, which would change what Adam's weight decay sees."""

    def scale(self, loss):
        return loss * (2.0 ** 10)


def gen_train(ngp, ex_utils, ub, nerfacc, Rays, tag, steps, warmup_iters):
    """The unperturbed trajectory (everything stored) + two with summation-order noise (series only)."""
    out = run_trajectory(ngp, ex_utils, ub, nerfacc, Rays, tag, steps, warmup_iters, noise_seed=None)
    for k in (1, 2):
        noisy = run_trajectory(ngp, ex_utils, ub, nerfacc, Rays, f"{tag}+noise{k}", steps, warmup_iters, noise_seed=100 + k)
        for name in ("mse", "bpp", "mb", "n_samples", "num_rays", "occupied", "final_sign_xyz", "final_w0_norm", "signs", "norms"):
            out[f"noise{k}_{name}"] = noisy[name]
    np.savez_compressed(os.path.join(HERE, f"train_toy_{tag}.npz"), steps=np.int64(steps), warmup_iters=np.int64(warmup_iters), **out)


def run_trajectory(ngp, ex_utils, ub, nerfacc, Rays, tag, steps, warmup_iters, noise_seed):
    c = TRAIN
    out = {}
    torch.manual_seed(11)
    ctx = ub.CNC_context_models(num_dim=3, resolutions_list=c["res3"], resolutions_list_2D=c["res2"], log2_hashmap_size=c["T3"],
                                log2_hashmap_size_2D=c["T2"], n_features=c["F"], sample_num=c["sample_num"],
                                max_context_layer_num=3, ste_binary=True, Pg_level=6, Pg_level_2D=4, Rb=c["Rb"],
                                step_update=c["step_update"], skip_levels_3D=[0, 1, 2], skip_levels_2D=[0])
    ctx.binary_vxl_len = c["Rb"]                       # the defaults that hard-wire 514 / 128 (as make_golden_context.py)
    ctx.init_binary_vxl_coords(scale=c["fine"] - 2)
    orig_idx, orig_pn = ctx.get_idx_coords2, ctx.get_pn_embed_frac
    ctx.get_idx_coords2 = lambda bv, resolution=c["fine"]: orig_idx(bv, resolution)
    ctx.get_pn_embed_frac = lambda e, i, resolution=c["fine"], axis="xy": orig_pn(e, i, resolution, axis)
    ctx.MAX_POINTS_NUM_TO_OOM = c["max_pts"]
    for k, v in ctx.state_dict().items():
        out["ctx_sd_" + k] = v.numpy().copy()
    est = nerfacc.OccGridEstimator(roi_aabb=torch.tensor(AABB), resolution=c["Rb"], levels=1)
    kw = dict(n_features_per_level=c["F"], n_neurons=c["n_neurons"], resolutions_list=c["res3"], log2_hashmap_size=c["T3"],
              resolutions_list_2D=c["res2"], log2_hashmap_size_2D=c["T2"])
    f = ngp.NGPRadianceField_mygrid_2D3D(aabb=est.aabbs[-1], ste_binary=True, ste_multistep=False, add_noise=False, Q=10, **kw)
    sd = f.state_dict()
    out["field_keys"] = np.array(list(sd.keys()))
    filled = fill_state(sd, seed=23)
    for k in filled:                                   # tables start small, as the reference initialises them (ngp.py:221-223)
        if k.endswith(".params"):
            filled[k] = filled[k] * (1e-4 / 1.3)
    f.load_state_dict(filled, strict=True)
    if noise_seed is not None:
        # a perturbation of a few ulps (2^-22 relative) of every dense weight: from there on every activation, every
        # gradient and every Adam ratio differs from the unperturbed run in its last bits, the way two correct
        # float32 implementations with different summation orders differ
        rng = np.random.default_rng(noise_seed + 7)
        with torch.no_grad():
            for name, p_ in list(f.named_parameters()) + list(ctx.named_parameters()):
                if not name.endswith(".params"):
                    p_.mul_(torch.from_numpy((1.0 + 2.0 ** -22 * rng.standard_normal(tuple(p_.shape))).astype(np.float32)))

    opt = torch.optim.Adam([{"params": f.parameters()}], lr=c["lr"], eps=1e-15, weight_decay=c["weight_decay"])
    opt2 = torch.optim.Adam([{"params": ctx.parameters()}], lr=c["lr"], eps=1e-15)

    def sched(o):
        return torch.optim.lr_scheduler.ChainedScheduler([
            torch.optim.lr_scheduler.LinearLR(o, start_factor=0.01, total_iters=warmup_iters),
            torch.optim.lr_scheduler.MultiStepLR(o, milestones=c["milestones"], gamma=0.33)])
    s1, s2 = sched(opt), sched(opt2)
    scaler = _LossScale()
    num_rays = c["init_batch_size"]
    rec = {k: [] for k in ("mse", "bpp", "mb", "n_samples", "num_rays", "occupied", "lr")}
    signs, norms = [], []
    torch.manual_seed(29)
    with RandTape() as tape, (SummationNoise(noise_seed) if noise_seed is not None else _NoNoise()):
        for step in range(steps):
            f.train(); est.train()
            o, d, pix, bk = (torch.from_numpy(a) for a in ball_batch(step, num_rays))
            est.update_every_n_steps(step=step, occ_eval_fn=lambda x: f.query_density(x) * c["render_step_size"], occ_thre=1e-2,
                                     n=c["step_update"])
            rgb, acc, depth, n, extra = ex_utils.render_image_with_occgrid(f, est, Rays(o, d), near_plane=0.0,
                                                                           render_step_size=c["render_step_size"], render_bkgd=bk,
                                                                           cone_angle=0.0, alpha_thre=0.0, return_extra=True)
            assert n > 0
            rec["num_rays"].append(num_rays)
            num_rays = int(num_rays * (c["target"] / float(n)))
            mse = F.mse_loss(rgb, pix)
            e = f.mlp_base
            bpp, mb = ctx.forward_binary_vxl_mixPg_3D2D(e.encoding_xyz, e.encoding_xy, e.encoding_xz, e.encoding_yz, est.binaries,
                                                        verbose=False, sample_num=None, step=step)
            loss = mse + c["lmbda"] * bpp
            opt.zero_grad(); opt2.zero_grad()
            scaler.scale(loss).backward()
            rec["lr"].append(opt.param_groups[0]["lr"])
            opt.step(); opt2.step(); s1.step(); s2.step()
            rec["mse"].append(mse.item()); rec["bpp"].append(bpp.item()); rec["mb"].append(float(mb)); rec["n_samples"].append(n)
            rec["occupied"].append(int(est.binaries.sum()))
            # state after the step: the signs of every table entry (what the binarised model IS) and norms of the dense parts
            e = f.mlp_base
            signs.append(np.packbits(np.concatenate([(t.params.detach().numpy() >= 0).reshape(-1) for t in
                                                     (e.encoding_xyz, e.encoding_xy, e.encoding_xz, e.encoding_yz)])))
            norms.append([float(p.detach().double().norm()) for p in list(f.parameters())[4:]] +
                         [float(p.detach().double().norm()) for p in ctx.parameters()])
            if step == 0:
                out["step0_rgb"] = rgb.detach().numpy(); out["step0_opacity"] = acc.detach().numpy()
                out["step0_binaries"] = est.binaries.numpy().copy()
    out["rand_like_shapes"] = np.array([",".join(str(s) for s in sh) for sh in tape.shapes])
    for k, v in rec.items():
        out[k] = np.asarray(v, np.float64)
    out["signs"] = np.stack(signs)
    out["norms"] = np.asarray(norms)
    out["final_sign_xyz"] = (f.mlp_base.encoding_xyz.params.detach().numpy() >= 0)
    out["final_w0_norm"] = np.float64(f.mlp_base.network[0].weight.detach().norm().item())
    out["final_ctx3d_w0"] = ctx.context_model_3D[0].weight.detach().numpy().copy()
    psnr = -10 * np.log10(np.asarray(rec["mse"]))
    print(tag, "psnr", np.round(psnr[[0, 1, 2, steps // 2, steps - 1]], 3), "bpp", np.round(rec["bpp"], 4)[[0, steps - 1]],
          "samples", rec["n_samples"][:4], rec["n_samples"][-1], "rays", rec["num_rays"][:4], "occ", rec["occupied"][0], rec["occupied"][-1])
    return out


def main():
    u = np.random.default_rng(0).normal(size=(1000, 3))
    u /= np.linalg.norm(u, axis=-1, keepdims=True)
    assert np.abs(sh4_f64(u) - _sh4_closed_form(u)).max() < 1e-13, "SH convention"
    ngp, ex_utils, ub, nerfacc, Rays = import_reference()
    which = sys.argv[1:] or ["field", "render", "train"]
    if "field" in which:
        gen_field(ngp)
    if "field_sh16" in which or "field" in which:
        gen_field(ngp, sh16=True)
    if "render" in which:
        gen_render(ngp, ex_utils, nerfacc, Rays)
    if "train" in which:
        gen_train(ngp, ex_utils, ub, nerfacc, Rays, "ref", steps=20, warmup_iters=1000)     # the reference's schedule
        gen_train(ngp, ex_utils, ub, nerfacc, Rays, "fast", steps=40, warmup_iters=10)      # same loop, the warm-up compressed
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
