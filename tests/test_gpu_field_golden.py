"""The radiance field, the render harness and the training step against goldens produced by the REFERENCE's own
Python (tests/golden/make_golden_field.py: `NGPRadianceField_mygrid_2D3D` ngp.py:365-566,
`render_image_with_occgrid{,_test}` examples/utils.py:83-216,317-489, the loop body of
train_CNC_nerf_synthetic.py:302-366 — all run on CPU in the build container with the oracle underneath).

Tolerances: north_star's 1e-4 for rendered RGB / densities (relative to the tensor's scale), sample counts and
sample positions exact; gradients 2e-4 of their largest entry (float32 sums of ~10^3-10^4 terms in another
order); the training trajectory within the band stated in `test_training_trajectory`."""
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
AABB = [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]
FIELD_CASES = {
    "f8": dict(n_features_per_level=8, n_neurons=160, resolutions_list=[6, 9, 14, 20, 26, 34], log2_hashmap_size=10,
               resolutions_list_2D=[10, 18, 34, 66], log2_hashmap_size_2D=9),
    "f2": dict(n_features_per_level=2, n_neurons=64, resolutions_list=[6, 9, 14, 20, 26, 34], log2_hashmap_size=10,
               resolutions_list_2D=[10, 18, 34, 66], log2_hashmap_size_2D=9),
}
TRAIN = dict(res3=[6, 9, 14, 20, 26, 34], res2=[10, 18, 34, 66], T3=10, T2=9, F=4, Rb=8, fine=34, sample_num=400,
             max_pts=20000, n_neurons=160, render_step_size=1e-2, init_batch_size=256, target=1 << 15, lmbda=2e-3,
             step_update=16, lr=6e-3, weight_decay=2e-6, milestones=[9000, 12000, 15000, 17000, 19000])


def fill_state(sd, seed):
    """Same seeded values as make_golden_field.fill_state (parameter values are not stored in the fixture)."""
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
    """Same NumPy batch as make_golden_field.ball_batch."""
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


def close(got, want, tol=1e-4, what=""):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert err <= tol * scale, (what, err, scale)


class cpu_rand_like:
    """Replay the reference's CPU random stream: `torch.rand_like(t)` draws `t.shape` floats from the CPU
    generator (what the reference did on its CPU tensors) and moves them to t's device.  Records the shapes."""

    def __init__(self):
        self.shapes = []

    def __enter__(self):
        self._orig = torch.rand_like

        def rl(t, *a, dtype=None, **k):
            self.shapes.append(tuple(t.shape))
            return torch.rand(t.shape, dtype=dtype or (t.dtype if t.is_floating_point() else torch.float32)).to(t.device)
        torch.rand_like = rl
        return self

    def __exit__(self, *a):
        torch.rand_like = self._orig


def build_field(cuda, kw, fused, seed=17, density_bias=None, table_scale=None):
    from cnc_amd.field import NGPRadianceField_mygrid_2D3D
    # sh_fp16_round=False: the render / training goldens were made with the float32 stand-in for tiny-cuda-nn's encoding
    f = NGPRadianceField_mygrid_2D3D(aabb=torch.tensor(AABB), ste_binary=True, ste_multistep=False, add_noise=False, Q=10,
                                     fused_features=fused, sh_fp16_round=False, **kw)
    sd = fill_state(f.state_dict(), seed)
    if density_bias is not None:
        sd["mlp_base.network.2.bias"][0] = density_bias
    if table_scale is not None:
        for k in sd:
            if k.endswith(".params"):
                sd[k] = sd[k] * table_scale
    f.load_state_dict(sd, strict=True)
    return f.to(cuda)


# ------------------------------------------------------------------------------------------------------- field
@pytest.mark.parametrize("bucket_min", [1 << 30, 0], ids=["exact_rows", "padded_rows"])
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("sh", ["sh_half", "sh_float"])
@pytest.mark.parametrize("case", ["f8", "f2"])
def test_field_matches_reference(cuda, case, fused, bucket_min, sh):
    """`padded_rows`: every call runs at a bucketed row count (field._bucket_rows: what the training step does
    from 4096 samples on) — the reference's values must come out of the first N rows all the same.
    `sh_half` (the product's default): the reference class with `tinycudann.Encoding` returning a HALF tensor, as
    tiny-cuda-nn does on the reference's CUDA path (field_toy_sh16.npz); `sh_float`: the float32 stand-in of rounds
    1-3 (field_toy.npz).  The weight gradient of the head's first layer tells the two apart (2.5 %)."""
    from cnc_amd.field import NGPRadianceField_mygrid_2D3D
    if bucket_min == 0 and not fused:
        pytest.skip("rows are only bucketed on the fused path")
    g = np.load(os.path.join(GOLD, "field_toy_sh16.npz" if sh == "sh_half" else "field_toy.npz"))
    kw = FIELD_CASES[case]
    f = NGPRadianceField_mygrid_2D3D(aabb=torch.tensor(AABB), ste_binary=True, ste_multistep=False, add_noise=False, Q=10,
                                     fused_features=fused, **({} if sh == "sh_half" else {"sh_fp16_round": False}), **kw)
    assert f.sh_fp16_round == (sh == "sh_half") and f.direction_encoding.fp16_round == f.sh_fp16_round
    assert f.fused_glue == fused            # the half rounding no longer turns the fused glue kernels off
    sd = f.state_dict()
    # the reference's state dict: same keys in the same order, same shapes, same small buffers
    assert list(sd.keys()) == [str(k) for k in g[f"{case}_keys"]]
    assert [",".join(str(s) for s in v.shape) for v in sd.values()] == [str(s) for s in g[f"{case}_shapes"]]
    for k, v in sd.items():
        if not (torch.is_floating_point(v) and k.endswith((".weight", ".bias", ".params"))):
            assert np.array_equal(v.numpy(), g[f"{case}_sd_{k}"]), k
            assert str(v.numpy().dtype) == str(g[f"{case}_sd_{k}"].dtype), k
    assert f.geo_feat_dim == int(g[f"{case}_geo_feat_dim"])
    f.load_state_dict(fill_state(sd, seed=17), strict=True)
    f = f.to(cuda)
    f.row_bucket_min = bucket_min
    x = torch.from_numpy(g[f"{case}_pos"]).to(cuda)
    if bucket_min == 0:
        assert f._bucket_rows(1000) == 1024 and f._bucket_rows(260000) == 262144 and f._bucket_rows(64) == 64
        f._bucket_rows = lambda n: (n + 64) // 64 * 64          # the fixture has 256 rows: always pad
    v = torch.from_numpy(g[f"{case}_dirs"]).to(cuda)
    # the base MLP's input as the reference composes it: [xyz levels | xy | xz | yz | x, sin, cos ...]
    with torch.no_grad():
        xu = (x - f.aabb[:3]) / (f.aabb[3:] - f.aabb[:3])
        want = g[f"{case}_mlp_in"]
        feats = f.mlp_base.features(xu[:64])
        close(feats, want, 1e-6, "mlp_in (op chain)")
        if fused:
            assert f.mlp_base._can_fuse(xu)
            ff = f.mlp_base.features_fused(xu[:64].contiguous())
            close(ff[:, :want.shape[1]], want, 1e-6, "mlp_in (fused)")
            assert float(ff[:, want.shape[1]:].abs().max()) == 0.0 if ff.shape[1] > want.shape[1] else True
    for grad_mode in (False, True):          # the gradient-free path takes different kernels (density-only unit, fused head)
        with torch.set_grad_enabled(grad_mode):
            density, feat = f.query_density(x, return_feat=True)
            rgb, sigma = f(x, v)
            d_only = f.query_density(x)
        close(density, g[f"{case}_density"], 1e-4, "density")
        close(d_only, g[f"{case}_density"], 1e-4, "density (no feat)")
        close(feat[:96], g[f"{case}_feat"], 1e-4, "geo features")
        close(rgb, g[f"{case}_rgb"], 1e-4, "rgb")
        close(sigma, g[f"{case}_sigma"], 1e-4, "sigma")
        # outside the box the density is exactly zero (selector, ngp.py:524)
        assert np.array_equal(density.detach().cpu().numpy() == 0, g[f"{case}_density"] == 0)
    loss = (rgb * torch.from_numpy(g[f"{case}_w_rgb"]).to(cuda)).sum() + (sigma * torch.from_numpy(g[f"{case}_w_sig"]).to(cuda)).sum()
    assert abs(loss.item() - float(g[f"{case}_loss"])) <= 1e-4 * max(abs(float(g[f"{case}_loss"])), 1.0)
    f.zero_grad()
    loss.backward()
    for k, p in f.named_parameters():
        close(p.grad, g[f"{case}_grad_{k}"], 2e-4, "grad " + k)
    # ... and the other setting's golden is NOT matched (the two differ where the harmonics enter)
    other = np.load(os.path.join(GOLD, "field_toy.npz" if sh == "sh_half" else "field_toy_sh16.npz"))
    k = "mlp_head.0.weight"
    want = other[f"{case}_grad_{k}"]
    err = np.abs(dict(f.named_parameters())[k].grad.cpu().numpy() - want).max()
    assert err > 2e-3 * np.abs(want).max(), "the half / float harmonics are indistinguishable: the golden pins nothing"


def test_sh_convention_is_real_spherical_harmonics(cuda):
    """The direction encoding equals real spherical harmonics (Condon-Shortley phase, index l^2 + l + m) computed
    from scipy's complex harmonics — the convention tiny-cuda-nn documents; its fp16 output rounding aside."""
    from scipy.special import sph_harm_y

    from cnc_amd.field import SHEncoding
    d = np.random.default_rng(0).normal(size=(2000, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    theta, phi = np.arccos(np.clip(d[:, 2], -1, 1)), np.arctan2(d[:, 1], d[:, 0])
    want = np.empty((d.shape[0], 16))
    for l in range(4):
        for m in range(-l, l + 1):
            Y = sph_harm_y(l, abs(m), theta, phi)
            want[:, l * l + l + m] = Y.real if m == 0 else np.sqrt(2) * (Y.real if m > 0 else Y.imag)
    got = SHEncoding()(torch.from_numpy(((d + 1) / 2).astype(np.float32)).to(cuda)).cpu().numpy()
    assert np.abs(got - want).max() < 5e-6


# ------------------------------------------------------------------------------------------------------ render
def _estimator(cuda, g):
    from cnc_amd.nerfacc import OccGridEstimator
    est = OccGridEstimator(roi_aabb=torch.tensor(AABB), resolution=32, levels=1).to(cuda)
    b = torch.from_numpy(g["binaries"]).to(cuda)
    est.binaries = b
    est.occs = b.reshape(-1).float() * 0.02
    return est


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
def test_training_render_matches_reference(cuda, fused):
    from cnc_amd.render import Rays, render_image_with_occgrid
    g = np.load(os.path.join(GOLD, "render_toy.npz"))
    f = build_field(cuda, FIELD_CASES["f8"], fused, density_bias=float(g["density_bias"]))
    est = _estimator(cuda, g)
    o, d = torch.from_numpy(g["rays_o"]).to(cuda), torch.from_numpy(g["rays_d"]).to(cuda)
    bk = torch.from_numpy(g["bkgd"]).to(cuda)
    step = float(g["render_step_size"])
    f.train(); est.train()
    torch.manual_seed(101)
    with cpu_rand_like() as tape:
        rgb, acc, depth, n, extra = render_image_with_occgrid(f, est, Rays(o, d), near_plane=0.0, render_step_size=step,
                                                              render_bkgd=bk, cone_angle=0.0, alpha_thre=0.0, return_extra=True)
    assert tape.shapes == [(o.shape[0],)]                  # one jitter draw per ray, as the reference (occ_grid.py:176)
    assert n == int(g["train_n"])                          # sample count: exact
    close(rgb, g["train_rgb"], 1e-4, "rgb")
    close(acc, g["train_opacity"], 1e-4, "opacity")
    close(depth, g["train_depth"], 1e-4, "depth")
    close(extra["sigmas"], g["train_extra_sigmas"], 1e-4, "per-sample sigma")
    # the samples themselves: same rays, same t (the march replays the reference's float adds)
    torch.manual_seed(101)
    with cpu_rand_like():
        ri, ts, te = est.sampling(o, d, sigma_fn=lambda a, b_, r: f.query_density(o[r] + d[r] * (a + b_)[:, None] / 2.0).squeeze(-1),
                                  near_plane=0.0, render_step_size=step, stratified=True)
    assert np.array_equal(ri.cpu().numpy(), g["train_ray_indices"])
    assert np.array_equal(ts.cpu().numpy(), g["train_t_starts"]) and np.array_equal(te.cpu().numpy(), g["train_t_ends"])
    loss = F.mse_loss(rgb, torch.from_numpy(g["train_pixels"]).to(cuda))
    assert abs(loss.item() - float(g["train_loss"])) <= 1e-4 * float(g["train_loss"])
    f.zero_grad()
    loss.backward()
    close(f.mlp_base.network[0].weight.grad, g["train_grad_w0"], 2e-4, "grad base W0")
    close(f.mlp_head[4].weight.grad, g["train_grad_head_w2"], 2e-4, "grad head W2")
    close(f.mlp_base.encoding_xyz.params.grad, g["train_grad_xyz"], 2e-4, "grad xyz table")
    close(f.mlp_base.encoding_xz.params.grad, g["train_grad_xz"], 2e-4, "grad xz table")


def test_evaluation_renders_match_reference(cuda):
    from cnc_amd.render import Rays, render_image_with_occgrid, render_image_with_occgrid_test
    g = np.load(os.path.join(GOLD, "render_toy.npz"))
    f = build_field(cuda, FIELD_CASES["f8"], True, density_bias=float(g["density_bias"]))
    est = _estimator(cuda, g)
    o, d = torch.from_numpy(g["rays_o"]).to(cuda).view(16, 16, 3), torch.from_numpy(g["rays_d"]).to(cuda).view(16, 16, 3)
    bk = torch.from_numpy(g["bkgd"]).to(cuda)
    step = float(g["render_step_size"])
    f.eval(); est.eval()
    with torch.no_grad():
        rgb, acc, depth, n = render_image_with_occgrid(f, est, Rays(o, d), near_plane=0.0, render_step_size=step,
                                                       render_bkgd=bk, test_chunk_size=96)
    assert n == int(g["eval_n"])
    close(rgb, g["eval_rgb"], 1e-4, "rgb"); close(acc, g["eval_opacity"], 1e-4, "opacity"); close(depth, g["eval_depth"], 1e-4, "depth")
    for tag in ("t0", "t1"):
        rgb, acc, depth, n = render_image_with_occgrid_test(1024, f, est, Rays(o, d), near_plane=0.0, render_step_size=step,
                                                            render_bkgd=bk, alpha_thre=float(g[f"test_{tag}_alpha_thre"]))
        assert n == int(g[f"test_{tag}_n"]), tag
        close(rgb, g[f"test_{tag}_rgb"], 1e-4, tag + " rgb")
        close(acc, g[f"test_{tag}_opacity"], 1e-4, tag + " opacity")
        close(depth, g[f"test_{tag}_depth"], 1e-4, tag + " depth")


# -------------------------------------------------------------------------------------------------- trajectory
class _NumpyBall:
    """fetch() / update_num_rays() over make_golden_field.ball_batch: batch k is the k-th call."""

    def __init__(self, device):
        self.device, self.k, self.num_rays = device, 0, 0

    def update_num_rays(self, n):
        self.num_rays = int(n)

    def fetch(self):
        from cnc_amd.render import Rays
        o, d, pix, bk = (torch.from_numpy(a).to(self.device) for a in ball_batch(self.k, self.num_rays))
        self.k += 1
        return {"rays": Rays(o, d), "pixels": pix, "color_bkgd": bk}


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("tag", ["ref", "fast"])
def test_training_trajectory(cuda, tag, fused, tmp_path):
    """The reference's loop body (train:302-366) on its own classes vs `Trainer.train_step`, same batches, same
    random stream, same initial state.  As long as every table entry has the reference's sign (the first 4-8
    steps) everything agrees to 1e-4 / exactly; after that the band is the reference's own spread under a few-ulp
    perturbation (see below and make_golden_field.py)."""
    from cnc_amd.trainer import TrainConfig, Trainer
    g = np.load(os.path.join(GOLD, f"train_toy_{tag}.npz"))
    c = TRAIN
    steps = int(g["steps"])
    cfg = TrainConfig(lmbda=c["lmbda"], Pg_level=6, Pg_level_2D=4, log2_hashmap_size=c["T3"], log2_hashmap_size_2D=c["T2"],
                      sample_num=c["sample_num"], max_context_layer_num=3, n_features=c["F"], n_neurons=c["n_neurons"],
                      fused_features=fused, resolutions_list=tuple(c["res3"]), resolutions_list_2D=tuple(c["res2"]),
                      step_update=c["step_update"], skip_levels_3D=(0, 1, 2), skip_levels_2D=(0,),
                      init_batch_size=c["init_batch_size"], target_sample_batch_size=c["target"], weight_decay=c["weight_decay"],
                      grid_resolution=c["Rb"], render_step_size=c["render_step_size"], lr=c["lr"],
                      milestones=tuple(c["milestones"]), warmup_iters=int(g["warmup_iters"]),
                      dimension_wise_resolution=c["fine"], out_dir=str(tmp_path),
                      sh_fp16_round=False)          # the golden run's tinycudann stand-in returned float32
    tr = Trainer(cfg, device=cuda, dataset=_NumpyBall(cuda))
    tr.ctx_thread = False          # one host thread: the reference's ORDER of random draws (render pass, then context pass)
    torch.manual_seed(11)                              # the context tables' CPU draws, as the golden run
    tr.context = tr.build_context()
    tr.context.MAX_POINTS_NUM_TO_OOM = c["max_pts"]
    tr.context.rand_like = lambda t: torch.rand_like(t)     # resolved at call time: the CPU replay below
    tr.context.load_state_dict({k[7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ctx_sd_")}, strict=True)
    tr.build_optimizers()
    tr.build_sinks()               # the entropy pass's gradient sink holds the NEW context heads: the planes' graph records
    sd = tr.field.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["field_keys"]]
    filled = fill_state(sd, seed=23)
    for k in filled:
        if k.endswith(".params"):
            filled[k] = filled[k] * (1e-4 / 1.3)
    tr.field.load_state_dict(filled, strict=True)
    rec = {k: [] for k in ("mse", "bpp", "mb", "n_samples", "num_rays", "occupied", "lr")}
    signs, norms, values = [], [], []
    torch.manual_seed(29)
    with cpu_rand_like() as tape:
        for step in range(steps):
            rec["num_rays"].append(tr.dataset.num_rays)
            rec["lr"].append(tr.opt.param_groups[0]["lr"])
            s = tr.train_step(step)
            assert s is not None
            rec["mse"].append(s["mse"]); rec["bpp"].append(s["bpp"]); rec["mb"].append(s["embed_bits_MB"])
            rec["n_samples"].append(s["n_rendering_samples"]); rec["occupied"].append(int(tr.estimator.binaries.sum()))
            e = tr.field.mlp_base
            tabs = (e.encoding_xyz, e.encoding_xy, e.encoding_xz, e.encoding_yz)
            flat = torch.cat([t.params.detach().reshape(-1) for t in tabs])
            signs.append(np.packbits((flat >= 0).cpu().numpy()))
            values.append(flat.cpu().numpy())
            norms.append([float(p.detach().double().norm()) for p in list(tr.field.parameters())[4:]] +
                         [float(p.detach().double().norm()) for p in tr.context.parameters()])
            if step == 0:
                assert np.array_equal(tr.estimator.binaries.cpu().numpy(), g["step0_binaries"])
    # the schedule under test is the one that ships: the planes' half of the entropy pass replayed from its captured graph
    # on every step between two occupancy refreshes (a capture that fails raises here: CNC_PLANES_GRAPH_STRICT)
    assert tr.planes_graph is not None and tr.planes_graph.captures > 0 and tr.planes_graph.replays > 0, \
        (tr.planes_graph and (tr.planes_graph.captures, tr.planes_graph.replays))
    want_shapes = [tuple(int(v) for v in s.split(",")) for s in g["rand_like_shapes"]]
    # the same draws in the same order as the reference; the per-ray jitter draw has the step's ray count
    assert len(tape.shapes) == len(want_shapes)
    assert [s for s in tape.shapes if len(s) != 1 or s[0] < 16] == [s for s in want_shapes if len(s) != 1 or s[0] < 16]
    r = {k: np.asarray(v, np.float64) for k, v in rec.items()}
    n_bits = values[0].shape[0]
    differ = [int(np.unpackbits(a ^ b)[:n_bits].sum()) for a, b in zip(signs, g["signs"])]
    norm_dev = np.abs(np.asarray(norms) / g["norms"] - 1).max(axis=1)
    if os.environ.get("CNC_TRAJ_PRINT"):
        print("table signs that differ from the reference after each step:", differ)
        print("largest relative deviation of a dense parameter's norm after each step:", np.array2string(norm_dev, precision=2))
        first = next((k for k, d in enumerate(differ) if d), None)
        if first is not None:
            idx = np.nonzero(np.unpackbits(signs[first] ^ g["signs"][first])[:n_bits])[0]
            F_ = c["F"]
            n3 = tr.field.mlp_base.encoding_xyz.params.numel()
            print("first step with a differing sign:", first, "entries", idx[:10], "of", n_bits, "(3-D table holds", n3, ")")
            for j in idx[:6]:
                hist = [float(v[j]) for v in values[max(0, first - 3): first + 1]]
                print("   entry", int(j), "row", int(j) // F_, "feature", int(j) % F_, "my values over the last steps:", hist)
        for k in ("mse", "bpp", "n_samples", "num_rays", "occupied"):
            print(k, "got ", np.array2string(r[k], precision=5, max_line_width=250))
            print(k, "want", np.array2string(g[k], precision=5, max_line_width=250))
    # ---- while no table entry has a different sign: the same model, so the same numbers (at least the first 3 steps)
    agree = next((k for k, d in enumerate(differ) if d), steps)
    print("steps before the first differing sign:", agree, "(tag", tag, "fused", fused, ")")
    # measured on every run so far: 8 (reference schedule) / 4 (compressed warm-up) — rounds 3 and 4, with and without
    # the fused field kernel in the sampler; one step of slack
    assert agree >= (7 if tag == "ref" else 3), differ
    for k in range(agree):
        assert r["n_samples"][k] == g["n_samples"][k] and r["num_rays"][k] == g["num_rays"][k], k
        assert r["occupied"][k] == g["occupied"][k], k
        for name in ("mse", "bpp", "mb"):
            assert abs(r[name][k] - g[name][k]) <= 1e-4 * g[name][k], (name, k, r[name][k], g[name][k])
        assert norm_dev[k] <= 1e-5, (k, norm_dev[k])              # every MLP tensor of the field and the context models
    assert np.allclose(r["lr"], g["lr"], rtol=1e-12, atol=0)
    # ---- afterwards: binarised tables + Adam are a chaotic system (an entry that lands within rounding of zero takes the
    # other sign, the next steps amplify it).  The golden holds the reference's OWN spread: two more runs of the
    # reference whose dense weights were perturbed by 2^-22 relative (a few ulps) at the start.  The run under test
    # must stay about as close to the unperturbed reference as those do (sign differences x1.5 + 2, series maxima x3: the
    # GPU run is itself not bit-reproducible (float atomics), so it is one more sample of the same spread).
    spread_signs = np.max([[int(np.unpackbits(a ^ b)[:n_bits].sum()) for a, b in zip(g["signs"], g[f"noise{k}_signs"])]
                           for k in (1, 2)], axis=0)
    assert all(d <= 1.5 * sp + 2 for d, sp in zip(differ, np.maximum.accumulate(spread_signs))), (differ, spread_signs.tolist())
    first_noise = min(next((k for k, d in enumerate(np.unpackbits(g["signs"] ^ g[f"noise{q}_signs"], axis=1)[:, :n_bits].sum(1)) if d), steps)
                      for q in (1, 2))
    assert agree >= min(first_noise, 3), (agree, first_noise)
    dev = {k: float(np.abs(r[k] / g[k] - 1).max()) for k in ("mse", "bpp", "mb", "n_samples", "num_rays")}
    ref_dev = {k: max(float(np.abs(g[f"noise{q}_{k}"] / g[k] - 1).max()) for q in (1, 2)) for k in dev}
    print(tag, "fused" if fused else "unfused", "first differing sign at step", agree, "(perturbed reference:", first_noise, ")",
          "max relative deviation per series:", {k: round(v, 4) for k, v in dev.items()}, "reference's own spread:",
          {k: round(v, 4) for k, v in ref_dev.items()}, "signs differing at the end:", differ[-1], "vs", int(spread_signs[-1]))
    for k in dev:      # per-step maxima of chaotic series, two samples of the reference's spread: a factor 3
        assert dev[k] <= 3.0 * ref_dev[k] + 1e-3, (k, dev[k], ref_dev[k])
    occ_dev = max(float(np.abs(g[f"noise{q}_occupied"] - g["occupied"]).max()) for q in (1, 2))
    assert np.abs(r["occupied"] - g["occupied"]).max() <= 1.5 * occ_dev + 2
    # the smoothed end of the run: the mean mse of the last 10 steps (the per-step bpp swings between 0.7 and 2.0 with the
    # context window drawn — its 10-step mean is no steadier than the series itself and is not asserted)
    mine, want = r["mse"][-10:].mean(), g["mse"][-10:].mean()
    own = max(abs(g[f"noise{q}_mse"][-10:].mean() / want - 1) for q in (1, 2))
    assert abs(mine / want - 1) <= 3.0 * own + 0.10, (mine, want, own)
