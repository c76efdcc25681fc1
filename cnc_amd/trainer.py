"""End-to-end driver for the CNC protocol: train -> evaluate -> encode -> decode -> evaluate.

Follows the behaviour of examples/train_CNC_nerf_synthetic.py (hyper-parameters :135-186, optimisers
and schedules :257-297, loop :302-366, evaluation / codec :384-506), not its text: the reference
script cannot travel to the GPU box and its datasets are not available offline, so the scene here is
a procedural one (`SyntheticBallDataset`, same `fetch`-style interface as
examples/datasets/nerf_synthetic.py:132-239).  Flag names of the reference's argparse are kept in
`TrainConfig`.

Data parallelism (new): one process per GPU; each rank draws its own rays, gradients of
ALL parameters live in one flat bucket that is all-reduced once per step (cnc_amd.dist.GradBucket);
the occupancy grid and the context-window draw are broadcast from rank 0 so replicas stay identical.
"""
from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import _gradsink, _table_adam
from . import dist as cdist
from .context import CNC_context_models
from .field import NGPRadianceField_mygrid_2D3D
from .nerfacc import OccGridEstimator
from .render import Rays, render_image_with_occgrid, render_image_with_occgrid_test, set_random_seed


@dataclass
class TrainConfig:
    # reference flags (train_CNC_nerf_synthetic.py:71-133)
    scene: str = "ball"
    lmbda: float = 2e-3
    Pg_level: int = 12
    Pg_level_2D: int = 4
    log2_hashmap_size: int = 19
    log2_hashmap_size_2D: int = 17
    sample_num: int = 200000
    max_context_layer_num: int = 3
    n_features: int = 4
    fused_features: bool = True      # encoders write straight into the base MLP's input matrix
    sh_fp16_round: bool = True       # direction encoding rounded through half, as tiny-cuda-nn hands it to the reference
    # hard-coded in the reference (:135-186)
    n_neurons: int = 160
    resolutions_list: Tuple[int, ...] = (18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514)
    resolutions_list_2D: Tuple[int, ...] = (130, 258, 514, 1026)
    step_update: int = 16
    skip_levels_3D: Tuple[int, ...] = (0, 1, 2)
    skip_levels_2D: Tuple[int, ...] = (0,)
    max_steps: int = 20000
    init_batch_size: int = 1024
    target_sample_batch_size: int = 1 << 18
    weight_decay: float = 2e-6
    aabb: Tuple[float, ...] = (-1.5, -1.5, -1.5, 1.5, 1.5, 1.5)
    near_plane: float = 0.0
    far_plane: float = 1.0e10
    grid_resolution: int = 128
    grid_nlvl: int = 1
    render_step_size: float = 5e-3
    alpha_thre: float = 0.0
    cone_angle: float = 0.0
    lr: float = 6e-3
    milestones: Tuple[int, ...] = (9000, 12000, 15000, 17000, 19000)
    warmup_iters: int = 1000
    seed: int = 42
    # evaluation
    test_views: int = 4
    image_size: int = 200
    dimension_wise_resolution: Optional[int] = None   # default: finest 3-D resolution
    out_dir: str = "./bitstreams/ball"
    log_every: int = 200


class SyntheticBallDataset:
    """Procedural scene: an opaque textured ball of radius 0.8 in front of a white background,
    seen from cameras on a sphere of radius 4 (focal as nerf_synthetic, camera_angle_x=0.6911).
    `fetch(num_rays)` returns random training pixels like SubjectLoader.fetch_data in training mode
    (nerf_synthetic.py:164-239); `view(i)` returns a whole test image."""

    RADIUS = 0.8

    def __init__(self, image_size=200, n_train_views=100, device="cuda", seed=0):
        self.H = self.W = image_size
        self.focal = 0.5 * image_size / math.tan(0.5 * 0.6911)
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        g = torch.Generator().manual_seed(1234)
        self.train_c2w = self._poses(n_train_views, g).to(self.device)
        self.test_c2w = self._poses(16, torch.Generator().manual_seed(4321)).to(self.device)
        self.num_rays = 1024

    @staticmethod
    def _poses(n, g):
        az = torch.rand(n, generator=g) * 2 * math.pi
        el = (torch.rand(n, generator=g) - 0.3) * 1.2
        eye = 4.0 * torch.stack([torch.cos(el) * torch.cos(az), torch.cos(el) * torch.sin(az), torch.sin(el)], -1)
        fwd = -eye / eye.norm(dim=-1, keepdim=True)
        up0 = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
        right = torch.linalg.cross(fwd, up0)
        right = right / right.norm(dim=-1, keepdim=True)
        up = torch.linalg.cross(right, fwd)
        return torch.cat([torch.stack([right, up, -fwd], dim=-1), eye[..., None]], dim=-1)   # [n,3,4]

    def update_num_rays(self, n):
        self.num_rays = int(n)

    def _rays(self, c2w, x, y):
        cam = torch.stack([(x - self.W / 2 + 0.5) / self.focal, -(y - self.H / 2 + 0.5) / self.focal,
                           -torch.ones_like(x)], dim=-1)
        d = (cam[:, None, :] * c2w[:, :3, :3]).sum(-1)
        d = d / torch.linalg.norm(d, dim=-1, keepdim=True)
        o = c2w[:, :3, 3].expand_as(d)
        return o.contiguous(), d.contiguous()

    def _shade(self, o, d):
        """Ground-truth pixel colour and alpha: first hit of the ball."""
        b = (o * d).sum(-1)
        c = (o * o).sum(-1) - self.RADIUS ** 2
        disc = b * b - c
        hit = disc > 0
        t = -b - torch.sqrt(disc.clamp_min(0))
        p = o + d * t[:, None]
        n = p / self.RADIUS
        if getattr(self, "_shade_consts", None) is None or self._shade_consts[0].device != p.device:
            self._shade_consts = (torch.tensor([0.0, 2.0, 4.0], device=p.device),
                                  torch.tensor([0.3, 0.5, 0.8], device=p.device))
        phase, light = self._shade_consts
        tex = 0.5 + 0.5 * torch.sin(p * 9.0 + phase)
        lam = (0.35 + 0.65 * (n * light).sum(-1).clamp(0, 1))[:, None]
        rgb = (tex * lam).clamp(0, 1)
        return rgb, hit.float()[:, None]

    def _train_images(self):
        """The training views as RGBA images in device memory, as SubjectLoader holds its images
        (nerf_synthetic.py:133-146): the analytic shading evaluated once per pixel instead of once per fetched ray."""
        if getattr(self, "_images", None) is None:
            ys, xs = torch.meshgrid(torch.arange(self.H, device=self.device),
                                    torch.arange(self.W, device=self.device), indexing="ij")
            x, y = xs.reshape(-1).float(), ys.reshape(-1).float()
            views = []
            for c2w in self.train_c2w:
                o, d = self._rays(c2w[None].expand(x.shape[0], 3, 4), x, y)
                views.append(torch.cat(self._shade(o, d), dim=-1).view(self.H, self.W, 4))
            self._images = torch.stack(views)
        return self._images

    def fetch(self, num_rays=None):
        n = self.num_rays if num_rays is None else num_rays
        img = torch.randint(0, self.train_c2w.shape[0], (n,), device=self.device, generator=self.gen)
        xi = torch.randint(0, self.W, (n,), device=self.device, generator=self.gen)
        yi = torch.randint(0, self.H, (n,), device=self.device, generator=self.gen)
        o, d = self._rays(self.train_c2w[img], xi.float(), yi.float())
        rgba = self._train_images()[img, yi, xi]            # the same numbers as shading the fetched rays
        rgb, alpha = rgba[:, :3], rgba[:, 3:]
        bkgd = torch.rand(3, device=self.device, generator=self.gen)     # random bkgd in training
        return {"rays": Rays(o, d), "pixels": rgb * alpha + bkgd * (1 - alpha), "color_bkgd": bkgd}

    def view(self, i):
        ys, xs = torch.meshgrid(torch.arange(self.H, device=self.device),
                                torch.arange(self.W, device=self.device), indexing="ij")
        x, y = xs.reshape(-1).float(), ys.reshape(-1).float()
        c2w = self.test_c2w[i % self.test_c2w.shape[0]][None].expand(x.shape[0], 3, 4)
        o, d = self._rays(c2w, x, y)
        rgb, alpha = self._shade(o, d)
        bkgd = torch.ones(3, device=self.device)
        return {"rays": Rays(o.view(self.H, self.W, 3), d.view(self.H, self.W, 3)),
                "pixels": (rgb * alpha + bkgd * (1 - alpha)).view(self.H, self.W, 3), "color_bkgd": bkgd}


def quantize_params(state: Dict[str, torch.Tensor], digits=13):
    """Uniform `digits`-bit quantisation of each MLP tensor (train_CNC_nerf_synthetic.py:30-50).
    Returns (quantised MB, original MB, quantised state)."""
    bits = bits_orig = 0
    out = {}
    for n, p in state.items():
        lo, hi = torch.min(p), torch.max(p)
        interval = (hi - lo) / (2 ** digits - 1) + 1e-6
        q = (p - lo) // interval
        out[n] = q * interval + lo
        bits += digits * p.numel() + 64
        bits_orig += 32 * p.numel()
    return bits / 8.0 / 1024 / 1024, bits_orig / 8.0 / 1024 / 1024, out


def get_binary_vxl_size(binary_vxl):
    """Entropy bound of the occupancy grid in MB (train_CNC_nerf_synthetic.py:53-68)."""
    with torch.no_grad():
        n = binary_vxl.numel()
        pos = torch.sum(binary_vxl)
        Pg = pos / n
        bits = pos * (-torch.log2(Pg)) + (n - pos) * (-torch.log2(1 - Pg)) + 32
    return Pg, bits.item() / 8.0 / 1024 / 1024, n


class LoaderDataset:
    """`SubjectLoader` / `SubjectLoader_Tanks` (cnc_amd.datasets) behind the three calls the Trainer makes:
    `fetch()` = one training batch as the reference draws it (`train_dataset[randint(len)]`,
    train_CNC_nerf_synthetic.py:305-306), `view(i)` = test image i, `update_num_rays`."""

    def __init__(self, train_loader, test_loader):
        self.train, self.test = train_loader, test_loader
        self._gen = None

    def seed_sampling(self, seed: int):
        """Per-rank stream for the image index drawn here and for the loader's pixel draws (data parallelism:
        without it every rank would render the same batch and the all-reduce would average N copies of one
        gradient)."""
        self._gen = torch.Generator().manual_seed(int(seed))
        self.train.seed_sampling(int(seed) + 1)

    def update_num_rays(self, n):
        self.train.update_num_rays(int(n))

    def fetch(self):
        return self.train[int(torch.randint(0, len(self.train), (1,), generator=self._gen).item())]

    def view(self, i):
        return self.test[i % len(self.test)]

    def __len__(self):
        return len(self.test)


_STEP_STREAMS = {}


def reserve_streams(device):
    """The two side streams of the training step on `device` (entropy pass; its planes' half), created — and used once —
    NOW, one pair per process and device.  Why there is such a call: the HIP runtime deals a process's streams onto four
    hardware queues in the order they first appear, the null stream included, and two streams on one queue run one after
    the other.  A Trainer made first gets a queue per stream (null, entropy, planes: three of four); made after other
    code has used streams of its own (the encoder backward's two side streams in bench.py's frame loop) its planes' stream
    landed on the null stream's queue and the step took 9.7 ms instead of 7.5.  A process that trains creates its Trainer
    first anyway; one that does other GPU work first calls this at start-up."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return None
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    dev = torch.device("cuda", key)
    if key not in _STEP_STREAMS:
        prio = int(os.environ.get("CNC_CTX_STREAM_PRIORITY", "0"))
        pair = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(3)]      # entropy pass, its planes' half, look-ahead
        for st in pair:
            with torch.cuda.stream(st):
                torch.zeros(1, device=dev)
        _STEP_STREAMS[key] = pair
    return _STEP_STREAMS[key]


class Trainer:
    def __init__(self, cfg: TrainConfig, device="cuda", dataset=None):
        self.cfg = cfg
        # world > 1: join the process group (RCCL; gloo under CNC_DIST_BACKEND) and take this rank's GPU
        self.rank, self.local_rank, self.world = cdist.init()
        self.device = torch.device(device)
        self.dp = self.world > 1 or cdist.forced()       # data-parallel control flow (forced: a one-rank group, test hook)
        if self.dp and self.device.type == "cuda":
            self.device = torch.device("cuda", cdist.local_device_index())
        elif self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())     # the worker threads select it by index
        set_random_seed(cfg.seed)
        c = cfg
        aabb = torch.tensor(c.aabb, device=self.device)
        self.estimator = OccGridEstimator(roi_aabb=aabb, resolution=c.grid_resolution, levels=c.grid_nlvl).to(self.device)
        self.field = NGPRadianceField_mygrid_2D3D(
            aabb=self.estimator.aabbs[-1], n_features_per_level=c.n_features, n_neurons=c.n_neurons,
            resolutions_list=c.resolutions_list, log2_hashmap_size=c.log2_hashmap_size,
            resolutions_list_2D=c.resolutions_list_2D, log2_hashmap_size_2D=c.log2_hashmap_size_2D,
            ste_binary=True, Q=10, fused_features=c.fused_features, sh_fp16_round=c.sh_fp16_round).to(self.device)
        self.context = self.build_context()
        # `dataset`: anything with fetch() / view(i) / update_num_rays(n) (LoaderDataset for real scenes); the
        # procedural scene otherwise (no dataset ships with the repository)
        self.dataset = dataset if dataset is not None else \
            SyntheticBallDataset(c.image_size, device=self.device, seed=c.seed + 1000 * self.rank)
        self.dataset.update_num_rays(c.init_batch_size)

        self.build_optimizers()
        self.loss_scale = 2.0 ** 10          # GradScaler(2**10), never unscaled (train:211,361-362)

        # the planes' half of the entropy pass as one captured graph per refresh interval (CNC_PLANES_GRAPH=0: op by op)
        self.planes_graph = None
        # ... in the data-parallel step as well (CNC_PLANES_GRAPH_DP=0: the joint entropy pass there), so that the step a
        # multi-GPU run measures is the single-GPU step + the exchange
        self.planes_graph_dp = os.environ.get("CNC_PLANES_GRAPH_DP", "1") == "1"
        self._pool_graph = None
        self._planes_replayed = False
        self._fwd_enqueued = None
        if self.device.type == "cuda" and os.environ.get("CNC_PLANES_GRAPH", "1") == "1":
            from ._planes_graph import PlanesGraph
            self.planes_graph = PlanesGraph(self)
        self.prefetch = os.environ.get("CNC_PREFETCH_BATCH", "1") == "1"
        self._next_data = None
        self._next_ready = None
        # the next batch's draw and march on a stream of their own (CNC_PREMARCH=0: the batch only, on the main stream)
        self.premarch = os.environ.get("CNC_PREMARCH", "1") == "1"
        self.ahead_stream = None
        if self.device.type == "cuda" and self.premarch:
            self.ahead_stream = reserve_streams(self.device)[2]
        # The entropy pass (context forward and backward) runs on its own stream next to the render pass — see train_step
        self.ctx_stream = None
        if self.device.type == "cuda" and os.environ.get("CNC_CTX_STREAM", "1") == "1":
            self.ctx_stream = reserve_streams(self.device)[0]
        # ... and its planes' half on a third one (CNC_CTX_STREAM_2D=0: both halves on the side stream, one after the other)
        self.ctx_stream_2D = None
        if self.ctx_stream is not None and os.environ.get("CNC_CTX_STREAM_2D", "1") == "1":
            self.ctx_stream_2D = reserve_streams(self.device)[1]
        # leaves are accumulated on the main stream, the entropy pass produces its gradients on the side stream: intended.
        # The switch is process-global, so it is held only for the duration of a train_step (see there).
        self._warn_switch = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None) \
            if self.ctx_stream is not None else None
        # ... and from its own host thread, started before the render pass (`_context_pass`): the two passes are ~300
        # launches each and the step is otherwise bound by the host issuing them one after the other.  Off = the
        # sequential schedule, which keeps the reference's order of random draws (the trajectory goldens need it).
        self.ctx_thread = self.ctx_stream is not None and os.environ.get("CNC_CTX_THREAD", "1") == "1"
        self._pool = None
        # per-step gradient sinks (cnc_amd._gradsink): the encoder scatters and the context heads' weight gradients add
        # into ONE buffer per parameter and pass instead of a fresh zero-filled tensor per call (CNC_GRAD_SINK=0: off)
        self.build_sinks()
        self.bucket = None
        self.time_comm = False          # bench hook: HIP events around the wait for the gradient all-reduce
        self._comm_events = []
        if self.dp:
            plist = list(self.field.parameters()) + list(self.context.parameters())
            # ray-loss gradients (all-reduced) + ONE tail slot: this rank's sample count, so that the sum over the
            # ranks arrives with the gradients instead of through a blocking collective in the middle of the step
            self.bucket = cdist.GradBucket(plist, tail=1)
            self.bucket_ctx = cdist.GradBucket(plist)      # entropy-loss gradients (replica-identical)
            self._count_host = torch.zeros(1, dtype=torch.float32)
            if self.device.type == "cuda":
                self._count_host = self._count_host.pin_memory()
            self._count_pending = None                     # (event, num_rays of the step the count belongs to)
            # how often the replicas had to be re-aligned (cdist.resync_parameters, every `step_update` steps)
            self.resync = {"checks": 0, "fired": 0, "tensors": 0, "bytes": 0}
            base = self.context.rand_like

            def synced_rand_like(t):
                # the threaded schedule draws (and broadcasts) on the MAIN thread before it forks: collectives of one
                # communicator must be issued in the same order on every rank, so the worker thread issues none
                if self._ctx_rand is not None:
                    r, self._ctx_rand = self._ctx_rand, None
                    return r
                r = base(t)
                torch.distributed.broadcast(r, 0)
                return r
            self._ctx_rand = None
            self.context.rand_like = synced_rand_like

    def build_context(self):
        """The context models of this configuration (train:221-241).  Their vertex tables draw from the CPU
        generator (randperm of the dense levels, utils_bpp_acc.py:312), so a caller that wants a particular
        draw seeds, calls this and assigns the result to `self.context` (then `build_optimizers()`)."""
        c = self.cfg
        return CNC_context_models(
            num_dim=3, resolutions_list=c.resolutions_list, resolutions_list_2D=c.resolutions_list_2D,
            log2_hashmap_size=c.log2_hashmap_size, log2_hashmap_size_2D=c.log2_hashmap_size_2D,
            n_features=c.n_features, sample_num=c.sample_num, max_context_layer_num=c.max_context_layer_num,
            ste_binary=True, Q=10, Pg_level=c.Pg_level, Pg_level_2D=c.Pg_level_2D, Rb=c.grid_resolution,
            step_update=c.step_update, skip_levels_3D=c.skip_levels_3D, skip_levels_2D=c.skip_levels_2D,
            device=self.device,
            dimension_wise_resolution=c.dimension_wise_resolution or c.resolutions_list[-1])

    def build_sinks(self):
        """The step's gradient sinks for the CURRENT field and context models (a caller that replaces `self.context` —
        see `build_context` — calls this and `build_optimizers()` again: the entropy pass's sink holds one slot per
        context-head parameter, and the planes' graph refuses to record without them)."""
        self.sink_render = self.sink_ctx = None
        if self.device.type == "cuda" and os.environ.get("CNC_GRAD_SINK", "1") == "1":
            tables = [e.params for e in self.field.mlp_base._encoders()]
            self.sink_render = _gradsink.GradSink(tables, [])
            self.sink_ctx = _gradsink.GradSink(tables, list(self.context.parameters()))

    def build_optimizers(self):
        """Both Adam groups and their chained schedules (train:257-297)."""
        c = self.cfg
        # one kernel per parameter list instead of the ~9 passes of the foreach implementation (0.8 -> 0.2 ms per step)
        one_pass = self.device.type == "cuda" and os.environ.get("CNC_FUSED_ADAM", "1") == "1"
        # the four tables as a parameter group of their own (same hyper-parameters, same schedule): what `_table_adam` steps
        tables = [e.params for e in self.field.mlp_base._encoders()]
        tids = {id(p) for p in tables}
        rest = [p for p in self.field.parameters() if id(p) not in tids]
        self.opt = torch.optim.Adam([{"params": rest}, {"params": tables}], lr=c.lr, eps=1e-15, weight_decay=c.weight_decay,
                                    fused=one_pass)
        self.opt2 = torch.optim.Adam(self.context.parameters(), lr=c.lr, eps=1e-15, fused=one_pass)
        # Single-process steps with the gradient sinks: the tables' update reads the gradient pieces where they lie (one
        # kernel instead of clone + multi-tensor add + the library's Adam over the sum; CNC_TABLE_ADAM=0 or
        # `self.fused_table_adam = False`: pieces flushed into `.grad`, library step — what data-parallel steps do)
        self.table_adam = None
        if one_pass and not self.dp and os.environ.get("CNC_TABLE_ADAM", "1") == "1" \
                and all(t.numel() % 4 == 0 for t in tables):
            self.table_adam = _table_adam.TableAdam(self.opt, tables, self.field.mlp_base._encoders())
        self.fused_table_adam = self.table_adam is not None

        def sched(o):
            return torch.optim.lr_scheduler.ChainedScheduler([
                torch.optim.lr_scheduler.LinearLR(o, start_factor=0.01, total_iters=c.warmup_iters),
                torch.optim.lr_scheduler.MultiStepLR(o, milestones=list(c.milestones), gamma=0.33)])
        self.sched, self.sched2 = sched(self.opt), sched(self.opt2)

    # -------------------------------------------------------------------------------- training
    def _context_pass_worker(self, step, fork, params, grad_mode, autocast):
        """`_context_pass` on the worker thread with the submitting thread's grad and autocast modes (both are
        thread-local in PyTorch and a fresh thread starts from the defaults)."""
        enabled, dtype = autocast
        with torch.set_grad_enabled(grad_mode), torch.autocast(self.device.type, dtype=dtype, enabled=enabled):
            return self._context_pass(step, fork, params)

    def _planes_thread_step(self, step: int, params) -> bool:
        """The planes' half of this step's entropy pass runs apart from the 3-D half (its own root, its own thread)."""
        c = self.cfg
        # (data parallel, `params` given: the 3-D half's gradients are returned to the caller, the planes' half leaves its own
        # in the sink and in the graph's static tensors as in a single-process step — both are added behind the collective)
        # On by default since round 6 (CNC_PLANES_GRAPH_DP=0 switches it off): RCCL at one rank (tests/test_gpu_rccl.py), 2 and
        # 8 ranks over gloo on one device (tests/test_gpu_multi.py: bit-identical replicas across five refreshes).
        if params is not None and not self.planes_graph_dp:
            return False
        return (self.planes_graph is not None and self.ctx_stream_2D is not None and c.lmbda > 0
                and step > c.step_update and torch.is_grad_enabled())

    def _planes_graph_step(self, step: int, params) -> bool:
        """... as a replay of the recorded graph: every such step but the occupancy-refresh ones."""
        return self._planes_thread_step(step, params) and step % self.cfg.step_update != 0

    def _on_planes_thread(self, fn, *args):
        """Submit `fn(*args)` to the planes' thread under the SUBMITTING thread's grad and autocast modes (thread-local in
        PyTorch; a pool thread starts from the defaults)."""
        if self._pool_graph is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool_graph = ThreadPoolExecutor(max_workers=1, thread_name_prefix="cnc-planes")
        grad_mode = torch.is_grad_enabled()
        enabled, dtype = torch.is_autocast_enabled(self.device.type), torch.get_autocast_dtype(self.device.type)

        def job():
            with torch.set_grad_enabled(grad_mode), torch.autocast(self.device.type, dtype=dtype, enabled=enabled):
                return fn(*args)
        return self._pool_graph.submit(job)

    def _planes_refresh_job(self, after, step: int) -> None:
        """On the planes' thread, an occupancy-refresh step: rebuild what the planes' half is built on (vote plan, projections,
        vertex lists: host round trips on the planes' stream only) and run that half op by op — next to the 3-D half, which
        needs none of it, instead of in front of it on one thread (the entropy pass's thread was the long pole of a refresh
        step: 14 ms against 7.5; 12 this way).  The graph for the steps that follow is recorded by the next step, in front
        of its fork: recording it here, behind this job, was built too — the 2.5 ms of host time it takes moved from the next
        step into this one (whose long pole is this thread), the sum did not change."""
        torch.cuda.set_device(self.device)
        e = self.field.mlp_base
        with torch.cuda.stream(self.ctx_stream_2D), _gradsink.activate(self.sink_ctx):
            self.ctx_stream_2D.wait_event(after)
            self.context.refresh_planes(self.estimator.binaries, step, e.encoding_xy.params)
            if self.planes_graph.ready():          # the grid did not change: the recorded graph still stands
                self.planes_graph.replay()
            else:
                self.planes_graph.run_eager()

    def _replay_planes(self, after) -> None:
        """On the planes' thread: the graph launch on the planes' stream, ordered after the event `after`."""
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(self.ctx_stream_2D), _gradsink.activate(self.sink_ctx):
            self.ctx_stream_2D.wait_event(after)
            self.planes_graph.replay()

    def _ensure_planes_graph(self, step: int, params) -> None:
        """Capture the planes' graph if this step replays one and the structures it was recorded for are gone (the first
        step behind an occupancy refresh).  On the thread that calls it, with NO other thread of the step running: a
        capture puts the device's default random generator into capture mode for its duration, and a draw from another
        thread (the sampler's jitter) fails meanwhile — so train_step calls this before it forks the entropy pass off."""
        if self._planes_graph_step(step, params) and not self.planes_graph.ready():
            try:
                with torch.cuda.stream(self.ctx_stream_2D), _gradsink.activate(self.sink_ctx):
                    self.planes_graph.capture()
            except Exception as e:       # an operation the runtime cannot record: the step falls back to the op-by-op pass
                if os.environ.get("CNC_PLANES_GRAPH_STRICT", "0") == "1":      # (the tests: a silent fallback there would
                    raise                                                       # hide a capture regression)
                import warnings
                warnings.warn(f"cnc_amd: capturing the planes' graph failed ({e}); continuing without it")
                self.planes_graph = None

    def _context_pass(self, step: int, fork, params=None):
        """Entropy loss forward + backward on the side stream (from whichever host thread calls it), ordered after the
        event `fork` of the main stream.  `params` = None: the gradient is accumulated into `.grad`; a parameter list:
        it is RETURNED (torch.autograd.grad, None for parameters the entropy loss does not reach) and `.grad` is left
        alone — the data-parallel step keeps the ray-loss gradient there for its all-reduce.
        Returns (bits_per_param, estimated MB, event that marks the end of the pass, gradients or None)."""
        c, side = self.cfg, self.ctx_stream
        torch.cuda.set_device(self.device)
        e = self.field.mlp_base
        side.wait_event(fork)
        with torch.cuda.stream(side), _gradsink.activate(self.sink_ctx):
            # The planes' half as ONE graph launch (cnc_amd._planes_graph): between occupancy refreshes, single-process
            # steps (the data-parallel step wants the gradients returned).  Captured at the first step after a refresh.
            pg, planes, replay = self.planes_graph, None, None
            self._planes_replayed = False
            if self._planes_graph_step(step, params):
                self._ensure_planes_graph(step, params)    # (captured by train_step already when this is the worker thread)
            if self._planes_graph_step(step, params):      # (still: a failed capture switches the graph off)
                # The graph launch itself is ~2 ms of HOST time (the runtime enqueues the ~110 nodes one by one, outside the
                # interpreter lock): from a thread of its own, so that this one goes straight on to the 3-D half.
                replay = self._on_planes_thread(self._replay_planes, side.record_event())
                planes = (None, pg.n_params)               # the bits join the totals below, behind the backward
                self._planes_replayed = True
            elif self._planes_thread_step(step, params):   # an occupancy-refresh step: rebuilt and run op by op over there
                replay = self._on_planes_thread(self._planes_refresh_job, side.record_event(), step)
                planes = (None, sum(t.params.numel() for t in e._encoders()[1:]))
                self._planes_replayed = True
            # the planes' half of the pass on a stream of its own, next to the 3-D half (both directions: autograd runs a
            # node's backward on its forward's stream)
            # (Back-propagating the planes' share of the loss as soon as their forward is enqueued — a second backward call,
            # before the 3-D half is launched — was built and measured: 8.25 -> 8.95 ms.  The first half of the step is bound
            # by the two host threads' launches, and the extra call sits in front of the 3-D forward's.)
            try:
                bits_per_param, mb = self.context.forward_binary_vxl_mixPg_3D2D(
                    e.encoding_xyz, e.encoding_xy, e.encoding_xz, e.encoding_yz, self.estimator.binaries,
                    sample_num=None, step=step, sync_MB=False, stream_2D=self.ctx_stream_2D, planes=planes)
                # issued from the side stream: the root gradient of a backward call is created on the ambient stream and
                # every node waits for it
                root = c.lmbda * bits_per_param * self.loss_scale
                grads = None
                if params is None:
                    root.backward()
                else:
                    grads = torch.autograd.grad(root, params, allow_unused=True)
            except BaseException:
                # the planes' job writes the sink and the graph's static gradients: wait for it (its own error, if any, is
                # secondary) and order this stream after what it enqueued before the error travels on
                if replay is not None:
                    try:
                        replay.result()
                    except BaseException:
                        pass
                    if self.ctx_stream_2D is not None:
                        side.wait_stream(self.ctx_stream_2D)
                raise
            if replay is not None:
                replay.result()                            # the graph launch has been enqueued (and did not fail)
            if self.ctx_stream_2D is not None:
                side.wait_stream(self.ctx_stream_2D)       # the planes' backward kernels: part of what `done` marks
            if planes is not None:                         # the reported totals: + the planes' bits (no gradient here)
                e_ = self.field.mlp_base
                n_all = sum(t.params.numel() for t in e_._encoders())
                bits_per_param = bits_per_param.detach() + pg.step_bits / n_all
                mb = mb + pg.step_bits * (1.0 / 8388608.0)         # / 8 / 1024 / 1024 (a power of two: same value)
            done = side.record_event()
        return bits_per_param, mb, done, grads

    def train_step(self, step: int, want_stats: bool = True) -> Optional[Dict[str, float]]:
        """One optimisation step.  `want_stats=False` leaves mse / psnr / bpp / embed_bits_MB out of the result
        (reading them back is a device->host sync per step; the reference only looks at them every 200 steps,
        train:368-381) — `n_rendering_samples` and `num_rays` are always there."""
        c = self.cfg
        self.field.train(); self.estimator.train(); self.context.train()
        self._planes_replayed = False       # (set by this step's entropy pass if the planes' half runs apart from it)
        # the batch: drawn at the end of the step before (`_prefetch`), while that step's backward kept the GPU busy
        data, self._next_data = self._next_data, None
        if data is None:
            data = self.dataset.fetch()
        elif self._next_ready is not None:          # drawn on the look-ahead stream: join it, hand the tensors over
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self._next_ready)
            self._next_ready = None
            for v in data.values():
                for t in (v if isinstance(v, tuple) else (v,)):
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(cur)
        self.estimator.update_every_n_steps(
            step=step, occ_eval_fn=lambda x: self.field.query_density(x) * c.render_step_size,
            occ_thre=1e-2, n=c.step_update)
        if step % c.step_update == 0 and step > 0:
            # the occupancy refresh has just synchronised the host: the moment to read the fused training forward's fp16
            # range guard (a saturated activation since the last look: the gradient pass leaves that kernel, with a warning)
            self.field.check_range_guard()
        if self.dp and step % c.step_update == 0:
            cdist.broadcast_module_buffers(self.estimator, ["occs", "binaries"])
        ctx_future = None
        for sink in (self.sink_render, self.sink_ctx):      # before either pass forks off: both are ordered after this
            if sink is not None:
                sink.zero()
        warn_before = True
        if self._warn_switch is not None:
            # process-global: held for this step only and put back to what the caller had (private getter when there is one)
            warn_before = bool(getattr(torch._C, "_warn_on_accumulate_grad_stream_mismatch", lambda: True)())
            self._warn_switch(False)
        try:
            if self.ctx_thread and self.ctx_stream is not None and c.lmbda > 0:
                # The ray loss and the entropy loss share nothing but the parameters and the occupancy grid (just
                # updated above).  The entropy pass starts NOW, on the side stream and from a second host thread, next
                # to the whole render pass.  What both passes read through a cache — the sign bit planes of the tables
                # — is made current on the main stream first; gradients are cleared before either backward.  Data
                # parallel: the worker returns its gradient instead of accumulating it (`.grad` = the bucket that is
                # all-reduced), and the window draw that every rank must share is made and broadcast here, on the main
                # thread.
                for enc in self.field.mlp_base._encoders():
                    if enc.ste_binary and enc.bitplane:
                        enc._bit_plane(enc.params)
                if self.bucket is None:
                    self.opt.zero_grad(set_to_none=True)
                    self.opt2.zero_grad(set_to_none=True)
                else:
                    self.bucket.zero()
                    self.bucket.bind(force=True)
                    self._ctx_rand = None
                    self._ctx_rand = self.context.rand_like(self.context.utils_rand)
                    self._ctx_rand.record_stream(self.ctx_stream)      # allocated here, read by the side stream
                if self._pool is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="cnc-context")
                autocast = (torch.is_autocast_enabled(self.device.type), torch.get_autocast_dtype(self.device.type))
                self._ensure_planes_graph(step, None if self.bucket is None else self.bucket.params)
                ctx_future = self._pool.submit(self._context_pass_worker, step,
                                               torch.cuda.current_stream(self.device).record_event(),
                                               None if self.bucket is None else self.bucket.params,
                                               torch.is_grad_enabled(), autocast)
            return self._train_step_tail(step, want_stats, data, ctx_future)
        except BaseException:
            # Never leave the worker running behind an exception: it writes `.grad` and reads the tables.  Wait for it
            # (its own error, if any, is secondary), order the main stream after whatever it enqueued, and drop the
            # window draw that was made for it.
            if ctx_future is not None:
                try:
                    ctx_future.result()
                except BaseException:
                    pass
                torch.cuda.current_stream(self.device).wait_stream(self.ctx_stream)
            if self.bucket is not None:
                self._ctx_rand = None
            raise
        finally:
            if self._warn_switch is not None:
                self._warn_switch(warn_before)

    def _prefetch(self, step: int = -1) -> None:
        """The NEXT step's batch, drawn now: the ray count it depends on has just been set (`update_num_rays`, from this
        step's sample count), this thread would otherwise wait for the entropy pass, and the ~25 small kernels of the draw
        run in the shadow of this step's backward instead of at the head of the next step, in front of everything (the
        entropy pass — the longer of the two — could not start before them: 0.5 ms).  The dataset draws from its own
        generator: the sequence of batches is the one `fetch()` at the top of each step produces.  Single-process steps
        only (data parallel: the ray count of the next step is known at ITS start, `_lagged_sample_count`)."""
        if not self.prefetch or self.dp:
            return
        c, pre = self.cfg, self.ahead_stream
        if pre is None:
            self._next_data = self.dataset.fetch()
            return
        # ... on the look-ahead stream, and with the batch also its MARCH (OccGridEstimator.premarch): rays and occupancy grid
        # are all the march reads, so the next step's two traversal passes and the host round trip for its sample count run
        # here, next to this step's backward, instead of at the head of the next step's render pass — the step's critical
        # chain.  Not in front of a refresh step: it replaces the grid.
        # (ordered after this step's render FORWARD on the main stream — the march that read `binaries`, the coarse occupancy
        # words computed lazily by whichever stream marches first; today a host sync earlier in the step already covers them —
        # not after its backward, which is what this runs next to)
        if self._fwd_enqueued is not None:
            pre.wait_event(self._fwd_enqueued)
        with torch.cuda.stream(pre):
            data = self.dataset.fetch()
            if self.premarch and step >= 0 and (step + 1) % c.step_update != 0:
                from .render import _as_ray_list
                rays, _ = _as_ray_list(data["rays"])
                self.estimator.premarch(rays.origins, rays.viewdirs, near_plane=c.near_plane, render_step_size=c.render_step_size,
                                        stratified=True, cone_angle=c.cone_angle)
            self._next_ready = pre.record_event()
        self._next_data = data

    def _lagged_sample_count(self, num_rays_now: int, n_samples: int) -> None:
        """Data-parallel ray budget without a collective of its own.  The reference resizes the next batch from this
        step's sample count (train:340-344); with N ranks the count that matters is the mean over the ranks, and asking
        for it here would put a blocking all-reduce + host sync between the render forward and its backward on every
        rank.  Instead this rank's count rides in the tail slot of the gradient bucket; the sum comes back with the
        gradients and is read at the NEXT step, right here — after the march has synchronised the host anyway — and
        resizes the batch after that: num_rays[k+1] = num_rays[k-1] * target / mean_count[k-1].  Same fixed point
        (target / samples-per-ray), one step later; every rank computes it from the same all-reduced number."""
        c = self.cfg
        if self._count_pending is not None:
            ev, rays_then = self._count_pending
            ev.synchronize()                        # long since complete: the step before this one
            mean = float(self._count_host[0]) / self.world
            if c.target_sample_batch_size > 0 and mean >= 1.0:
                self.dataset.update_num_rays(int(rays_then * (c.target_sample_batch_size / mean)))
        self._count_pending = None
        self.bucket.tail.fill_(float(n_samples))    # enqueued before the backward; the bucket is zeroed before this

    def _train_step_tail(self, step, want_stats, data, ctx_future):
        c = self.cfg
        rays, pixels, bkgd = data["rays"], data["pixels"], data["color_bkgd"]
        with _gradsink.activate(self.sink_render):
            rgb, acc, depth, n_samples, extra = render_image_with_occgrid(
                self.field, self.estimator, rays, near_plane=c.near_plane, render_step_size=c.render_step_size,
                render_bkgd=bkgd, cone_angle=c.cone_angle, alpha_thre=c.alpha_thre, return_extra=True)
        self._fwd_enqueued = torch.cuda.current_stream(self.device).record_event() if self.device.type == "cuda" else None
        if self.device.type == "cuda":
            # the fused training forward's fp16 range guard, every step and without a wait: the words of the step before have
            # arrived by now (the sampler has synchronised the host since), this step's are sent on their way
            self.field.poll_range_guard()
            self.field.snapshot_range_guard()
        if not self.dp:
            if n_samples == 0:
                if ctx_future is not None:
                    torch.cuda.current_stream(self.device).wait_event(ctx_future.result()[2])
                return None
            if c.target_sample_batch_size > 0:
                self.dataset.update_num_rays(int(len(pixels) * (c.target_sample_batch_size / float(n_samples))))
        # world > 1: no rank leaves the step (the collective below must be entered by everyone; a rank without samples
        # adds a zero ray gradient), and the ray budget follows the all-reduced count of the step before
        mse = F.mse_loss(rgb, pixels)
        bpp, mb = 0.0, 0.0
        e = self.field.mlp_base
        ctx_args = (e.encoding_xyz, e.encoding_xy, e.encoding_xz, e.encoding_yz, self.estimator.binaries)
        main = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None

        def join(result):
            """Order the main stream after the entropy pass; its outputs were allocated on the side stream."""
            bits_per_param, mb_, done, grads = result
            main.wait_event(done)
            for t in (bits_per_param, mb_) + tuple(g for g in (grads or ()) if g is not None):
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)
            return bits_per_param, mb_, grads

        table_pieces = None
        if self.bucket is None:
            if ctx_future is not None:
                (mse * self.loss_scale).backward()
                self._prefetch(step)
                bpp, mb, _ = join(ctx_future.result())
            elif self.ctx_stream is not None and c.lmbda > 0 and mse.requires_grad:
                # The sequential schedule of the same idea (one host thread; the reference's order of random draws):
                #   render backward (main stream: few launches, GPU-heavy)
                #   || context forward + context backward (side stream: ~250 launches, host-bound forward)
                # The side stream forks BEFORE the render backward is enqueued; its host-side syncs (window bounds,
                # nonzero) wait for the side stream only.
                self.opt.zero_grad(set_to_none=True)
                self.opt2.zero_grad(set_to_none=True)
                fork = main.record_event()
                (mse * self.loss_scale).backward()
                bpp, mb, _ = join(self._context_pass(step, fork))
            else:
                loss = mse
                if c.lmbda > 0:
                    with _gradsink.activate(self.sink_ctx):
                        bpp, mb = self.context.forward_binary_vxl_mixPg_3D2D(*ctx_args, sample_num=None, step=step,
                                                                             sync_MB=False)
                    loss = loss + c.lmbda * bpp
                self.opt.zero_grad(set_to_none=True)
                self.opt2.zero_grad(set_to_none=True)
                (loss * self.loss_scale).backward()
            # both passes are joined to this stream: what their kernels added to the sinks goes to `.grad`, once — or, for
            # the tables, straight into their Adam update (`_table_adam`: the pieces are summed there)
            table_pieces = {} if (self.table_adam is not None and self.fused_table_adam) else None
            for sink in (self.sink_render, self.sink_ctx):
                if sink is not None:
                    sink.flush(table_pieces=table_pieces)
            if self._planes_replayed:
                self.planes_graph.flush(table_pieces=table_pieces)       # what autograd returned inside the planes' graph
        else:
            # Data-parallel step.  The ray loss differs per rank, the entropy loss does not (same tables, same
            # window draw on every rank): so only the ray-loss gradient is exchanged, and its all-reduce runs
            # on the communicator's stream WHILE the entropy pass is still under way.  The entropy gradient is equal
            # across ranks only up to the order of its float atomics (~1e-9 relative), so the replicas are
            # compared at every occupancy refresh and re-aligned to rank 0 where they differ (below).
            A, B = self.bucket, self.bucket_ctx
            if ctx_future is None:
                if c.lmbda > 0:
                    with _gradsink.activate(self.sink_ctx):
                        bpp, mb = self.context.forward_binary_vxl_mixPg_3D2D(*ctx_args, sample_num=None, step=step,
                                                                             sync_MB=False)
                A.zero()
                A.bind(force=True)
            self._lagged_sample_count(len(pixels), n_samples)
            if mse.requires_grad:          # a rank whose rays met no sample has nothing to add (its peers do): the
                (mse * self.loss_scale).backward()     # collective below must still be entered by everyone
            if self.sink_render is not None:
                self.sink_render.flush()               # `.grad` = the bucket's views: the encoder scatters join it here
            work = A.allreduce(average=False, async_op=True)
            ctx_grads = None
            if ctx_future is not None:
                bpp, mb, ctx_grads = join(ctx_future.result())
            elif c.lmbda > 0:
                B.zero()
                B.bind(force=True)
                (c.lmbda * bpp * self.loss_scale).backward()
                if self.sink_ctx is not None:
                    self.sink_ctx.flush()              # into `.grad` = B's views
            if work is not None:
                if self.time_comm:      # how long the compute stream stalls for the collective (what was NOT hidden
                    e0 = torch.cuda.Event(enable_timing=True)        # behind the entropy pass)
                    e0.record()
                work.wait()
                if self.time_comm:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self._comm_events.append((e0, e1))
            # the summed sample count goes to the host behind the collective; nobody waits for it before the next step
            self._count_host.copy_(A.tail, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._count_pending = (ev, len(pixels))
            A.grads.div_(self.world)
            if ctx_grads is not None:
                pairs = [(v, g) for v, g in zip(A.views, ctx_grads) if g is not None]
                torch._foreach_add_([v for v, _ in pairs], [g for _, g in pairs])
                if self.sink_ctx is not None:
                    self.sink_ctx.flush()              # `.grad` = A's views (bound before the fork): after the mean
                if self._planes_replayed:
                    self.planes_graph.flush()
            elif c.lmbda > 0:
                A.grads.add_(B.flat)
            A.bind(force=True)
        if self.table_adam is not None:
            if table_pieces is not None:
                self.table_adam.step(table_pieces)         # leaves the tables' `.grad` None: the library's step skips them
            else:
                self.table_adam.steps_done += 1            # the library steps them below
        self.opt.step()
        if c.lmbda > 0:
            self.opt2.step()
        if self.table_adam is not None and table_pieces is not None:
            self.table_adam.mark_planes_current()      # (the two steps above dropped every cache: the tables' planes stand)
        self.sched.step()
        if c.lmbda > 0:
            self.sched2.step()
        if self.bucket is not None and (step + 1) % c.step_update == 0:
            # Only the entropy gradient can differ between replicas (float atomics in another order); the tensors it
            # does not reach — the field's MLPs — stay bit-identical by construction.  Compare bit checksums (one
            # small collective) and broadcast only what differs, instead of all 161 MB every time.
            n, nbytes = cdist.resync_parameters(self.bucket.params)
            self.resync["checks"] += 1
            self.resync["fired"] += 1 if n else 0
            self.resync["tensors"] += n
            self.resync["bytes"] += nbytes
        if not want_stats:
            return {"n_rendering_samples": n_samples, "num_rays": len(pixels)}
        # the step's scalars in one device->host copy
        mse_f, bpp_f, mb_f = torch.stack([mse.detach(), torch.as_tensor(bpp, device=mse.device).detach().float(),
                                          torch.as_tensor(mb, device=mse.device).detach().float()]).tolist()
        return {"mse": mse_f, "psnr": -10.0 * math.log10(max(mse_f, 1e-12)), "bpp": bpp_f,
                "embed_bits_MB": mb_f, "n_rendering_samples": n_samples, "num_rays": len(pixels)}

    def train(self, steps: Optional[int] = None, log=print):
        steps = self.cfg.max_steps if steps is None else steps
        tic = time.time()
        last = None
        for step in range(steps + 1):
            logging = step % self.cfg.log_every == 0 or step == steps
            s = self.train_step(step, want_stats=logging)
            if s is not None and logging:
                last = s
            if log and s is not None and step % self.cfg.log_every == 0 and self.rank == 0:
                log(f"elapsed_time={time.time() - tic:.2f}s | step={step} | psnr={s['psnr']:.2f} | "
                    f"n_rendering_samples={s['n_rendering_samples']} | num_rays={s['num_rays']} | "
                    f"bits_per_param={s['bpp']:.3f} | embed_bits_MB={s['embed_bits_MB']:.3f}")
        return last

    # ------------------------------------------------------------------------------ evaluation
    @torch.no_grad()
    def evaluate(self, n_views: Optional[int] = None) -> float:
        """Mean PSNR over test views; views are sharded over ranks and the mean is all-reduced."""
        c = self.cfg
        self.field.eval(); self.estimator.eval()
        n_views = c.test_views if n_views is None else n_views
        lo, hi = cdist.shard_range(n_views, self.rank, self.world)
        tot = 0.0
        for i in range(lo, hi):
            d = self.dataset.view(i)
            rgb, acc, depth, _ = render_image_with_occgrid_test(
                1024, self.field, self.estimator, d["rays"], near_plane=c.near_plane,
                render_step_size=c.render_step_size, render_bkgd=d["color_bkgd"], cone_angle=c.cone_angle,
                alpha_thre=c.alpha_thre)
            mse = F.mse_loss(rgb, d["pixels"])
            tot += -10.0 * math.log10(max(mse.item(), 1e-12))
        tot = cdist.sum_over_ranks(tot, self.device)
        return tot / max(n_views, 1)

    # ------------------------------------------------------------------------------ codec
    @torch.no_grad()
    def encode(self, prefix: Optional[str] = None):
        os.makedirs(self.cfg.out_dir, exist_ok=True)
        prefix = prefix or os.path.join(self.cfg.out_dir, "b")
        os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
        e = self.field.mlp_base
        self.context.eval()
        return self.context.encode_binary_vxl_mixPg_3D2D(e.encoding_xyz, e.encoding_xy, e.encoding_xz,
                                                         e.encoding_yz, self.estimator.binaries,
                                                         filename_prefix=prefix) + (prefix,)

    @torch.no_grad()
    def decode_into_field(self, Pgs, prefix):
        """Wipe the four tables, decode them from the .b files, install them (train:445-470)."""
        e = self.field.mlp_base
        recs = [torch.ones_like(t.params.data) for t in (e.encoding_xyz, e.encoding_xy, e.encoding_xz, e.encoding_yz)]
        for t in (e.encoding_xyz, e.encoding_xy, e.encoding_xz, e.encoding_yz):
            t.params.zero_()             # under no_grad: bumps _version, which keys the sign-plane cache
            t.invalidate_caches()
        recs = self.context.decode_binary_vxl_mixPg_3D2D(e.encoding_xyz, e.encoding_xy, e.encoding_xz,
                                                         e.encoding_yz, *recs, self.estimator.binaries, Pgs,
                                                         filename_prefix=prefix)
        self.field.update_embedding_params(*recs)

    def sizes_MB(self, coded_MB: float) -> Dict[str, float]:
        ctx_MB = sum(p.numel() * 32 for p in self.context.parameters()) / 8.0 / 1024 / 1024
        _, occ_MB, _ = get_binary_vxl_size(self.estimator.binaries)
        mlp = {k: v for k, v in self.field.state_dict().items() if "encoding" not in k and k != "aabb"}
        mlp_MB, _, _ = quantize_params(mlp, digits=13)
        return {"embeddings": coded_MB, "context_models": ctx_MB, "occupancy_grid": occ_MB, "mlp_13bit": mlp_MB,
                "total": coded_MB + ctx_MB + occ_MB + mlp_MB}

    # ------------------------------------------------------------------------------ container
    def _mlp_state(self):
        return {k: v for k, v in self.field.state_dict().items() if "encoding" not in k and k != "aabb"}

    @torch.no_grad()
    def save_container(self, path: str) -> Dict[str, float]:
        """Encode the tables and write everything a decoder needs into one file; returns sizes in KB."""
        from .container import write_container
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            Pgs, est_MB, coded_MB, prefix = self.encode(os.path.join(td, "b"))
            streams = {f[2:-2]: open(os.path.join(td, f), "rb").read()
                       for f in sorted(os.listdir(td)) if f.endswith(".b")}
        meta = {"Pgs": {k: float(v) for k, v in Pgs.items()}, "n_features": self.cfg.n_features,
                "resolutions_list": list(self.cfg.resolutions_list),
                "resolutions_list_2D": list(self.cfg.resolutions_list_2D),
                "log2_hashmap_size": self.cfg.log2_hashmap_size,
                "log2_hashmap_size_2D": self.cfg.log2_hashmap_size_2D}
        size = write_container(path, meta=meta, table_streams=streams, binaries=self.estimator.binaries,
                               field_mlp=self._mlp_state(), context_state=self.context.state_dict())
        return {"file_KB": size / 1024.0, "embeddings_KB": coded_MB * 1024.0, "estimate_KB": est_MB * 1024.0}

    @torch.no_grad()
    def load_container(self, path: str) -> None:
        """Inverse of save_container on a freshly constructed Trainer with the same config: restores
        occupancy, context models and the (13-bit) MLP, then decodes the four tables."""
        from .container import read_container
        import tempfile
        meta, streams, binaries, mlp, ctx = read_container(path, device=self.device)
        self.estimator.binaries = binaries.to(self.device)
        self.context.load_state_dict(ctx, strict=True)
        self.field.load_state_dict(mlp, strict=False)
        Pgs = {k: torch.tensor(v, device=self.device) for k, v in meta["Pgs"].items()}
        with tempfile.TemporaryDirectory() as td:
            for name, blob in streams.items():
                open(os.path.join(td, f"b_{name}.b"), "wb").write(blob)
            self.decode_into_field(Pgs, os.path.join(td, "b"))
