#!/usr/bin/env python
"""bench.py — CNC hot path on MI355X: march 800x800 rays through an occupancy grid, push every
ray-sample through the 16-level x 2^19 x F8 hash-grid encoder (forward), scatter the gradient back
(backward); at N>1 each rank does that for its own camera and the table gradient is all-reduced
over RCCL.  Prints ONE JSON line (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W]         (N > 1: spawns N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

metric: encoded ray-samples/s (BASELINE.json).  `value` counts every marched sample once per
step and divides by the wall time of the WHOLE step (march + positions + encode fwd + encode bwd
[+ all-reduce]); per-kernel rates are in `kernels`, the dominant kernel's roofline in `roofline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cnc_amd import synthetic  # noqa: E402
from cnc_amd.backends import gridencoder_backend as enc  # noqa: E402
from cnc_amd.nerfacc import grid as ngrid  # noqa: E402
from cnc_amd.backends import nerfacc_cuda as ngrid_cuda  # noqa: E402

F, L, D = 8, 16, 3
LOG2_T = 19
# samples per encoder call (CNC_BENCH_CHUNK overrides).  Larger calls change little: every backward call reads and
# writes the table slabs of the binned levels once (~200 MB) whatever its size, and at 2^22 samples per call the frame
# takes 83.9 instead of 85.2 ms (the backward 1.007 instead of 1.014 ms per 2^20 samples; 2^24 overflows the bins).
CHUNK = int(os.environ.get("CNC_BENCH_CHUNK", 1 << 20))
STEP_SIZE = 5e-3
AABB = (-1.5, -1.5, -1.5, 1.5, 1.5, 1.5)
HBM_PEAK = 8.0e12
BYTES_FWD = 4 * D + L * (2 ** D) * F * 4 + L * F * 4            # 4,620 B / sample
BYTES_BWD = 4 * D + L * F * 4 + 2 * L * (2 ** D) * F * 4        # 8,716 B / sample


class Timed:
    """HIP-event timing of kernel launches on torch's current stream (the stream the C ABI is
    handed), accumulated per kernel name."""

    def __init__(self):
        self.pending = []
        self.acc = {}

    def launch(self, name, units, fn):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.pending.append((name, units, e0, e1))

    def collect(self):
        for name, units, e0, e1 in self.pending:
            a = self.acc.setdefault(name, [0.0, 0, 0])
            a[0] += e0.elapsed_time(e1) * 1e-3
            a[1] += 1
            a[2] += units
        self.pending = []


def reserve_training_streams(dev):
    """Before anything else touches the GPU: the training step's two side streams (cnc_amd.trainer.reserve_streams says
    why the order matters: four hardware queues, dealt out in order of first use)."""
    from cnc_amd.trainer import reserve_streams
    reserve_streams(dev)


def build_workload(dev, rank):
    offs = synthetic.level_offsets(synthetic.RES_16L, LOG2_T, D)
    assert int(offs[-1]) == 6120776
    g = torch.Generator(device="cpu").manual_seed(42)
    # raw parameters as GridEncoder initialises them, U(-1e-4, 1e-4) (ngp.py:221-223); CNC always
    # binarises them (ste_binary=True): forward sees sign(table), backward applies the STE mask
    table = (torch.rand((int(offs[-1]), F), generator=g) * 2 - 1) * 1e-4
    w = dict(
        offsets=torch.as_tensor(offs, device=dev),
        offsets_host=[int(o) for o in offs],
        resolutions=torch.tensor(synthetic.RES_16L, dtype=torch.int32, device=dev),
        table=table.to(dev),
        binaries=synthetic.ball_binaries(128, AABB, 1.0, device=dev),
        aabbs=torch.tensor([AABB], dtype=torch.float32, device=dev),
        aabb0=torch.tensor(AABB, dtype=torch.float32, device=dev),
    )
    # one camera per rank on a circle (weak scaling: every GPU renders its own 800x800 view)
    o, d = synthetic.pinhole_rays(800, 800, 0.6911, 4.0, azimuth=0.7 + 0.785398 * rank, elevation=0.5)
    w["rays_o"], w["rays_d"] = o.to(dev), d.to(dev)
    n = o.shape[0]
    w["near"], w["far"] = torch.zeros(n, device=dev), torch.full((n,), float("inf"), device=dev)
    w["t_order"] = torch.arange(2, device=dev, dtype=torch.int64).expand(n, 2).contiguous()
    from cnc_amd.dist import GradBucket
    w["table_param"] = torch.nn.Parameter(w["table"])
    w["bucket"] = GradBucket([w["table_param"]])
    w["grad_table"] = w["bucket"].views[0]
    w["out"] = torch.empty((L, CHUNK, F), device=dev)
    w["bits"] = torch.empty((int(offs[-1]) * F + 7) // 8, dtype=torch.uint8, device=dev)
    w["clip"] = torch.zeros(1, dtype=torch.int32, device=dev)
    return w


def march_frame(w, box):
    """slab test + count pass + cumsum + fill pass of (ray, t_start, t_end): cnc_march_samples, the form of the march
    the renderer consumes (same t values and order as traverse_grids' interval edges) -> box["s"], box["ex"]."""
    rays_o, rays_d = w["rays_o"], w["rays_d"]
    t_lo, t_hi, hit = ngrid_cuda.ray_aabb_intersect(rays_o, rays_d, w["aabbs"], -float("inf"), float("inf"),
                                                    float("inf"))
    # The fill pass also emits each sample's position o + d (t_start + t_end) / 2 (rgb_sigma_fn,
    # examples/utils.py:251-262) normalised to the unit cube (the radiance field's aabb mapping, ngp.py:518-519)
    # — bit-equal to the separate cnc_sample_positions pass of rounds 1-3, which re-read (ray, t0, t1) per
    # sample — and the ray id as int32 (int64 only at the nerfacc boundary).
    box["ex"] = {"positions": True, "aabb": w["aabb0"], "ray_indices": "int32"}
    box["s"] = ngrid_cuda.march_samples(rays_o, rays_d, None, w["binaries"], w["aabbs"],
                                        torch.cat([t_lo, t_hi], -1), w["t_order"], hit, w["near"], w["far"],
                                        STEP_SIZE, 0.0, extras=box["ex"])


def probe_chunk_of(x):
    """A chunk from the middle of the frame's sample stream (the first one is atypical: 36k grazing rays of ~30
    samples, 2.1 ms against 1.1-1.16)."""
    S = x.shape[0]
    mid = (S // CHUNK // 2) * CHUNK
    return x[mid:mid + min(CHUNK, S)]


def step(w, timed, world):
    """One pass of the hot path over one 800x800 frame.  Returns the number of ray-samples."""
    rays_o, rays_d = w["rays_o"], w["rays_d"]
    n_rays = rays_o.shape[0]
    t0 = time.perf_counter()
    box = {}
    timed.launch("march(ray_aabb+count+cumsum+fill incl. positions)", n_rays, lambda: march_frame(w, box))
    ray_indices, t_starts, t_ends = box["s"][:3]
    S = t_starts.shape[0]
    x = box["ex"]["positions"]

    gt = w["grad_table"]
    finish_exchange(w, timed, world)                   # the previous frame's gradient exchange ran next to this march
    gt.zero_()                                         # zeros_like(embeddings), ngp.py:129
    out = w["out"]
    # STE_binary(params) of the reference (ngp.py:244-245) = one pass that writes the sign bit plane
    timed.launch("pack_sign_bits", w["table"].shape[0], lambda: enc.pack_sign_bits(w["table"], w["bits"], w["clip"]))
    for s in range(0, S, CHUNK):
        n = min(CHUNK, S - s)
        xs = x[s:s + n]
        o = out[:, :n, :] if n == CHUNK else out.view(-1)[: L * n * F].view(L, n, F)
        timed.launch("grid_encode_forward", n, lambda: enc.grid_encode_forward_bits(
            xs, w["bits"], w["offsets"], w["resolutions"], o, n, D, F, L, 128))
        # the encoder output doubles as a resident, non-trivial upstream gradient [L, n, F]
        timed.launch("grid_encode_backward", n, lambda: enc.grid_encode_backward(
            o, xs, w["table"], w["offsets"], w["resolutions"], gt, n, D, F, L, 0, 128, None, None, None, None,
            ste_binary=True, ste_clip_count=w["clip"],
            binned=enc.plan_binned_levels(synthetic.RES_16L, w["offsets_host"], D, F, n)))
    w["probe_chunk"] = probe_chunk_of(x)
    if w.get("exchange_on", world > 1):
        # the only exchange of the path: one flat-bucket all-reduce of the table gradient, asynchronous on the
        # communicator's stream.  What the next frame does before it needs the table or its gradient again — the march
        # (occupancy grid and rays only) — runs next to it; `finish_exchange` is where the compute stream joins.
        w["exchange"] = w["bucket"].allreduce(average=False, async_op=True)
    return S


def finish_exchange(w, timed, world):
    """Wait for the pending all-reduce of the table gradient (if any) and turn the sum into the mean."""
    work = w.pop("exchange", None)
    if work is None:
        return

    def join():
        work.wait()
        w["bucket"].flat.div_(world)
    timed.launch("allreduce(grad_table): wait + mean, after the next march", w["bucket"].flat.numel() * 4, join)


def cpu_baseline(w):
    """Oracle (C port of the reference kernels, OpenMP) on the host cores: march + encode fwd + bwd
    over forty image rows of the same frame (~1e7 ray-samples); bounded so the default run stays short."""
    import oracle
    oracle.build()
    threads = oracle.max_threads()
    n_rays_sub = 40 * 800   # forty image rows through the middle of the ball (~1e7 samples)
    mid = 380 * 800
    o = w["rays_o"][mid:mid + n_rays_sub].cpu().numpy()
    d = w["rays_d"][mid:mid + n_rays_sub].cpu().numpy()
    binaries = w["binaries"].cpu().numpy()
    aabbs = w["aabbs"].cpu().numpy()
    table = w["table"].cpu().numpy()
    offs, res = w["offsets"].cpu().numpy(), w["resolutions"].cpu().numpy()
    t0 = time.perf_counter()
    _, sm, _ = oracle.traverse_grids(o, d, binaries, aabbs, None, None, STEP_SIZE, 0.0)
    pos = o[sm["ray_indices"]] + d[sm["ray_indices"]] * sm["vals"][:, None]
    x = ((pos - aabbs[0, :3]) / (aabbs[0, 3:] - aabbs[0, :3])).astype(np.float32)
    S = x.shape[0]
    y = oracle.grid_encode_forward(x, table, offs, res, threads=threads)
    oracle.grid_encode_backward(y, x, table, offs, res, threads=threads)
    dt = time.perf_counter() - t0
    port = {"value": S / dt, "unit": "ray-samples/s", "cores": threads, "kind": "port",
            "sample": f"{n_rays_sub} rays of the same frame -> {S} samples, march + encode fwd + bwd, all OpenMP x{threads}, {dt:.1f}s"}
    # the "PyTorch-CPU gridencoder fallback" BASELINE.json names: index math + index_select +
    # autograd's index_add_ (oracle/torch_cpu_encoder.py), on 2^16 samples from the middle of the same
    # sample stream, same table; the encoder only (the march above is not repeated)
    from oracle import torch_cpu_encoder as tce
    n_t = min(1 << 16, S)
    xt = torch.from_numpy(np.ascontiguousarray(x[S // 2 - n_t // 2: S // 2 - n_t // 2 + n_t]))
    tt = torch.from_numpy(table)
    gt = torch.from_numpy(np.ascontiguousarray(y[:, :n_t]))
    t0 = time.perf_counter()
    tce.forward_backward(xt, tt, offs, res, gt, ste_binary=True)
    dtt = time.perf_counter() - t0
    torch_fallback = {"value": n_t / dtt, "unit": "ray-samples/s", "cores": torch.get_num_threads(), "kind": "port",
                      "sample": f"{n_t} samples of the same stream, pure-torch CPU GridEncoder fwd+bwd "
                                f"({torch.get_num_threads()} torch threads), {dtt:.1f}s"}
    return port, torch_fallback


def compulsory_backward_bytes(x, offsets_host):
    """Bytes a backward call on these points cannot avoid: the points and every upstream gradient row once, every
    touched table row read and written once.  The touched rows are counted with torch on the GPU (index arithmetic
    of gridencoder.cu:45-87 in int64, uint32 wrap by masking), outside the timed region."""
    n = x.shape[0]
    touched = 0
    primes = (1, 2654435761, 805459861)
    for l, R in enumerate(synthetic.RES_16L):
        rows = offsets_host[l + 1] - offsets_host[l]
        p0 = torch.floor(x * float(R - 2) + 0.5).to(torch.int64)
        dense = R ** 3 <= rows
        keys = []
        for corner in range(8):
            c = p0 + torch.tensor([(corner >> 0) & 1, (corner >> 1) & 1, (corner >> 2) & 1], device=x.device)
            if dense:
                idx = c[:, 0] + c[:, 1] * R + c[:, 2] * R * R
            else:
                idx = ((c[:, 0] * primes[0]) ^ (c[:, 1] * primes[1]) ^ (c[:, 2] * primes[2])) & 0xFFFFFFFF
            keys.append(idx % rows)
        touched += int(torch.unique(torch.cat(keys)).numel())
    return {"points": n * 4 * D, "gradient_rows": n * L * F * 4, "touched_table_rows": touched,
            "table_rmw": 2 * touched * F * 4, "total": n * 4 * D + n * L * F * 4 + 2 * touched * F * 4}


def kernel_source_hashes():
    """git blob hashes of the kernel sources, as tools/summarise_pmc.py stores them next to the measured traffic."""
    import hashlib
    out = {}
    for rel in ("cnc_amd/csrc/grid_encode.hip", "cnc_amd/csrc/grid_encode_merge.hip", "cnc_amd/csrc/grid_encode_binned.hip",
                "cnc_amd/csrc/grid_encode_overlap.hip", "cnc_amd/csrc/encoder_common.hpp", "cnc_amd/csrc/common.hpp",
                "cnc_amd/csrc/march.hip"):
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            data = open(path, "rb").read()
            out[rel] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    return out


def field_entries(dev):
    """The gradient-free radiance field on the reference composition (12 x 3-D T=2^19 + 3 x 4 planes T=2^17, F = 8,
    H = 160) at N = 2^20 samples: ONE fused kernel (positions -> density, positions + directions -> rgb + density,
    cnc_field_fused_forward) against the chain it replaces (four encoder launches into a [N, 256] matrix in HBM,
    hipBLASLt GEMMs, glue kernels).  HIP events on the current stream, median of 7 after 3 warm-up calls."""
    from cnc_amd.field import NGPRadianceField_mygrid_2D3D
    torch.manual_seed(1)
    f = NGPRadianceField_mygrid_2D3D(aabb=list(AABB), n_features_per_level=8, n_neurons=160,
                                     resolutions_list=(18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514),
                                     log2_hashmap_size=19, resolutions_list_2D=(130, 258, 514, 1026),
                                     log2_hashmap_size_2D=17).to(dev)
    with torch.no_grad():
        for e in f.mlp_base._encoders():
            e.params.uniform_(-1, 1)
    n = 1 << 20
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.rand(n, 3, device=dev, generator=g) * 3.0 - 1.5
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)

    def timed_ms(fn):
        ts = []
        for it in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            if it >= 3:
                ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    out = {}
    k0 = f.mlp_base.network[0].in_features
    H, geo = 160, f.geo_feat_dim
    flop_density = 2.0 * n * (k0 * H + H)
    flop_rgb = 2.0 * n * (k0 * H + H * (1 + geo) + (16 + geo) * H + H * H + H * 3)
    with torch.no_grad():
        for name, fused in (("fused", True), ("chain", False)):
            f.fused_field = fused
            md = timed_ms(lambda: f.query_density(x))
            mr = timed_ms(lambda: f(x, d))
            out[f"field_density({name})"] = {"launches": 7, "avg_ms": md, "units_per_s": n / md * 1e3,
                                            "tflops_useful": flop_density / md / 1e9, "timed_region": False}
            out[f"field_rgb+density({name})"] = {"launches": 7, "avg_ms": mr, "units_per_s": n / mr * 1e3,
                                                "tflops_useful": flop_rgb / mr / 1e9, "timed_region": False}
    out["field_note"] = ("N = 2^20 uniform positions in the box, reference composition at F = 8 (K0 = 255, H = 160, geo 79); "
                         f"fused = cnc_field_fused_forward, precision {f.fused_field_precision}, kernel {f.fused_field_kernel} "
                         "(default: two waves per 32-sample tile, features in LDS -> v_mfma_f32_16x16x32_f16, three fp16 "
                         "products per term, i.e. a matrix ceiling of 2.5 PFLOP/s / 3; the exact-fp32 kernel of the range "
                         "guard is enqueued behind every call and returns at once); chain = encoders -> [N,256] in HBM -> "
                         "hipBLASLt -> glue kernels; tflops_useful counts the layers' multiply-adds only")
    out["field_density_speedup"] = out["field_density(chain)"]["avg_ms"] / out["field_density(fused)"]["avg_ms"]
    out["field_rgb_speedup"] = out["field_rgb+density(chain)"]["avg_ms"] / out["field_rgb+density(fused)"]["avg_ms"]
    return out


RES_3D_B = (18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514)       # the reference composition (train:150-155,169)
RES_2D_B = (130, 258, 514, 1026)
N_INPUT_B = 1 << 18


def input_B_bytes(Fb):
    """SURVEY 8(d), Input B: 12 x 3-D levels + 3 planes x 4 levels.  (forward, backward) algorithmic bytes per sample."""
    corners = len(RES_3D_B) * 8 + 3 * len(RES_2D_B) * 4
    levels = len(RES_3D_B) + 3 * len(RES_2D_B)
    pts = 4 * (3 + 3 * 2)
    return pts + corners * Fb * 4 + levels * Fb * 4, pts + levels * Fb * 4 + 2 * corners * Fb * 4


def touched_rows(x, res, offsets_host, dims):
    """Distinct table rows the corners of the points `x` [N, dims] touch, per level (index arithmetic of
    gridencoder.cu:45-87 in int64, uint32 wrap by masking)."""
    primes = (1, 2654435761, 805459861)
    touched = 0
    for l, R in enumerate(res):
        rows = offsets_host[l + 1] - offsets_host[l]
        p0 = torch.floor(x * float(R - 2) + 0.5).to(torch.int64)
        dense = R ** dims <= rows
        keys = []
        for corner in range(1 << dims):
            c = p0 + torch.tensor([(corner >> d) & 1 for d in range(dims)], device=x.device)
            if dense:
                idx = sum(c[:, d] * R ** d for d in range(dims))
            else:
                idx = c[:, 0] * primes[0]
                for d in range(1, dims):
                    idx = idx ^ (c[:, d] * primes[d])
                idx = idx & 0xFFFFFFFF
            keys.append(idx % rows)
        touched += int(torch.unique(torch.cat(keys)).numel())
    return touched


def input_B_entries(dev, x_unit, traffic):
    """SURVEY 8(d) "Synthetic input B": the four encoders of the reference composition (12 x 3-D T = 2^19 + 3 planes x 4
    levels T = 2^17) on N = 2^18 marched samples, F = 8 and F = 2, through the calls the product's training step makes —
    forward: the bit-plane gather of each encoder writing point-major into the base network's [N, ld] input
    (`_FusedFeatures`, cnc_amd/field.py); backward: the four scatters of the gradient of that matrix into a per-step
    gradient sink, routed as `gridencoder_backend.grid_encode_backward` routes a 2^18-sample call (the run-aggregating
    atomic kernel: one binned level x 2^18 samples is below `plan_binned_levels`' work threshold).  HIP events on the
    current stream around the four launches (+ the sinusoid columns in the forward), median of 9 after 3 warm-up calls.
    Outside the timed region."""
    from cnc_amd.field import NGPRadianceField_mygrid_2D3D, _FusedFeatures
    from cnc_amd._gradsink import GradSink
    n = min(N_INPUT_B, x_unit.shape[0])
    x = x_unit[:n].contiguous()
    kernels, roof = {}, {}
    for Fb in (8, 2):
        torch.manual_seed(3)
        f = NGPRadianceField_mygrid_2D3D(aabb=list(AABB), n_features_per_level=Fb, n_neurons=160, resolutions_list=RES_3D_B,
                                         log2_hashmap_size=19, resolutions_list_2D=RES_2D_B, log2_hashmap_size_2D=17).to(dev)
        mb = f.mlp_base
        encs = mb._encoders()
        with torch.no_grad():
            for e in encs:
                e.params.uniform_(-1, 1)
        params = [e.params for e in encs]
        sink = GradSink(params, [])
        ld, cols = mb._layout()

        xs = (x, x[:, :2].contiguous(), x[:, ::2].contiguous(), x[:, 1:].contiguous())
        planes = [e._bit_plane(e.params) for e in encs]
        clips = [c for _, c in planes]
        feat = torch.zeros(n, ld, device=dev)

        def fwd():          # the four launches of `_FusedFeatures.forward` (the sinusoid columns' kernel is no encoder)
            for e_, (bits, _), xi, col in zip(encs, planes, xs, cols):
                enc.grid_encode_forward_bits(xi, bits, e_.offsets_list, e_.resolutions_list, feat, n, e_.num_dim, e_.n_features,
                                             e_.n_levels, 128, None, None, None, out_ld=ld, out_col=col)

        fwd()
        grad = torch.randn_like(feat)

        def bwd():
            _FusedFeatures.scatter(mb, grad, xs, params, clips, n, sink)

        def median_ms(fn):
            ts = []
            for it in range(12):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                if it >= 3:
                    ts.append(e0.elapsed_time(e1))
            return sorted(ts)[len(ts) // 2]

        t_f, t_b = median_ms(fwd), median_ms(bwd)
        b_f, b_b = input_B_bytes(Fb)
        touched = sum(touched_rows(xi, e._res_host, e._off_host, e.num_dim) for xi, e in zip(xs, encs))
        levels = len(RES_3D_B) + 3 * len(RES_2D_B)
        comp_b = n * 36 + n * levels * Fb * 4 + 2 * touched * Fb * 4
        comp_f = n * 36 + n * ld * 4 + touched * Fb // 8               # points, the output rows, the touched rows' sign bits
        for tag, ms, per, comp in (("fwd", t_f, b_f, comp_f), ("bwd", t_b, b_b, comp_b)):
            name = f"input_B_{tag}(F={Fb})"
            kernels[name] = {"launches": 9, "avg_ms": ms, "units_per_s": n / ms * 1e3, "timed_region": False}
            e = traffic.get(f"input_B_{tag}_F{Fb}") if isinstance(traffic.get(f"input_B_{tag}_F{Fb}"), dict) else None
            r = {"avg_launch_ms": ms, "samples_per_launch": n, "bytes_per_sample": per, "bound": "hbm",
                 "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                 "achieved_algorithmic": per * n / ms / 1e6, "algorithmic_over_peak": per * n / (ms * 1e-3) / HBM_PEAK,
                 "compulsory_bytes": comp, "touched_table_rows": touched,
                 "traffic": None, "achieved": None, "frac": None}
            if e:
                r.update({"traffic": e["bytes"], "achieved": e["bytes"] / ms / 1e6, "frac": e["bytes"] / (ms * 1e-3) / HBM_PEAK,
                          "frac_bounds": [e["bytes_min(read requests x 64 B)"] / (ms * 1e-3) / HBM_PEAK,
                                          e["bytes_max(read requests x 128 B)"] / (ms * 1e-3) / HBM_PEAK],
                          "traffic_over_compulsory": e["bytes"] / comp,
                          "atomic_requests": e.get("atomic_requests"),
                          "atomic_unit_busy_frac": None if not e.get("atomic_requests") else e["atomic_requests"] / 21.0e9 / (ms * 1e-3)})
            roof[f"{tag}_F{Fb}"] = r
        del f, sink, grad, feat
    roof["note"] = ("SURVEY 8(d) Input B: 12 x 3-D (T = 2^19) + 3 x 4 planes (T = 2^17), N = 2^18 marched samples of the bench "
                    "frame, the four encoder calls of the training step's render pass (point-major rows of the [N, ld] matrix; "
                    "backward into a gradient sink, run-aggregating atomic kernel); traffic = HBM bytes per call set from "
                    "separate rocprofv3 --pmc passes of tools/bench_input_b.py (profiles/traffic.json, same counter arithmetic "
                    "as `roofline`); frac = traffic / duration / 8 TB/s; the backward's own bound is the memory side's atomic "
                    "request rate (atomic_unit_busy_frac = TCC_ATOMIC requests / 21 G/s / duration)")
    return kernels, roof


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N copies of this script, one rank per GPU,
    with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); rank 0 prints the line."""
    import socket
    import subprocess
    n = args.gpus
    if os.environ.get("CNC_BENCH_ONE_DEVICE") != "1" and torch.cuda.device_count() < n:
        raise SystemExit(f"bench.py --gpus {n}: only {torch.cuda.device_count()} GPU(s) visible")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
        if p.returncode != 0:          # one rank died: the others would wait in a collective forever
            for q in procs:
                if q.poll() is None:
                    q.kill()
    return rc


def guarded_train_step(dev, world=1, rank=0):
    """`train_step_entry`, reported as {"error": ...} instead of ending the run when it raises: it is an extra block of
    the line (timed_region: false), measured after the judged throughput."""
    try:
        return train_step_entry(dev, world, rank)
    except Exception as exc:       # noqa: BLE001 - whatever it is, the headline line still has to be printed
        import traceback
        print(f"[bench rank {rank}] train_step failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
        return {"error": f"{type(exc).__name__}: {exc}", "timed_region": False, "world_size": world}


def train_step_entry(dev, world=1, rank=0):
    """The WHOLE model in the loop (configs[1]/[2]; configs[3] at world > 1): 12x3-D (T=2^19) + 3x4 2-D (T=2^17) levels
    at F=8, sample_num=150000, occupancy marcher, radiance-field MLPs, volume rendering, context models + entropy
    loss, Adam — `cnc_amd.trainer.Trainer.train_step` on the procedural scene, 2^18 target samples per step and rank.
    At world > 1 every rank renders its own rays; the ray-loss gradient (one flat 161 MB bucket) is all-reduced
    asynchronously while the context backward runs (trainer.py).  Reported: the whole job's rendered samples / s
    (all ranks' samples over the slowest rank's time), the all-reduce alone, and how much of it the step still
    waits for."""
    from cnc_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(n_features=8, sample_num=150000, image_size=400, out_dir="/tmp/cnc_bench_bits")
    tr = Trainer(cfg, device=dev)
    warm = int(os.environ.get("CNC_BENCH_TRAIN_WARM", "240"))       # occupancy warm-up, adaptive ray budget settled, surfaces formed: from here on the march hands the
    for step in range(warm):                # sampler 6-8x the samples that survive it, as for the rest of a 30k-step run
        tr.train_step(step, want_stats=False)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
        tr.time_comm = True
    n_steps, samples, rays = int(os.environ.get("CNC_BENCH_TRAIN_STEPS", "60")), 0, 0     # (test hooks: shorter runs)
    t0 = time.perf_counter()
    for step in range(warm, warm + n_steps):
        s = tr.train_step(step, want_stats=False)       # loss scalars are read back on log steps only (as train:368)
        if s is not None:
            samples += s["n_rendering_samples"]
            rays += s["num_rays"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"ms_per_step": dt / n_steps * 1e3, "rendered_samples_per_s": samples / dt, "rays_per_s": rays / dt,
           "samples_per_step": samples / n_steps, "steps": n_steps, "timed_region": False, "world_size": world,
           "config": "full model, F=8, 12x3D(2^19)+3x4x2D(2^17), sample_num=150000, lmbda=2e-3, procedural ball "
                     "scene, target 2^18 samples/step" + ("/rank" if world > 1 else "") +
                     f", steps {warm}-{warm + n_steps - 1} (includes the occupancy refresh every 16 steps)"}
    # occupancy refreshes inside the timed steps and how many of them found the grid unchanged (context.py keeps the vote
    # plan and the planes' slot lists then); refresh-step wall time from a second, short loop over whole refresh periods
    rs0 = dict(tr.context.refresh_stats)
    per_step = []
    for step in range(warm + n_steps, warm + n_steps + 32):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tr.train_step(step, want_stats=False)
        torch.cuda.synchronize()
        per_step.append(((time.perf_counter() - t1) * 1e3, step % cfg.step_update == 0))
    rs1 = dict(tr.context.refresh_stats)
    ref_ms = [m for m, r in per_step if r]
    ord_ms = sorted(m for m, r in per_step if not r)
    out["refresh"] = {"every": cfg.step_update, "refresh_step_ms": ref_ms, "ordinary_step_ms_median_synced": ord_ms[len(ord_ms) // 2],
                      "refreshes_since_start": rs1["refreshes"], "unchanged_grid_kept_plan": rs1["skipped"],
                      "in_the_last_32_steps": {k: rs1[k] - rs0[k] for k in rs1}}
    pg = tr.planes_graph
    out["schedule"] = {"streams": 1 + (tr.ctx_stream is not None) + (tr.ctx_stream_2D is not None),
                       "entropy_pass_thread": bool(tr.ctx_thread),
                       "planes_graph": None if pg is None else {"captures": pg.captures, "replays": pg.replays},
                       "field_forward": "fused kernel, saving form" if tr.field.fused_train else "library GEMMs",
                       "field_weight_grads": "one kernel" if tr.field.fused_wgrad else "split-K library GEMMs",
                       "range_guard_left_fused_forward": not tr.field.fused_train and os.environ.get("CNC_FUSED_TRAIN", "1") == "1",
                       "batch_prefetch": bool(tr.prefetch)}
    if world == 1:
        out["extras"] = trained_model_entries(tr, dev)
    if world > 1:
        out["resync"] = dict(tr.resync)      # how often the replicas' checksums differed at the refresh points
        exposed = sum(a.elapsed_time(b) for a, b in tr._comm_events) / max(len(tr._comm_events), 1)
        tr.time_comm = False
        flat = tr.bucket.flat
        alone = []
        for _ in range(6):                 # the same bucket, nothing else on the GPU
            torch.cuda.synchronize()
            torch.distributed.barrier()
            t1 = time.perf_counter()
            torch.distributed.all_reduce(flat)
            torch.cuda.synchronize()
            alone.append((time.perf_counter() - t1) * 1e3)
        alone_ms = sorted(alone[1:])[len(alone[1:]) // 2]
        tot = torch.tensor([float(samples), float(rays), dt, exposed, alone_ms], dtype=torch.float64, device=dev)
        mx = tot.clone()
        torch.distributed.all_reduce(tot, op=torch.distributed.ReduceOp.SUM)
        torch.distributed.all_reduce(mx, op=torch.distributed.ReduceOp.MAX)
        dt_max = float(mx[2])
        out.update({"ms_per_step": dt_max / n_steps * 1e3, "rendered_samples_per_s": float(tot[0]) / dt_max,
                    "rays_per_s": float(tot[1]) / dt_max, "samples_per_step": float(tot[0]) / n_steps,
                    "allreduce_bytes": int(flat.numel() * flat.element_size()),
                    "allreduce_alone_ms": float(mx[4]), "allreduce_exposed_ms_per_step": float(mx[3]),
                    "allreduce_hidden_frac": max(0.0, 1.0 - float(mx[3]) / max(float(mx[4]), 1e-9)),
                    "allreduce_note": "alone = blocking all-reduce of the same flat gradient bucket on an idle GPU (median "
                                      "of 5, slowest rank); exposed = HIP-event time the compute stream waits for the "
                                      "asynchronous all-reduce after the context backward (mean per step, slowest rank)"})
    return out


def trained_model_entries(tr, dev):
    """On the model `train_step_entry` has just trained for ~300 steps (full size, F = 8): the 800x800 evaluation render
    (`render_image_with_occgrid_test`, examples/utils.py:317-489 — march in bounded rounds, fused gradient-free field,
    fused compositing) and the codec round trip (`encode_binary_vxl_mixPg_3D2D` / `decode_...`, context models on the
    GPU + the CPU range coder).  Outside the timed region; reported with what was measured."""
    from cnc_amd.render import render_image_with_occgrid_test
    from cnc_amd.trainer import SyntheticBallDataset
    c = tr.cfg
    out = {}
    try:
        tr.field.eval(); tr.estimator.eval()
        view = SyntheticBallDataset(800, device=dev, seed=0).view(0)
        args = dict(near_plane=c.near_plane, render_step_size=c.render_step_size, render_bkgd=view["color_bkgd"],
                    cone_angle=c.cone_angle, alpha_thre=c.alpha_thre)
        with torch.no_grad():
            for _ in range(2):
                render_image_with_occgrid_test(1024, tr.field, tr.estimator, view["rays"], **args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                rgb, _, _, n_shaded = render_image_with_occgrid_test(1024, tr.field, tr.estimator, view["rays"], **args)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            mse = torch.nn.functional.mse_loss(rgb, view["pixels"]).item()
        out["eval_render"] = {"ms_per_image": dt * 1e3, "rays": 640000, "shaded_samples": int(n_shaded),
                              "rays_per_s": 640000 / dt, "psnr_vs_procedural_scene": -10.0 * np.log10(max(mse, 1e-12)),
                              "timed_region": False,
                              "note": "800x800 view of the procedural ball after the ~300 training steps above"}
    except Exception as exc:       # noqa: BLE001
        out["eval_render"] = {"error": f"{type(exc).__name__}: {exc}"}
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Pgs, est_MB, coded_MB, prefix = tr.encode()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):     # `update_embedding_params` prints, as the reference's does
            tr.decode_into_field(Pgs, prefix)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out["codec"] = {"encode_s": t1 - t0, "decode_s": t2 - t1, "estimated_MB": float(est_MB), "coded_MB": float(coded_MB),
                        "coded_over_estimate": float(coded_MB) / max(float(est_MB), 1e-12), "timed_region": False,
                        "note": "four tables (12x3-D T=2^19 + 3x4 planes T=2^17, F = 8) -> .b files and back"}
    except Exception as exc:       # noqa: BLE001
        out["codec"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)      # the backward keeps getting faster for ~8 steps (786 vs 732 M/s at 6 vs 2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true")
    ap.add_argument("--no-field", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", str(args.gpus) if args.gpus > 1 else "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (not used by the driver): CNC_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # CNC_BENCH_BACKEND=gloo swaps RCCL for gloo, so the N>1 control flow can be exercised on a
    # single-GPU box (tests/test_gpu_bench_multi.py)
    if os.environ.get("CNC_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
        os.environ["CNC_DIST_ONE_DEVICE"] = "1"       # the Trainer's own device pick follows (cnc_amd.dist)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = "none"
    # CNC_DIST_FORCE=1 (test hook, cnc_amd.dist.forced): a ONE-rank process group on RCCL — the frame's gradient exchange
    # (async all-reduce next to the following march, join, mean) runs through the communicator on a 1-GPU box
    exchange_on = world > 1 or os.environ.get("CNC_DIST_FORCE") == "1"
    if exchange_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("CNC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        me = {"rank": rank, "device": str(dev), "world_size": torch.distributed.get_world_size(),
              "backend": torch.distributed.get_backend()}
        print(f"[bench rank {rank}] {me}", file=sys.stderr, flush=True)
        ranks = [None] * world
        torch.distributed.all_gather_object(ranks, me)
    else:
        ranks = [{"rank": 0, "device": str(dev), "world_size": 1, "backend": backend}]

    if not args.no_train_step:
        reserve_training_streams(dev)
    w = build_workload(dev, rank)
    w["exchange_on"] = exchange_on
    timed = Timed()

    def barrier():
        if exchange_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(w, timed, world)
    finish_exchange(w, timed, world)
    barrier()
    timed.pending, timed.acc = [], {}
    t0 = time.perf_counter()
    samples = 0
    for _ in range(args.steps):
        samples += step(w, timed, world)
    finish_exchange(w, timed, world)          # the last frame's exchange ends inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    timed.collect()

    # outside the timed region (rank 0): (i) the generic fp32-table gather (the `_gridencoder` drop-in
    # entry point, no bit plane); (ii) the backward call's two halves one after the other on ONE stream
    # (coarse levels = k_grid_encode_bwd_merge, finest levels = k_bwd_bin + k_bwd_owner), so each half
    # has its own duration next to its own bound in `roofline.parts`
    extra = Timed()
    if rank == 0:
        n = CHUNK
        xs = torch.rand((n, 3), device=dev)
        for _ in range(5):
            extra.launch("grid_encode_forward_fp32_table(uniform pts)", n, lambda: enc.grid_encode_forward(
                xs, w["table"], w["offsets"], w["resolutions"], w["out"], n, D, F, L, 0, 128, 0.0, None, None, None,
                ste_binary=True))
        if w.get("probe_chunk") is not None and w["probe_chunk"].shape[0] == CHUNK:
            # the forward as the product's field issues it: point-major rows straight into a 256-wide MLP input
            # (out_ld / out_col, cnc_amd/field.py) instead of the [L, N, F] array of the drop-in entry
            feat = torch.empty((CHUNK, 256), device=dev)
            for it in range(7):
                if it == 2:
                    torch.cuda.synchronize()
                    extra.pending = [p_ for p_ in extra.pending if not p_[0].startswith("grid_encode_forward(point-major")]
                extra.launch("grid_encode_forward(point-major into a [N,256] MLP input)", CHUNK, lambda: enc.grid_encode_forward_bits(
                    w["probe_chunk"], w["bits"], w["offsets"], w["resolutions"], feat, CHUNK, D, F, L, 128, None, None, None,
                    out_ld=256, out_col=0))
            del feat
        ngrid.traverse_grids(w["rays_o"], w["rays_d"], w["binaries"], w["aabbs"], step_size=STEP_SIZE, cone_angle=0.0)
        for _ in range(3):      # the `nerfacc.csrc` drop-in entry (intervals + samples, 27 B / sample), for the record
            extra.launch("traverse_grids drop-in (ray_aabb+traverse x2+cumsum)", w["rays_o"].shape[0],
                         lambda: ngrid.traverse_grids(w["rays_o"], w["rays_d"], w["binaries"], w["aabbs"],
                                                      step_size=STEP_SIZE, cone_angle=0.0))
        nb_plan = enc.plan_binned_levels(synthetic.RES_16L, w["offsets_host"], D, F, n)
        if nb_plan is not None and w.get("probe_chunk") is not None and w["probe_chunk"].shape[0] == CHUNK:
            xs_r, o_r = w["probe_chunk"], w["out"]
            nr = xs_r.shape[0]
            enc.grid_encode_forward_bits(xs_r, w["bits"], w["offsets"], w["resolutions"], o_r, nr, D, F, L, 128)
            nb = nb_plan[0]
            k = L - nb
            gt = w["grad_table"]
            for it in range(7):
                if it == 2:               # two untimed rounds first (scratch allocation, caches)
                    torch.cuda.synchronize()
                    extra.pending = [p_ for p_ in extra.pending if not p_[0].startswith("bwd_")]
                extra.launch("bwd_coarse_levels(k_grid_encode_bwd_merge), alone", nr, lambda: enc.grid_encode_backward(
                    o_r[:k], xs_r, w["table"], w["offsets"][:k + 1], w["resolutions"][:k], gt, nr, D, F, k, 0, 128,
                    None, None, None, None, ste_binary=True, ste_clip_count=w["clip"], binned=None,
                    interleave_levels=True))
                extra.launch("bwd_finest_levels(k_bwd_bin+k_bwd_owner), alone", nr, lambda: enc.grid_encode_backward(
                    o_r[k:], xs_r, w["table"], w["offsets"][k:], w["resolutions"][k:], gt, nr, D, F, nb, 0, 128,
                    None, None, None, None, ste_binary=True, ste_clip_count=w["clip"], binned=(nb, nb_plan[1]),
                    overlap_streams=False))
        torch.cuda.synchronize()
        extra.collect()

    tot = torch.tensor([float(samples), elapsed], dtype=torch.float64, device=dev)
    if exchange_on:
        s = tot[0:1].clone()
        e = tot[1:2].clone()
        torch.distributed.all_reduce(s, op=torch.distributed.ReduceOp.SUM)
        torch.distributed.all_reduce(e, op=torch.distributed.ReduceOp.MAX)
        samples_all, elapsed_max = s.item(), e.item()
    else:
        samples_all, elapsed_max = float(samples), elapsed

    # the DP training step (configs[3]) runs on EVERY rank; rank 0 reports the aggregate.  It comes after the
    # headline numbers are in and must not take the line down with it.
    ts_multi = None
    if world > 1 and not args.no_train_step:
        ts_multi = guarded_train_step(dev, world, rank)

    if rank == 0:
        kernels = {}
        for name, (secs, launches, units) in timed.acc.items():
            kernels[name] = {"launches": launches, "avg_ms": secs / launches * 1e3,
                             "units_per_s": units / secs}
        for name, (secs, launches, units) in extra.acc.items():
            kernels[name] = {"launches": launches, "avg_ms": secs / launches * 1e3,
                             "units_per_s": units / secs, "timed_region": False}
        kb = timed.acc["grid_encode_backward"]
        kf = timed.acc["grid_encode_forward"]
        # dominant kernel = the one with the most accumulated time in the timed region
        dom_name, bytes_per, k = (("grid_encode_backward", BYTES_BWD, kb) if kb[0] >= kf[0]
                                  else ("grid_encode_forward", BYTES_FWD, kf))
        launch_s = k[0] / k[1]
        algorithmic = bytes_per * k[2] / k[0]   # SURVEY §8(d) bytes / s, averaged over launches
        # Measured HBM bytes per launch (PMC FETCH_SIZE/WRITE_SIZE, profiles/traffic.json).  The kernels move
        # FEWER bytes than the algorithmic count (bit-plane gather, cells merged before the table is touched),
        # so the algorithmic rate can exceed the HBM peak and is no fraction of anything; `frac` is the
        # measured traffic over the measured duration against the HBM peak.
        tj = {}
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            tj = json.load(open(tp))
        # the byte counts are tied to the kernels they were measured on: a kernel source that changed since
        # (git blob hash) marks them stale
        measured_on = tj.get("_sources", {})
        now = kernel_source_hashes()
        stale = sorted(k for k in now if measured_on.get(k) != now[k]) if measured_on else ["(no source hashes in traffic.json)"]
        req_peak = float(tj.get("_fabric_request_rate_peak_G_per_s", 50.0)) * 1e9

        def entry_of(name):
            e = tj.get(name)
            return e if isinstance(e, dict) else None

        def rates(e, dur):
            """calibrated bytes / s, both bounds, and the L2 -> fabric request rate of one entry over `dur` seconds"""
            if not e:
                return {}
            req = (e.get("read_requests") or 0) + (e.get("write_requests") or 0)
            return {"traffic": e["bytes"], "achieved": e["bytes"] / dur / 1e9, "frac": e["bytes"] / dur / HBM_PEAK,
                    "traffic_bounds": [e["bytes_min(read requests x 64 B)"], e["bytes_max(read requests x 128 B)"]],
                    "frac_bounds": [e["bytes_min(read requests x 64 B)"] / dur / HBM_PEAK,
                                    e["bytes_max(read requests x 128 B)"] / dur / HBM_PEAK],
                    "fabric_requests": {"read": e.get("read_requests"), "write": e.get("write_requests"),
                                        "achieved_G_per_s": req / dur / 1e9, "peak_G_per_s": req_peak / 1e9,
                                        "frac": req / dur / req_peak},
                    # row atomics and plain requests are served by ONE memory-side unit in which an atomic takes the place
                    # of ~2.7 reads (tools/mix_probe.hip, profiles/r05_backward_closure.md): its busy time for this entry
                    "memory_side_unit": {"atomic_requests": e.get("atomic_requests") or 0.0, "atomic_peak_G_per_s": 21.0,
                                         "plain_requests": req - (e.get("atomic_requests") or 0.0),
                                         "plain_peak_G_per_s": [req_peak / 1e9, 57.0],
                                         "busy_frac": [((e.get("atomic_requests") or 0.0) / 21.0e9
                                                        + (req - (e.get("atomic_requests") or 0.0)) / pk) / dur
                                                       for pk in (57.0e9, req_peak)]}}

        dom = entry_of(dom_name)
        rr = rates(dom, launch_s)
        nb = (enc.plan_binned_levels(synthetic.RES_16L, w["offsets_host"], D, F, CHUNK) or (0, 0))[0]
        desc = {"grid_encode_backward": f"one cnc_grid_encode_backward_overlapped call = k_grid_encode_bwd_merge ({L - nb} coarse "
                                        f"levels, runs merged across rays, atomics) + k_bwd_bin + k_bwd_owner ({nb} finest levels, LDS "
                                        "accumulation); avg_launch_ms is the whole call between two events on the "
                                        "caller's stream (the library joins its side streams before returning)",
                "grid_encode_forward": "k_grid_encode_fwd_bits"}
        comp = None
        if dom_name == "grid_encode_backward" and w.get("probe_chunk") is not None and w["probe_chunk"].shape[0] == CHUNK:
            comp = compulsory_backward_bytes(w["probe_chunk"], w["offsets_host"])
        roofline = {"kernel": dom_name, "kernel_parts": desc[dom_name], "bound": "hbm",
                    "achieved": rr.get("achieved"), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": rr.get("frac"), "traffic": rr.get("traffic"),
                    "frac_basis": "HBM bytes per launch from rocprofv3 --pmc passes (profiles/traffic.json), corrected per access "
                                  "class as calibrated in profiles/r03_counter_calibration.md — a streamed read request = 128 B "
                                  "(FETCH_SIZE tallies 64), a gathered 32-byte row = one 64-byte request, WRITE_SIZE as reported — "
                                  "over avg_launch_ms over 8 TB/s.  frac_bounds: every read request 64 B / 128 B.",
                    "frac_bounds": rr.get("frac_bounds"), "traffic_bounds": rr.get("traffic_bounds"),
                    "traffic_stale": bool(stale), "traffic_stale_sources": stale,
                    "fabric_requests": rr.get("fabric_requests"),
                    "memory_side_unit": rr.get("memory_side_unit"),
                    "fabric_requests_note": "the finest levels sit on the L2 -> fabric REQUEST rate, not on bytes: a gather of one "
                                            "32-byte gradient row costs a whole request (tools/fetch_calib.hip: 51.5 G gathers/s "
                                            "from 2 GiB, a 16 B/lane stream 46.9 G requests/s = 6.0 TB/s)",
                    "achieved_algorithmic": algorithmic / 1e9,
                    "algorithmic_over_peak": algorithmic / HBM_PEAK,
                    "traffic_over_algorithmic": None if not rr else rr["traffic"] / (bytes_per * k[2] / k[1]),
                    "compulsory_bytes": None if comp is None else comp["total"], "compulsory_parts": comp,
                    "traffic_over_compulsory": None if (comp is None or not rr) else rr["traffic"] / comp["total"],
                    "bytes_per_sample": bytes_per, "samples_per_launch": k[2] / k[1],
                    "avg_launch_ms": launch_s * 1e3}
        # each half of the backward call against the bound it actually sits on
        parts = {}
        pc = extra.acc.get("bwd_coarse_levels(k_grid_encode_bwd_merge), alone")
        pf = extra.acc.get("bwd_finest_levels(k_bwd_bin+k_bwd_owner), alone")
        if pc:
            dur = pc[0] / pc[1]
            e = entry_of("k_grid_encode_bwd_merge")
            req = None if not e else e.get("atomic_requests")
            parts["k_grid_encode_bwd_merge"] = {
                "avg_ms": dur * 1e3, "bound": "memory-side fp32 atomic requests (tools/atomic_probe.hip: 21 G requests/s)",
                "note": "since the cells are merged across rays the kernel is vector-issue-bound, not request-bound: without "
                        "its atomics it runs 10 % faster, without the MFMA accumulation 35 % (docs/engineering_log.md §4.2b, second pass)",
                "atomic_requests_per_launch": req, "achieved_G_requests_per_s": None if not req else req / dur / 1e9,
                "peak_G_requests_per_s": 21.0, "frac": None if not req else req / dur / 21e9}
            parts["k_grid_encode_bwd_merge"].update({"hbm_" + k_: v for k_, v in rates(e, dur).items() if k_ in ("traffic", "frac")})
        if pf:
            dur = pf[0] / pf[1]
            e = entry_of("k_bwd_bin+k_bwd_owner")
            parts["k_bwd_bin+k_bwd_owner"] = dict({"avg_ms": dur * 1e3, "bound": "L2 -> fabric request rate (gathers); hbm bytes beside it",
                                                   "peak_GBps": HBM_PEAK / 1e9}, **rates(e, dur))
        roofline["parts"] = parts
        ef = entry_of("grid_encode_forward")
        fl = kf[0] / kf[1]
        other = dict({"kernel": "grid_encode_forward", "bound": "L2->L1 line rate of the byte gathers (docs/engineering_log.md §4.3); HBM only for "
                                                               "the 512 B/sample output stream", "unit": "GB/s",
                      "achieved_algorithmic": BYTES_FWD * kf[2] / kf[0] / 1e9, "bytes_per_sample": BYTES_FWD,
                      "avg_launch_ms": fl * 1e3}, **rates(ef, fl))
        pm = extra.acc.get("grid_encode_forward(point-major into a [N,256] MLP input)")
        if pm:
            other["point_major_variant"] = {
                "avg_launch_ms": pm[0] / pm[1] * 1e3, "samples_per_s": pm[2] / pm[0],
                "note": "the same gather writing point-major rows into a [N, 256] feature matrix (out_ld=256), the form the "
                        "product's field uses (no [L,N,F] array, no permute / cat afterwards); the 512 B/sample still have to "
                        "reach HBM once — only a gather fused into the MLP kernel would remove them",
                "output_bytes_per_sample": L * F * 4}
        out = {
            "metric": "ray-samples/s/GPU (16Lx2^19xF8 grid)", "value": samples_all / elapsed_max,
            "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: 800x800 rays/GPU marched (step 5e-3) through a 128^3 ball "
                                   "occupancy, every sample encoded fwd (sign bit plane) + bwd (STE mask) on the "
                                   "binarised 16Lx2^19xF8 hash grid"
                                   + (", grad table all-reduced (RCCL)" if world > 1 else ""),
                       "rays_per_gpu": 640000, "samples_per_step_rank0": samples // args.steps,
                       "chunk": CHUNK, "table_rows": 6120776, "n_features": F, "levels": L},
            "ranks": ranks,
            "roofline": roofline, "roofline_forward": other, "kernels": kernels,
        }
        if not args.no_field:
            try:
                fe = field_entries(dev)
                out["field"] = {k: fe.pop(k) for k in ("field_note", "field_density_speedup", "field_rgb_speedup")}
                kernels.update(fe)
            except Exception as exc:       # noqa: BLE001 - an extra block must not take the line down
                import traceback
                print(f"[bench rank {rank}] field entries failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
                out["field"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_field and w.get("probe_chunk") is not None:
            try:
                kb_, rb_ = input_B_entries(dev, w["probe_chunk"], tj)
                kernels.update(kb_)
                out["roofline_input_B"] = rb_
            except Exception as exc:       # noqa: BLE001 - an extra block must not take the line down
                import traceback
                print(f"[bench rank {rank}] input B failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
                out["roofline_input_B"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_train_step:
            ts = guarded_train_step(dev) if world == 1 else ts_multi
            if isinstance(ts, dict) and "extras" in ts:
                out.update(ts.pop("extras"))
            out["train_step"] = ts
            if "error" not in ts:
                kernels["train_step(full model: march+field+render+context+adam)"] = {
                    "launches": ts["steps"], "avg_ms": ts["ms_per_step"], "units_per_s": ts["rendered_samples_per_s"],
                    "timed_region": False}
        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 at N=1 only
            out["cpu_baseline"], out["cpu_baseline_torch"] = cpu_baseline(w)
        print(json.dumps(out), flush=True)
    if exchange_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
