"""Context models + entropy estimate + codec for the binarised hash-grid embeddings.

Host-side mirror of examples/utils_bpp_acc.py of the reference:
    `_cnt_np_embed` :27-75, `encoder`/`decoder` :77-110, `align_and_pack` :113-139,
    `CNC_context_models` :193-999 (tables :260-402, training pass :533-706, encode :709-865,
    decode :867-999), `Bernoulli_entropy` :1002-1013.
Same class / method names, argument meaning, return values and file naming, so the reference's
drivers can call it unchanged.  The three passes share their per-level arithmetic here (the
reference repeats it three times); every kernel call goes through the HIP mirrors
(`_gridencoder`, `pack_and_align`), the entropy coder through libcnc_codec.so.

Differences that do not change results:
  * device is taken from the constructor (`device=`) instead of hard-coded 'cuda' module
    globals (utils_bpp_acc.py:19-20), so the tables can be built and inspected on CPU;
  * `align_and_pack(...).sum(dim=1)` — a padded [N, M, F] tensor (M up to 288) that is
    immediately reduced — keeps the reference's dataflow for now (bit-identical means);
  * random draws go through `self.rand_like` / `self.randperm` hooks (default torch) so tests
    can replay the reference's CPU random stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import contextlib

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as nnf
from torch.autograd import Function

if os.environ.get("CNC_PROFILE_RANGES", "0") == "1":      # named ranges for tools/aten_by_range.py; two dispatcher
    from torch.profiler import record_function as _range   # calls each, ~40 per training pass, so off by default
else:
    def _range(name):
        return _NO_RANGE
    _NO_RANGE = contextlib.nullcontext()

from .backends import context_backend as _ctxk
from .backends import gridencoder_backend as _backend
from .backends import pack_and_align
from .gridencoder import STE_binary, STE_multistep
from .mlp import Linear

from ._codec import lib as _codec_lib  # noqa: E402  (libcnc_codec.so, include/cnc_codec.h)


def get_grid_index(hashmap_size, resolution, pos_grid):
    """Row index of integer grid vertices (reference twin of the kernel's hash:
    examples/utils.py:492-511).  Dense x + y*R + z*R^2 when the level fits, else xor of
    coordinate*prime; int64 arithmetic, which equals the kernel's uint32 wrap-around for the
    power-of-two table sizes hashed levels have."""
    D = pos_grid.shape[-1]
    pos = pos_grid.to(torch.long)
    if resolution ** D <= hashmap_size:
        idx = torch.zeros(pos.shape[:-1], dtype=torch.long, device=pos.device)
        stride = 1
        for d in range(D):
            idx = idx + pos[..., d] * stride
            stride *= resolution
    else:
        primes = (1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737)
        idx = pos[..., 0] * primes[0]
        for d in range(1, D):
            idx = idx ^ (pos[..., d] * primes[d])
    return idx % hashmap_size


class _cnt_np_embed(Function):
    """Fraction of +1 votes of the finest 3-D level projected on a plane (utils_bpp_acc.py:27-75)."""

    @staticmethod
    def forward(ctx, inputs, embeddings, resolution, hashmap_size, axis):
        axis_id = ("xy", "xz", "yz").index(axis)
        N = inputs.shape[0]
        n_features = embeddings.shape[-1]
        assert inputs.shape[-1] == 3
        inputs = inputs.to(torch.int16).contiguous()
        embeddings = embeddings.contiguous()
        scale = resolution - 2
        pn_embed = torch.zeros([scale, scale, n_features, 2], device=inputs.device)
        _backend.cnt_np_embed(inputs, embeddings, pn_embed, N, resolution, n_features, hashmap_size, axis_id)
        pn_embed_sum = torch.sum(pn_embed, dim=-1, keepdim=True) + 1e-6
        ctx.save_for_backward(inputs, embeddings, pn_embed_sum)
        ctx.dims = (N, resolution, n_features, hashmap_size, axis_id)
        return pn_embed / pn_embed_sum

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, pn_embed_sum = ctx.saved_tensors
        N, resolution, n_features, hashmap_size, axis_id = ctx.dims
        grad_embeddings = torch.zeros_like(embeddings)
        _backend.cnt_np_embed_backward(inputs, embeddings, pn_embed_sum, grad.contiguous(),
                                       grad_embeddings, N, resolution, n_features, hashmap_size, axis_id)
        return None, grad_embeddings, None, None, None


_LEVEL_CONSTS = {}


def _level_consts(off, n_features, device):
    """(offsets, level lengths in rows, entries per level) as device tensors, made once per table layout."""
    key = (off, n_features, str(device))
    if key not in _LEVEL_CONSTS:
        lengths = [off[i + 1] - off[i] for i in range(len(off) - 1)]
        _LEVEL_CONSTS[key] = (torch.tensor(off, dtype=torch.long, device=device),
                              torch.tensor(lengths, dtype=torch.long, device=device),
                              torch.tensor([l * n_features for l in lengths], dtype=torch.float32, device=device))
    return _LEVEL_CONSTS[key]


class _LevelSums(Function):
    """Sum of every level's entries of a [rows, F] table: out[l] = sum(table[off[l]:off[l+1]])."""

    @staticmethod
    def forward(ctx, table, off):
        cs = torch.cumsum(table.sum(dim=1, dtype=torch.float64), 0)
        idx = _level_consts(off, table.shape[1], table.device)[0]
        cs0 = torch.cat([cs.new_zeros(1), cs])
        ctx.off, ctx.shape = off, table.shape
        return (cs0[idx[1:]] - cs0[idx[:-1]]).to(table.dtype)

    @staticmethod
    def backward(ctx, g):
        off = ctx.off
        lengths = _level_consts(off, ctx.shape[1], g.device)[1]
        rows = torch.repeat_interleave(g, lengths, output_size=off[-1] - off[0])
        grad = rows.new_zeros(ctx.shape[0])
        grad[off[0]:off[-1]] = rows
        return grad[:, None].expand(ctx.shape), None


class _cnt_np_embed_planned(Function):
    """`_cnt_np_embed` from a `VotePlan` (vertex list pre-sorted by pixel / by table row): same counts
    bit for bit, no atomics (cnc_amd/csrc/cnt_votes.hip)."""

    @staticmethod
    def forward(ctx, plan, embeddings, axis):
        axis_id = ("xy", "xz", "yz").index(axis)
        n_features = embeddings.shape[-1]
        embeddings = embeddings.contiguous()
        scale = plan.resolution - 2
        pn_embed = torch.empty([scale, scale, n_features, 2], device=embeddings.device)
        _backend.cnt_np_embed_planned(plan, embeddings, pn_embed, n_features, axis_id)
        pn_embed_sum = torch.sum(pn_embed, dim=-1, keepdim=True) + 1e-6
        ctx.save_for_backward(embeddings, pn_embed_sum)
        ctx.plan, ctx.axis_id = plan, axis_id
        return pn_embed / pn_embed_sum

    @staticmethod
    def backward(ctx, grad):
        embeddings, pn_embed_sum = ctx.saved_tensors
        grad_embeddings = torch.zeros_like(embeddings)
        g_over_sum = (torch.reciprocal(pn_embed_sum) * grad).contiguous()   # gv * grad, gridencoder.cu:1035-1040
        _backend.cnt_np_embed_planned_backward(ctx.plan, embeddings, g_over_sum, grad_embeddings,
                                               embeddings.shape[-1], ctx.axis_id)
        return None, grad_embeddings, None


class _cnt_np_embed_planned3(Function):
    """The three projections (xy, xz, yz) of `_cnt_np_embed_planned` as ONE autograd node: the forward shares the
    packed votes, the backward is one pass over the table rows that WRITES the table gradient (no zero-fill, no
    accumulation of three table-sized tensors)."""

    @staticmethod
    def forward(ctx, plan, embeddings):
        n_features = embeddings.shape[-1]
        embeddings = embeddings.contiguous()
        scale = plan.resolution - 2
        outs, sums = [], []
        for axis_id in range(3):
            pn_embed = torch.empty([scale, scale, n_features, 2], device=embeddings.device)
            _backend.cnt_np_embed_planned(plan, embeddings, pn_embed, n_features, axis_id)
            pn_sum = torch.sum(pn_embed, dim=-1, keepdim=True) + 1e-6
            outs.append(pn_embed / pn_sum)
            sums.append(pn_sum)
        ctx.save_for_backward(embeddings, *sums)
        ctx.plan = plan
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_xy, g_xz, g_yz):
        embeddings, *sums = ctx.saved_tensors
        gs = [(torch.reciprocal(sm) * g).contiguous() for sm, g in zip(sums, (g_xy, g_xz, g_yz))]   # gridencoder.cu:1035-1040
        grad_embeddings = torch.empty_like(embeddings)
        _backend.cnt_np_embed_planned_backward3(ctx.plan, embeddings, gs, grad_embeddings, embeddings.shape[-1])
        return None, grad_embeddings


class _vote_tables3(Function):
    """`_cnt_np_embed_planned3` + the fraction / select / pad chain of `_ring_of_zeros` as one node: returns the three
    [R * R, F] tables the dimension-wise context encodes from.  One kernel per plane each way for what were five and
    seven library launches (a keepdim sum over an axis of length 2, add, div, select, pad and their backward)."""

    @staticmethod
    def forward(ctx, plan, embeddings):
        from . import _lib
        F_ = embeddings.shape[-1]
        embeddings = embeddings.contiguous()
        S = plan.resolution - 2
        R = plan.resolution
        L = _lib.lib()
        tables, sums = [], []
        for axis_id in range(3):
            cnt = torch.empty([S, S, F_, 2], device=embeddings.device)
            _backend.cnt_np_embed_planned(plan, embeddings, cnt, F_, axis_id)
            table = torch.empty([R * R, F_], device=embeddings.device)
            sm = torch.empty([S, S, F_], device=embeddings.device)
            _lib.check(L.cnc_vote_fraction_table(cnt.data_ptr(), S, F_, table.data_ptr(), sm.data_ptr(),
                                                 _lib.stream(embeddings.device)), "vote_fraction_table")
            tables.append(table)
            sums.append(sm)
        ctx.save_for_backward(embeddings, *sums)
        ctx.plan = plan
        return tuple(tables)

    @staticmethod
    def backward(ctx, g_xy, g_xz, g_yz):
        from . import _lib
        embeddings, *sums = ctx.saved_tensors
        F_ = embeddings.shape[-1]
        S = ctx.plan.resolution - 2
        L = _lib.lib()
        gs = []
        for sm, g in zip(sums, (g_xy, g_xz, g_yz)):
            out = torch.empty([S, S, F_, 2], device=embeddings.device)
            if g is None:
                out.zero_()
            else:
                _lib.check(L.cnc_vote_fraction_table_backward(g.contiguous().data_ptr(), sm.data_ptr(), S, F_,
                                                              out.data_ptr(), _lib.stream(embeddings.device)),
                           "vote_fraction_table_backward")
            gs.append(out)
        grad_embeddings = torch.empty_like(embeddings)
        _backend.cnt_np_embed_planned_backward3(ctx.plan, embeddings, gs, grad_embeddings, F_)
        return None, grad_embeddings


def _encode_host(x, p, file_name):
    """x, p: contiguous float32 HOST tensors.  Writes the .b file, returns its size in bits."""
    n = x.numel()
    L = _codec_lib()
    cap = int(L.cnc_rc_bound(n))
    buf = np.empty(cap, dtype=np.uint8)
    nbytes = L.cnc_rc_encode_pm1(p.data_ptr(), x.data_ptr(), n, buf.ctypes.data, cap)
    if nbytes < 0:
        raise RuntimeError("range coder: output buffer too small")
    with open(file_name, "wb") as fout:
        fout.write(buf[:nbytes].tobytes())
    return int(nbytes) * 8


def _decode_host(p, file_name):
    with open(file_name, "rb") as fin:
        stream = np.frombuffer(fin.read(), dtype=np.uint8)
    out = torch.empty(p.numel(), dtype=torch.float32)
    _codec_lib().cnc_rc_decode_pm1(p.data_ptr(), p.numel(), stream.ctypes.data, stream.shape[0], out.data_ptr())
    return out


def encoder(x, p, file_name):
    """Code x in {-1,+1} with P(x=+1)=p into `file_name` (.b); returns the size in bits
    (utils_bpp_acc.py:77-93)."""
    assert file_name[-2:] == ".b"
    x = x.detach().to(torch.float32).cpu().contiguous().view(-1)
    p = p.detach().to(torch.float32).cpu().contiguous().view(-1)
    return _encode_host(x, p, file_name)


def decoder(p, file_name):
    """Inverse of `encoder`: returns float32 ±1 on p's device (utils_bpp_acc.py:95-110)."""
    assert file_name[-2:] == ".b"
    dvc = p.device
    return _decode_host(p.detach().to(torch.float32).cpu().contiguous().view(-1), file_name).to(dvc)


class CoderPool:
    """The reference codes its 33 streams one after the other, each behind two blocking `.cpu()` copies
    (utils_bpp_acc.py:77-110, call sites :722-804).  The streams are independent files, so here they are
    coded CONCURRENTLY: `encode` / `decode` start a device->pinned-host copy on a side stream and hand the
    rest (wait for the copy, run the C coder — ctypes drops the GIL —, touch the file) to a worker thread;
    the GPU goes on computing the next stream's probabilities meanwhile.  Same bytes as `encoder` /
    `decoder` (tests/test_gpu_context.py).  At most `max_in_flight` streams hold pinned buffers at once."""

    def __init__(self, workers=None, max_in_flight=8):
        import concurrent.futures as cf
        import threading
        self.pool = cf.ThreadPoolExecutor(max_workers=workers or min(8, os.cpu_count() or 1))
        self.slots = threading.Semaphore(max_in_flight)
        self.copy_stream = {}

    def _to_host(self, t):
        """float32 flat copy of `t` in pinned memory + the event that says it has arrived."""
        t = t.detach().to(torch.float32).contiguous().view(-1)
        if not t.is_cuda:
            return t, None
        side = self.copy_stream.setdefault(t.device, torch.cuda.Stream(device=t.device))
        side.wait_stream(torch.cuda.current_stream(t.device))
        host = torch.empty(t.numel(), dtype=torch.float32, pin_memory=True)
        with torch.cuda.stream(side):
            host.copy_(t, non_blocking=True)
            t.record_stream(side)
            ev = torch.cuda.Event()
            ev.record(side)
        return host, ev

    def encode(self, x, p, file_name):
        """Future of the stream's size in bits."""
        assert file_name[-2:] == ".b"
        self.slots.acquire()
        (xh, ex), (ph, ep) = self._to_host(x), self._to_host(p)

        def job():
            try:
                for ev in (ex, ep):
                    if ev is not None:
                        ev.synchronize()
                return _encode_host(xh, ph, file_name)
            finally:
                self.slots.release()
        return self.pool.submit(job)

    def decode(self, p, file_name):
        """Future of the decoded float32 +-1 HOST tensor (pinned when p lives on a GPU)."""
        assert file_name[-2:] == ".b"
        self.slots.acquire()
        ph, ep = self._to_host(p)

        def job():
            try:
                if ep is not None:
                    ep.synchronize()
                return _decode_host(ph, file_name)
            finally:
                self.slots.release()
        return self.pool.submit(job)

    def shutdown(self):
        self.pool.shutdown(wait=True)


class align_and_pack(Function):
    """Ragged [T, F] rows grouped by `unique_cnt` -> padded [N, max(cnt), F] (utils_bpp_acc.py:113-139)."""

    @staticmethod
    def forward(ctx, voxel_features, unique_cnt, V, dim=3):
        voxel_features = voxel_features.contiguous()
        unique_cnt = unique_cnt.contiguous()
        cumsum = torch.cat([torch.zeros(1, dtype=unique_cnt.dtype, device=unique_cnt.device),
                            torch.cumsum(unique_cnt, dim=0)])
        N = unique_cnt.numel()
        M = int(unique_cnt.max()) if N else 0          # host sync, as in the reference (:121)
        F = voxel_features.shape[-1]
        T = int(cumsum[-1])
        packed = pack_and_align.align_and_pack_forward(voxel_features, unique_cnt, cumsum, N, M, F, 0.0, dim)
        ctx.save_for_backward(voxel_features, unique_cnt, cumsum)
        ctx.dims = (N, M, F, T, dim)
        return packed

    @staticmethod
    def backward(ctx, dL_packed):
        voxel_features, unique_cnt, cumsum = ctx.saved_tensors
        N, M, F, T, dim = ctx.dims
        d_feat = pack_and_align.align_and_pack_backward(dL_packed.contiguous(), voxel_features,
                                                        unique_cnt, cumsum, N, M, F, T, dim)
        return d_feat, None, None, None


class _segment_reduce(Function):
    """Per-slot reduction of ragged rows without the padded tensor (extension, see
    cnc_segment_weighted_sum): mode 0 sum(w*v), 1 sum(w*v)/sum(w), 2 mean.  Gradient w.r.t. values."""

    @staticmethod
    def forward(ctx, values, cumsum, weights, mode, order=None):
        values = values.contiguous()
        if weights is not None:
            weights = weights.contiguous()
        out = pack_and_align.segment_weighted_sum(values, weights, cumsum, mode, order)
        ctx.save_for_backward(cumsum, weights, order)
        ctx.mode, ctx.T = mode, values.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        cumsum, weights, order = ctx.saved_tensors
        wsum = None
        if ctx.mode == 1:
            wsum = pack_and_align.segment_weighted_sum(weights.unsqueeze(-1).contiguous(), None, cumsum, 0)
        return _ctxk.segment_backward(g, cumsum, weights, wsum, ctx.T, ctx.mode, order), None, None, None, None


def _cum(cnt):
    return torch.cat([torch.zeros(1, dtype=torch.long, device=cnt.device), torch.cumsum(cnt, dim=0)])


def my_meshgrid3D(start=(0, 0, 0), end=(1000, 1000, 1000), dtype=torch.int32, device="cuda"):
    """[lx, ly, lz, 3] integer lattice (utils_bpp_acc.py:142-161)."""
    if isinstance(start, int):
        start, end = (start,) * 3, (end,) * 3
    axes = [torch.arange(s, e, device=device, dtype=dtype) for s, e in zip(start, end)]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1)


class Bernoulli_entropy(nn.Module):
    """bits of x in {-1,+1} under P(+1)=clamp(p, 1e-6, 1-1e-6) (utils_bpp_acc.py:1002-1013)."""

    def forward(self, x, p):
        p = torch.clamp(p, min=1e-6, max=1 - 1e-6)
        pos_mask = (1 + x) / 2.0
        neg_mask = (1 - x) / 2.0
        return -torch.log2(p) * pos_mask + -torch.log2(1 - p) * neg_mask


def _level_table_path(cache_dir, R, rows, D):
    return os.path.join(cache_dir, f"ctx_level_R{R}_T{rows}_D{D}.pt")


def _load_level_table(cache_dir, R, rows, D, device):
    """(pos_sorted int16 [R^D, D], unique_value, unique_cnt) of one level from the cache, or None."""
    if not cache_dir:
        return None
    path = _level_table_path(cache_dir, R, rows, D)
    if not os.path.exists(path):
        return None
    try:
        blob = torch.load(path, map_location="cpu")
        if blob.get("key") != (R, rows, D) or blob["pos_sorted"].shape != (R ** D, D):
            return None
        return tuple(blob[k].to(device) for k in ("pos_sorted", "unique_value", "unique_cnt"))
    except Exception:          # a truncated or foreign file is a cache miss, never an error
        return None


def _save_level_table(cache_dir, R, rows, D, pos_sorted, unique_value, unique_cnt):
    if not cache_dir:
        return
    os.makedirs(cache_dir, exist_ok=True)
    path = _level_table_path(cache_dir, R, rows, D)
    tmp = f"{path}.{os.getpid()}.tmp"
    torch.save({"key": (R, rows, D), "pos_sorted": pos_sorted.cpu(), "unique_value": unique_value.cpu(),
                "unique_cnt": unique_cnt.cpu()}, tmp)
    os.replace(tmp, path)      # atomic: concurrent ranks never see a half-written table


def _zero_order_bits(pos_num, neg_num, Pg):
    """pos * -log2(Pg) + neg * -log2(1 - Pg) (utils_bpp_acc.py:478-485) with the logarithms' arguments
    floored at 1e-9 — below the smallest non-zero frequency any table can have (1 / 2^22), so every value the
    reference can produce is unchanged.  A level whose entries all share one sign (Pg = 0 or 1; it happens to
    coded levels of small tables within a few Adam steps) then costs 0 bits and has a zero gradient, where
    the literal expression is 0 * inf: NaN in the forward, and — for all levels computed in one vector, as
    `level_stats` does — a NaN gradient into the table even when that level's value is never used."""
    return pos_num * (-torch.log2(Pg.clamp_min(1e-9))) + neg_num * (-torch.log2((1 - Pg).clamp_min(1e-9)))


class CNC_context_models(nn.Module):
    MAX_POINTS_NUM_TO_OOM = 20000000

    def __init__(self, num_dim=3,
                 resolutions_list=(16, 22, 31, 42, 57, 78, 106, 146, 199, 273, 374, 512),
                 resolutions_list_2D=(128, 256, 512, 1024), log2_hashmap_size=19,
                 log2_hashmap_size_2D=21, n_features=4, sample_num=20000, max_context_layer_num=3,
                 ste_binary=False, ste_multistep=False, add_noise=False, Q=100, quantize_epoch=1000,
                 Pg_level=-1, Pg_level_2D=-1, Rb=128, step_update=16, skip_levels_3D=(0, 1, 2, 3),
                 skip_levels_2D=(0,), use_dimension_wise=True, use_overlap_area_pool=True,
                 device="cuda", dimension_wise_resolution=514, fused_segments=True, planned_votes=True,
                 table_cache_dir=None, fused_heads=True):
        super().__init__()
        # prediction heads + Bernoulli rate as fused HIP kernels (cnc_amd/csrc/ctx_head.hip); False = the
        # reference's op-by-op dataflow (cat -> Linear(s) -> clamp / log2 / masks / sum)
        self.fused_heads = fused_heads
        # on-disk cache of the per-level sorted vertex tables: a pure function of
        # (resolution, table rows, num_dim), so levels are shared between configurations.  Opt-in (argument
        # or CNC_CTX_TABLE_CACHE): on an MI355X rebuilding all 12 levels takes 1.2 s, about what reading
        # 1.3 GB from disk costs; on slower devices the cache pays.
        table_cache_dir = table_cache_dir or os.environ.get("CNC_CTX_TABLE_CACHE") or None
        dev = torch.device(device)
        self.dev = dev
        # hash fusion as one segmented-reduction kernel instead of pack -> multiply -> sum over a
        # padded [slots, max collisions, F] tensor (False = the reference's dataflow)
        self.fused_segments = fused_segments
        # count the dimension-wise votes from a per-refresh sorted plan instead of with atomics
        self.planned_votes = planned_votes
        self.use_overlap_area_pool = use_overlap_area_pool
        self.use_dimension_wise = use_dimension_wise
        self.rand_like = torch.rand_like
        self.randperm = torch.randperm

        resolutions_list = torch.tensor(list(resolutions_list), device=dev)
        resolutions_list_2D = torch.tensor(list(resolutions_list_2D), device=dev)
        n_levels, n_levels_2D = resolutions_list.numel(), resolutions_list_2D.numel()
        self.num_dim = num_dim
        self.resolutions_list = resolutions_list
        self.resolutions_list_2D = resolutions_list_2D
        self.scales_list = (resolutions_list - 2).unsqueeze(-1)
        self.scales_list_2D = (resolutions_list_2D - 2).unsqueeze(-1)
        self.log2_hashmap_size = log2_hashmap_size
        self.log2_hashmap_size_2D = log2_hashmap_size_2D
        self.n_features = n_features
        self.n_levels, self.n_levels_2D = n_levels, n_levels_2D
        self.ste_binary, self.ste_multistep, self.add_noise = ste_binary, ste_multistep, add_noise
        self.Q = Q
        self.quantize_epoch = quantize_epoch
        self.quantize_epoch_cnt = 0
        if Pg_level == -1 or Pg_level >= n_levels:
            Pg_level = n_levels
        self.Pg_level = max(Pg_level, 1)
        if Pg_level_2D == -1 or Pg_level_2D >= n_levels_2D:
            Pg_level_2D = n_levels_2D
        self.Pg_level_2D = max(Pg_level_2D, 1)
        Pg_level = self.Pg_level
        self.skip_levels_3D = skip_levels_3D
        self.skip_levels_2D = skip_levels_2D
        self.sample_num = sample_num
        self._noncoded_3D = None
        self.max_context_layer_num = max_context_layer_num

        def offsets(res, T, D):
            o = [0]
            for R in res.tolist():
                o.append(o[-1] + int(np.ceil(min(T, R ** D) / 8) * 8))
            return torch.tensor(o, dtype=torch.long, device=dev)

        max_params = 2 ** log2_hashmap_size
        self.offsets_list = offsets(resolutions_list, max_params, num_dim)
        self.offsets_list_2D = offsets(resolutions_list_2D, 2 ** log2_hashmap_size_2D, 2)
        offsets_list = self.offsets_list
        # host copies: slicing a table with device scalars costs a device->host sync per slice
        self._off3_host = [int(v) for v in self.offsets_list.tolist()]
        self._res3_host = [int(v) for v in resolutions_list.tolist()]
        self._off2_host = [int(v) for v in self.offsets_list_2D.tolist()]
        self._res2_host = [int(v) for v in self.resolutions_list_2D.tolist()]

        # finest level that is still stored densely (utils_bpp_acc.py:288-293)
        self.n_levels_thresh = n_levels - 1
        self.resolution_thresh = resolutions_list[-1]
        for i in range(n_levels - 1):
            if resolutions_list[i] ** num_dim <= max_params and resolutions_list[i + 1] ** num_dim > max_params:
                self.n_levels_thresh = i + 1
                self.resolution_thresh = resolutions_list[i] + 0.0

        # per level: every grid vertex sorted by the hash slot it lands in (:296-335)
        unique_value_list, cumsum_list, count_list, pos_sorted_list = [], [], [], []
        zero = torch.zeros(1, dtype=torch.long, device=dev)
        for i in reversed(range(Pg_level)):
            R = int(resolutions_list[i].item())
            rows_i = int(offsets_list[i + 1] - offsets_list[i])
            cached = _load_level_table(table_cache_dir, R, rows_i, num_dim, dev)
            if cached is not None:
                pos_sorted, unique_value, unique_cnt = cached
            else:
                pos_grid = my_meshgrid3D(0, R, device=dev).view(-1, 3)
                indexes = get_grid_index(rows_i, R, pos_grid)
                # stable: vertices of one slot stay in lattice order on every device (the reference's
                # unstable CUDA sort leaves that order unspecified; nothing downstream depends on it)
                indexes_sorted, order = torch.sort(indexes, descending=False, dim=0, stable=True)
                pos_sorted = torch.index_select(pos_grid.to(torch.int16), dim=0, index=order)
                unique_value, unique_cnt = torch.unique_consecutive(indexes_sorted, return_counts=True)   # sorted already: no second sort
                del pos_grid, indexes, indexes_sorted, order
                _save_level_table(table_cache_dir, R, rows_i, num_dim, pos_sorted, unique_value, unique_cnt)
            if R <= self.resolution_thresh:
                # dense levels: random slot order so a sampled window is spatially spread (:311-315)
                shuffle = self.randperm(unique_value.nelement()).to(dev)
                unique_value, pos_sorted, unique_cnt = unique_value[shuffle], pos_sorted[shuffle], unique_cnt[shuffle]
            unique_value_list.insert(0, unique_value.to(torch.long))
            cumsum_list.insert(0, torch.cat([zero, torch.cumsum(unique_cnt, dim=0)]).to(torch.long))
            count_list.insert(0, unique_cnt)
            pos_sorted_list.insert(0, pos_sorted)
        self.unique_value_list = unique_value_list
        self.pos_grid_sorted_list = pos_sorted_list

        lens = [c.numel() for c in cumsum_list]
        self.unique_count_cumsum_list = torch.zeros([Pg_level, max(lens)], dtype=torch.long, device=dev)
        self.unique_count_list = torch.zeros([Pg_level, max(lens)], dtype=torch.long, device=dev)
        for i in range(Pg_level):
            self.unique_count_cumsum_list[i, :lens[i]] = cumsum_list[i]
            self.unique_count_list[i, :lens[i] - 1] = count_list[i]

        # how many hash slots per level enter the per-step entropy estimate (:350-366)
        hp = torch.tensor([lens[i] - 1 for i in range(Pg_level)], device=dev)
        self.hashparams_num_levels = hp
        self.sample_num_levels = self._sample_allocation(sample_num)
        self.ttl_hashparams_num_levels = int(hp.sum().item())
        coded = [n for n in range(n_levels) if n not in skip_levels_3D and n < Pg_level]
        self.ttl_hashparams_num_valid_levels = int(sum(hp[n].item() for n in coded))
        self.ttl_sample_num = int(self.sample_num_levels.sum().item())
        self.ttl_sample_num_valid_levels = int(sum(self.sample_num_levels[n].item() for n in coded))

        self.utils_rand = torch.rand(size=[Pg_level]).to(dev)
        self.utils_nlevel_idx = torch.arange(Pg_level, device=dev)
        self.utils_points_per_param_levels = [
            ((resolutions_list[i] ** num_dim) / hp[i]).item() for i in range(Pg_level)]

        ar = torch.arange(0, Rb, device=dev, dtype=torch.int32)
        self.binary_vxl_2D_idx = torch.stack(torch.meshgrid(ar, ar, indexing="ij"), dim=-1)

        self.context_model_3D = nn.Sequential(
            Linear(n_features * max_context_layer_num + 1, 32), nn.LeakyReLU(),
            Linear(32, 32), nn.LeakyReLU(),
            Linear(32, n_features),
        ).to(dev)
        heads = []
        for n in range(1, self.Pg_level_2D):
            ctx_layers = min(n, max_context_layer_num)
            heads.append(nn.Sequential(
                Linear(n_features * (ctx_layers + int(use_dimension_wise)) + 1, n_features)))
        self.context_model_2D = nn.Sequential(*heads).to(dev)
        self.entropy_model = Bernoulli_entropy()

        self.binary_vxl_len = Rb
        self.dimension_wise_resolution = dimension_wise_resolution
        self.init_binary_vxl_coords(scale=dimension_wise_resolution - 2)
        self.step_update = step_update
        # a refresh whose occupancy grid equals the one the structures were built from keeps them (forward_..._3D2D)
        self.skip_unchanged_refresh = os.environ.get("CNC_SKIP_UNCHANGED_REFRESH", "1") == "1"
        self.refresh_stats = {"refreshes": 0, "skipped": 0}
        self.structures_version = 0          # + 1 whenever a refresh rebuilds the vote plan / the planes' vertex lists
        self._occ_built_from, self._occ_built_how = None, None
        self.idx_coords2_tmp = None
        self.vote_plan = None
        self.batched_inputs_list = None
        self._rows_2D_cat = [None, None, None]
        self._plane_cat = [None, None, None]
        self._noncoded_2D = None
        # (extension) the coded levels of a plane in one pass instead of one pass per level — same numbers
        self.plane_batched = os.environ.get("CNC_CTX_PLANE_BATCH", "1") == "1"

    # ------------------------------------------------------------------------------- helpers
    def _sample_allocation(self, sample_num):
        hp = self.hashparams_num_levels
        alloc = torch.round(hp * (sample_num / hp.sum())).to(torch.long)
        return hp if alloc[-1] > hp[-1] else alloc

    def _query(self, points, binary_vxl, resolution=None, resolution_list=None):
        N = points.shape[0]
        # (the kernels write every entry: no zero fill of the ~10^7-entry buffers the reference allocates with zeros)
        new = torch.empty if points.is_cuda else torch.zeros
        mask = new([N], dtype=torch.int16, device=points.device)
        overlap = new([N], dtype=torch.int32, device=points.device)
        vxl = binary_vxl.squeeze(0).contiguous()
        if resolution_list is None:
            pack_and_align.query_mask_3D(points.contiguous(), vxl, mask, overlap, int(resolution), N)
        else:
            pack_and_align.query_mask_3D_qlist(points.contiguous(), vxl, mask, overlap,
                                               resolution_list.contiguous(), N)
        return mask.to(torch.bool), overlap

    def query_binary_vxl(self, points_n_orig, binary_vxl, n, mem_save=False, verbose=False,
                         return_overlap_area=False):
        mask, overlap = self._query(points_n_orig, binary_vxl, resolution=self.resolutions_list[n])
        return (mask, overlap) if return_overlap_area else mask

    def query_binary_vxl_qlist(self, points_n_orig_list, binary_vxl, n_list, return_overlap_area=False):
        mask, overlap = self._query(points_n_orig_list, binary_vxl,
                                    resolution_list=self.resolutions_list[n_list])
        return (mask, overlap) if return_overlap_area else mask

    def fetch_2D_batches(self, binary_vxl_2D, n):
        """All vertices of 2-D level n inside (or one ring around) occupied projected cells:
        their hash rows and normalised positions (utils_bpp_acc.py:431-456)."""
        # level constants from the host copies and the occupied cells of a plane listed once for its levels: the
        # refresh pass runs this 9 times inside a training step, each device scalar read here was a stall
        Rb = binary_vxl_2D.shape[-1]
        R = self._res2_host[n]
        T, rem = divmod(R - 2, Rb)
        assert rem == 0
        # one entry per plane, keyed on the storage a (fresh) view of the plane points at — `binary_2D[k]` is a new tensor
        # object at every call, the nonzero below is a host sync
        key = (binary_vxl_2D.data_ptr(), binary_vxl_2D._version, tuple(binary_vxl_2D.shape), tuple(binary_vxl_2D.stride()))
        cache = self.__dict__.setdefault("_occ_cells_2D", {})
        cached = cache.get(key)
        if cached is None:
            if len(cache) >= 8:
                cache.clear()
            cached = (binary_vxl_2D, self.binary_vxl_2D_idx.view(-1, 2)[binary_vxl_2D.reshape(-1) == 1])
            cache[key] = cached
        cells = cached[1]
        if self.fused_segments and cells.is_cuda:
            return _ctxk.plane_ring_vertices(cells.contiguous(), T, R, self._off2_host[n + 1] - self._off2_host[n])
        occ = cells.view(-1, 1, 1, 2) * T
        ar = torch.arange(0, T + 2, device=self.dev)
        ring = torch.stack(torch.meshgrid(ar, ar, indexing="ij"), dim=-1).view(1, T + 2, T + 2, 2)
        points_n_orig = (occ + ring).to(torch.long)
        indexes_2D = get_grid_index(self._off2_host[n + 1] - self._off2_host[n], R, points_n_orig.view(-1, 2))
        points_n = (points_n_orig - 0.5) / float(R - 2)
        return indexes_2D, points_n.view(-1, 2)

    def get_STE_params(self, Encoding, mode="ste_binary"):
        assert mode in ("ste_binary", "ste_multistep", "add_noise")
        params = Encoding.params
        if mode == "ste_binary":
            return STE_binary.apply(params)
        if mode == "ste_multistep":
            return STE_multistep.apply(params, self.Q)
        return params + (self.rand_like(params) - 0.5) * (1 / self.Q)

    def level_stats(self, params_q, off_host):
        """`get_BiRF_wentropy_leveln` for every level of one table at once: (Pg [L], bits [L]).
        One pass over the table instead of L sliced reductions and ~10 scalar kernels per level; the
        level sums are differences of a float64 running sum of the row sums (exact for +-1 tables)."""
        off = tuple(off_host)
        if self.fused_heads and params_q.is_cuda and len(off) <= 33:
            return _ctxk.level_stats(params_q, off)
        sums = _LevelSums.apply(params_q, off)
        ttl = _level_consts(off, params_q.shape[1], params_q.device)[2]
        pos_num, neg_num = (ttl + sums) / 2.0, (ttl - sums) / 2.0
        Pg = pos_num / ttl
        return Pg, _zero_order_bits(pos_num, neg_num, Pg)

    def get_BiRF_wentropy_leveln(self, params_q, n, offsets_list=None):
        """Level frequency Pg_n = #(+1)/numel and the zero-order bit count (utils_bpp_acc.py:472-486)."""
        off = self._off2_host if offsets_list is self.offsets_list_2D else \
            (self._off3_host if offsets_list is None or offsets_list is self.offsets_list else offsets_list)
        level = params_q[int(off[n]):int(off[n + 1])]
        ttl = level.numel()
        s = torch.sum(level)
        pos_num, neg_num = (ttl + s) / 2.0, (ttl - s) / 2.0
        Pg_n = pos_num / ttl
        return Pg_n, _zero_order_bits(pos_num, neg_num, Pg_n), ttl

    def init_binary_vxl_coords(self, scale=512):
        t = scale // self.binary_vxl_len
        resolution = scale + 2
        self._idx_coord_t = t          # the cell-to-vertex factor the candidate lattice below was built for
        self.idx_coord_base = my_meshgrid3D(-1, t + 1, device=self.dev).unsqueeze(0)
        self.idx_coord_temp = my_meshgrid3D(0, self.binary_vxl_len, device=self.dev).view(-1, 1, 1, 1, 3)
        self.pn_frac_offsets_list = torch.tensor([0, resolution * resolution], device=self.dev, dtype=torch.int32)
        self.pn_frac_resolutions_list = torch.tensor([resolution], device=self.dev, dtype=torch.int32)

    def get_idx_coords2(self, binary_vxl, resolution=None):
        """Unique finest-level vertices inside / one ring around occupied cells (utils_bpp_acc.py:498-512)."""
        resolution = self.dimension_wise_resolution if resolution is None else resolution
        t = (resolution - 2) // self.binary_vxl_len
        # The candidate lattice (`idx_coord_base`) is the one `init_binary_vxl_coords(scale)` built: the reference's
        # expression only means something for resolution == scale + 2 (utils_bpp_acc.py:397-398,498-512), and the two
        # branches below agree only then — so anything else is refused instead of answered two different ways.
        if t != getattr(self, "_idx_coord_t", t):
            raise ValueError(f"get_idx_coords2: resolution {resolution} does not match the lattice of "
                             f"init_binary_vxl_coords (factor {self._idx_coord_t}, not {t})")
        occ = binary_vxl.squeeze(0)
        if self.fused_segments and occ.is_cuda and occ.dim() == 3 and t >= 1:
            # The same sorted set without materialising (t + 2)^3 candidates per occupied cell and sorting ~10^7 of
            # them (7 ms of the 21 ms a refresh step costs): cell c covers the vertices c t .. c t + t + 1 of an axis,
            # so vertex u is covered iff one of the FINE cells u - 2, u - 1, u (fine cell i = coarse cell i // t) is
            # occupied — three shifted ORs per axis on a bool volume, then the coordinates of the set entries
            # (row-major order = ascending x R^2 + y R + z, what torch.unique returned).
            m = occ.to(torch.bool)
            for axis in range(3):
                up = m.repeat_interleave(t, dim=axis)
                n = up.shape[axis]
                shape = list(up.shape)
                shape[axis] = n + 2
                out = torch.zeros(shape, dtype=torch.bool, device=up.device)
                for sft in range(3):
                    out.narrow(axis, sft, n).logical_or_(up)
                m = out
            return torch.nonzero(m).to(self.idx_coord_base.dtype)      # int32, as the candidate-lattice branch returns
        sel = self.idx_coord_temp[occ.reshape(-1)]
        coords = (sel * t + self.idx_coord_base).view(-1, 3) + 1
        lin = coords[..., 0] * resolution * resolution + coords[..., 1] * resolution + coords[..., 2]
        lin = torch.unique(lin, dim=0)
        return torch.stack([lin // (resolution * resolution), (lin // resolution) % resolution,
                            lin % resolution], dim=-1)

    def get_pn_embed_frac(self, embeddings_3D_q, idx_coords2, resolution=None, axis="xy", plan=None):
        resolution = self.dimension_wise_resolution if resolution is None else resolution
        if plan is not None:
            frac = _cnt_np_embed_planned.apply(plan, embeddings_3D_q, axis)
        else:
            frac = _cnt_np_embed.apply(idx_coords2, embeddings_3D_q, resolution, 2 ** self.log2_hashmap_size, axis)
        return self._ring_of_zeros(frac)

    def get_pn_embed_frac_planes(self, embeddings_3D_q, plan):
        """`get_pn_embed_frac` for the xy, xz and yz planes at once (one autograd node, see `_cnt_np_embed_planned3`)."""
        if self.fused_heads and embeddings_3D_q.is_cuda and embeddings_3D_q.dtype == torch.float32:
            return list(_vote_tables3.apply(plan, embeddings_3D_q))
        out = []
        for frac in _cnt_np_embed_planned3.apply(plan, embeddings_3D_q):
            out.append(self._ring_of_zeros(frac))
        return out

    def _ring_of_zeros(self, frac):
        """[R-2, R-2, F, 2] vote fractions -> the [R * R, F] table of the +1 fractions with a ring of zero pixels.
        The reference permutes to [1, F, R-2, R-2], pads the two spatial axes and permutes back
        (utils_bpp_acc.py:520-526): the same values as padding the first two axes in place."""
        return nnf.pad(frac[..., 0], pad=[0, 0, 1, 1, 1, 1]).reshape(-1, self.n_features)

    @staticmethod
    def _project(binary_vxl, axis):
        return torch.any(binary_vxl.squeeze(0), dim={"xy": 2, "xz": 1, "yz": 0}[axis])

    # ---- shared per-level arithmetic -------------------------------------------------------
    def _mean_2D(self, Encoding_2D, n, points_n, Pg_n, binary_vxl_2D, pn_embed_frac, order,
                 unique_cnt, outspace_params=None, detach_pn=False, cum=None):
        """P(+1) per distinct hash slot of 2-D level n: context = lower levels (+ dimension-wise
        3-D vote fraction) -> linear head -> mean over the vertices colliding in the slot."""
        ctx_layers = min(n, self.max_context_layer_num)
        context = Encoding_2D(points_n, n - ctx_layers, n, outspace_params=outspace_params,
                              binary_vxl=binary_vxl_2D, PV=0)
        context_pn = None
        if self.use_dimension_wise:
            context_pn = Encoding_2D.forward_given_params(points_n, self.pn_frac_offsets_list,
                                                          self.pn_frac_resolutions_list,
                                                          pn_embed_frac, binary_vxl_2D)
            if detach_pn:
                context_pn = context_pn.detach()
        if self.fused_heads and context.is_cuda:
            # [context | context_pn | Pg] -> Linear, read in place by one kernel (no cat, no repeated Pg column)
            mean = _ctxk.context_mlp(self.context_model_2D[n - 1], context, context_pn, Pg_n)
        else:
            Pg_col = Pg_n.reshape(1, 1).repeat(context.shape[0], 1)
            parts = [context, Pg_col] if context_pn is None else [context, context_pn, Pg_col]
            mean = self.context_model_2D[n - 1](torch.cat(parts, dim=-1))
        if self.fused_segments:
            # the sort by hash slot (index_select(mean, 0, order)) is folded into the reduction
            return _segment_reduce.apply(mean, _cum(unique_cnt) if cum is None else cum, None, 2, order)
        mean = torch.index_select(mean, dim=0, index=order)
        mean = align_and_pack.apply(mean, unique_cnt, 0.0, 2)
        return torch.sum(mean, dim=1) / unique_cnt.unsqueeze(-1)

    def _fuse_3D(self, mean_pts, mask_cnt, overlap_w):
        """Hash fusion: combine the per-vertex predictions of one slot (overlap-weighted or plain mean)."""
        if self.fused_segments:   # overlap_w holds the raw (clamped) overlaps, one per vertex
            return _segment_reduce.apply(mean_pts, _cum(mask_cnt), overlap_w if self.use_overlap_area_pool else None,
                                         1 if self.use_overlap_area_pool else 2)
        mean = align_and_pack.apply(mean_pts, mask_cnt, 0.0)
        if self.use_overlap_area_pool:
            return torch.sum(mean * overlap_w, dim=1)
        return torch.sum(mean, dim=1) / mask_cnt.unsqueeze(-1)

    def _slot_masks(self, mask, overlap, unique_cnt, idx=None, picked=None):
        """Per slot: number of its vertices next to occupied space, whether any is, and the
        normalised overlap weights of those vertices (utils_bpp_acc.py:668-682).  With `idx` (the indices of
        the True entries of `mask`) the second result is the INDEX list of the slots rather than a bool mask."""
        if self.fused_segments:
            per_slot = pack_and_align.segment_weighted_sum(mask.unsqueeze(-1).to(torch.float).contiguous(),
                                                           None, _cum(unique_cnt.contiguous()), 0)[:, 0]
            mask_exist = per_slot > 0
            if idx is None:
                mask_cnt = per_slot.to(torch.long)[mask_exist]
                picked = overlap[mask]
            else:
                # index lists instead of boolean masks: ONE sync for the slots, reused by the caller for the table rows
                mask_exist = torch.nonzero(mask_exist).squeeze(1)
                mask_cnt = per_slot.to(torch.long).index_select(0, mask_exist)
                if picked is not None:          # clamp(overlap[idx], min = 1) as float32 already (cnc_ctx_compact)
                    return mask_cnt, mask_exist, picked
                picked = overlap.index_select(0, idx)
            return mask_cnt, mask_exist, torch.clamp(picked, min=1).to(torch.float)
        mask_packed = align_and_pack.apply(mask.unsqueeze(-1).to(torch.float), unique_cnt, 0)
        per_slot = torch.sum(mask_packed[:, :, 0], dim=1)
        mask_exist = per_slot > 0
        mask_cnt = per_slot.to(torch.long)[mask_exist]
        ov = torch.clamp(overlap[mask], min=1)
        ov = align_and_pack.apply(ov.unsqueeze(-1).to(torch.float), mask_cnt, 0)
        ov = ov / torch.sum(ov, dim=1, keepdim=True)
        return mask_cnt, mask_exist, ov

    def _chunks_3D(self, n):
        """Slot ranges of level n so that one chunk holds <= MAX_POINTS_NUM_TO_OOM vertices
        (:798-809).  The chunking is part of the bitstream (one .b file per chunk)."""
        hp_n = int(self.hashparams_num_levels[n].item())
        per = min(int(self.MAX_POINTS_NUM_TO_OOM // self.utils_points_per_param_levels[n]), hp_n)
        steps = int(np.ceil(hp_n / per))
        return [(sn, sn * per, min((sn + 1) * per, hp_n)) for sn in range(steps)]

    def _level_chunk_3D(self, Encoding_xyz, n, v0, v1, Pg_n, binary_vxl, outspace_params):
        """Context pass for slots [v0, v1) of coded 3-D level n (shared by encode and decode).
        Returns (mean [num_valid, F], mask_exist, hash_rows [v1-v0])."""
        p0 = self.unique_count_cumsum_list[n, v0]
        p1 = self.unique_count_cumsum_list[n, v1]
        points_n_orig = self.pos_grid_sorted_list[n][p0:p1]
        points_n = (points_n_orig - 0.5) / self.scales_list[n, :]
        mask, overlap = self.query_binary_vxl(points_n_orig, binary_vxl, n, return_overlap_area=True)
        unique_cnt = self.unique_count_list[n, v0:v1]
        mask_cnt, mask_exist, overlap_w = self._slot_masks(mask, overlap, unique_cnt)
        ctx_layers = min(n, self.max_context_layer_num)
        context = Encoding_xyz(points_n[mask], n - ctx_layers, n, outspace_params=outspace_params,
                               binary_vxl=binary_vxl.squeeze(), PV=0)
        if self.fused_heads and context.is_cuda:
            mean_pts = _ctxk.context_mlp(self.context_model_3D, context, None, Pg_n)
        else:
            mean_pts = self.context_model_3D(torch.cat([context, Pg_n.reshape(1, 1).repeat(context.shape[0], 1)], dim=-1))
        mean = self._fuse_3D(mean_pts, mask_cnt, overlap_w)
        rows = self.unique_value_list[n][v0:v1] + self.offsets_list[n]
        return mean, mask_exist, rows

    def _sorted_slots_2D(self, binary_vxl_2D, n):
        indexes_2D, points_n = self.fetch_2D_batches(binary_vxl_2D, n)
        indexes_sorted, order = torch.sort(indexes_2D, descending=False, dim=0, stable=True)
        unique_value, unique_cnt = torch.unique_consecutive(indexes_sorted, return_counts=True)   # sorted already: no second sort
        return points_n, order, unique_value.to(torch.long) + self._off2_host[n], unique_cnt

    def _slot_lists_2D(self, binary_2D):
        """Per plane and coded level: (points, slot order, table rows, slot counts, their running sums)."""
        return [[(lambda t: t + (_cum(t[3]),))(self._sorted_slots_2D(binary_2D[k], n))
                 for n in range(self.n_levels_2D) if self._coded_2D(n)]
                for k in range(3)]

    def _refresh_plane_cats(self, binary_2D):
        """What `_plane_bits` needs of the three planes, from ONE sort: the vertices of every (plane, coded level) keyed
        (plane, level, table row) in one int32, sorted stably once, distinct keys counted once — the same slot order,
        rows and counts as a stable sort + unique per level (nine sorts, nine syncs), concatenated per plane.  Returns
        False when the keys do not fit (the caller then builds the per-level lists)."""
        coded = [n for n in range(self.n_levels_2D) if self._coded_2D(n)]
        nl, F = self.n_levels_2D, self.n_features
        shift = max(int(self._off2_host[n + 1] - self._off2_host[n] - 1).bit_length() for n in coded)
        if (3 * nl) << shift >= 2 ** 31:
            return False
        self._list_occupied_cells_2D(binary_2D)
        keys, pts, sizes = [], [], []
        for k in range(3):
            for n in coded:
                r, p = self.fetch_2D_batches(binary_2D[k], n)
                if r.dtype != torch.int32:
                    return False
                keys.append(r + ((k * nl + n) << shift))
                pts.append(p)
                sizes.append(r.shape[0])
        keys_sorted, order = torch.sort(torch.cat(keys), stable=True)
        uv, uc = torch.unique_consecutive(keys_sorted, return_counts=True)          # sync 1: the number of slots
        consts = self.__dict__.setdefault("_plane_cat_consts", {})
        ck = (nl, shift, str(uv.dtype), str(uv.device))
        if ck not in consts:          # two small host->device copies (synchronising ones), once instead of per refresh
            consts[ck] = (torch.tensor([(k * nl) << shift for k in range(4)], dtype=uv.dtype, device=uv.device),
                          torch.tensor(self._off2_host[:nl], dtype=torch.long, device=uv.device))
        bases, off_lut = consts[ck]
        slot_at = torch.searchsorted(uv, bases).tolist()                            # sync 2: slots per plane
        pts_all = torch.cat(pts)
        # the slots' table rows for the three planes at once (the key's level field is plane * nl + level)
        rows_all = (uv & ((1 << shift) - 1)).to(torch.long) + off_lut[(uv >> shift) % nl]
        at = 0
        for k in range(3):
            a, b = slot_at[k], slot_at[k + 1]
            p_at = [at]
            for i in range(len(coded)):
                p_at.append(p_at[-1] + sizes[k * len(coded) + i])
            A, B = p_at[0], p_at[-1]
            self._plane_cat[k] = dict(
                pts=pts_all[A:B], order=order[A:B] - A, rows=rows_all[a:b], cum=_cum(uc[a:b]),
                segs=[(p_at[i] - A, p_at[i + 1] - A, 0, n * F, n) for i, n in enumerate(coded)])
            at = B
        return True

    def _list_occupied_cells_2D(self, binary_2D):
        """The occupied cells of the three projections from ONE `nonzero` (one host sync instead of three), left in the
        cache `fetch_2D_batches` reads."""
        if not all(b.is_cuda and b.shape == binary_2D[0].shape for b in binary_2D):
            return
        cache = self.__dict__.setdefault("_occ_cells_2D", {})
        keys = [(b.data_ptr(), b._version, tuple(b.shape), tuple(b.stride())) for b in binary_2D]
        if all(k in cache for k in keys):
            return
        flat = torch.stack([b.reshape(-1) for b in binary_2D])            # [3, Rb Rb]
        hit = (flat == 1).nonzero()                                       # [M, 2]: (plane, cell), plane-major
        per_plane = torch.searchsorted(hit[:, 0].contiguous(), torch.arange(4, device=hit.device)).tolist()
        idx2 = self.binary_vxl_2D_idx.view(-1, 2)
        cells = idx2[hit[:, 1]]
        if len(cache) >= 8:
            cache.clear()
        for k, b in enumerate(binary_2D):
            cache[keys[k]] = (b, cells[per_plane[k]:per_plane[k + 1]])

    def _plane_batch_ok(self, p_q):
        """The coded levels of a plane can be evaluated together when every one of them looks at the levels below it
        down to level 0 (n <= max_context_layer_num: the windows [n - min(n, max), n) all start at 0) and the fused
        kernels are in use."""
        coded = [n for n in range(self.n_levels_2D) if self._coded_2D(n)]
        return (self.plane_batched and self.fused_heads and self.fused_segments and p_q.is_cuda and len(coded) > 1
                and min(coded) >= 1 and max(coded) <= self.max_context_layer_num)

    def _plane_bits(self, k, Ec, p_q, Pg_all, bits_all, binary_vxl_2D, pn_frac, refresh):
        """Rate of one plane's table: zero-order bits of the levels that are not coded + the context-coded levels
        (utils_bpp_acc.py:556-572 for n = 1 ..) as ONE encoder call over the concatenated vertex lists (levels
        [0, max n); level n's head reads the first n F columns), one dimension-wise lookup, the heads on their row
        ranges, one per-slot mean, one rate kernel.  Same numbers as the level-by-level loop; a third of the launches."""
        coded = [n for n in range(self.n_levels_2D) if self._coded_2D(n)]
        F = self.n_features
        if self._plane_cat[k] is None:
            levels = self.batched_inputs_list[k]
            at, p_at = 0, []
            for (pts, *_rest) in levels:
                p_at.append(at)
                at += pts.shape[0]
            p_at.append(at)
            cnt = torch.cat([lv[3] for lv in levels])
            self._plane_cat[k] = dict(
                pts=torch.cat([lv[0] for lv in levels]).contiguous(),
                order=torch.cat([lv[1] + p_at[i] for i, lv in enumerate(levels)]),
                rows=torch.cat([lv[2] for lv in levels]), cum=_cum(cnt),
                segs=[(p_at[i], p_at[i + 1], 0, n * F, n) for i, n in enumerate(coded)])
        if self._noncoded_2D is None or self._noncoded_2D.device != bits_all.device:
            self._noncoded_2D = torch.tensor([0.0 if n in coded else 1.0 for n in range(self.n_levels_2D)],
                                             dtype=bits_all.dtype, device=bits_all.device)
        pb = self._plane_cat[k]
        with _range("ctx/2D_mean"):
            context = Ec(pb["pts"], 0, max(coded), binary_vxl=binary_vxl_2D, PV=0)
            context_pn = None
            if self.use_dimension_wise:
                context_pn = Ec.forward_given_params(pb["pts"], self.pn_frac_offsets_list, self.pn_frac_resolutions_list,
                                                     pn_frac, binary_vxl_2D)
            mean_pts = _ctxk.context_heads([self.context_model_2D[n - 1] for n in coded], context, context_pn, Pg_all,
                                           pb["segs"])
            mean = _segment_reduce.apply(mean_pts, pb["cum"], None, 2, pb["order"])
        with _range("ctx/2D_entropy"):
            # (the zero-order levels' bits: an elementwise product and a sum, not torch.dot — the BLAS call cannot be
            # recorded into a HIP graph, cnc_amd._planes_graph)
            return (bits_all * self._noncoded_2D).sum() + self._bits(p_q, pb["rows"], mean)

    def _bits(self, table_q, rows, mean):
        """Rate of the coded rows of a binarised table under the predicted P(+1): sum of
        Bernoulli_entropy(table_q[rows], mean) — one kernel with the gather and the reduction when fused."""
        if self.fused_heads and mean.is_cuda:
            return _ctxk.bernoulli_bits(table_q, rows.contiguous(), mean)
        return torch.sum(self.entropy_model(table_q[rows, :], mean))

    def _coded_2D(self, n):
        return not (n in self.skip_levels_2D or n >= self.Pg_level_2D)

    def _coded_3D(self, n):
        return not (n in self.skip_levels_3D or n >= self.Pg_level)

    # ------------------------------------------------------------------------------- training
    def forward_binary_vxl_mixPg_3D2D(self, Encoding_xyz, Encoding_xy, Encoding_xz, Encoding_yz,
                                      binary_vxl=None, verbose=False, sample_num=None, step=0, sync_MB=True,
                                      stream_2D=None, planes=None):
        """Entropy estimate (bits per parameter) of the four binarised tables under the context
        models; differentiable w.r.t. tables and context models (utils_bpp_acc.py:533-706).

        `stream_2D` (extension): a second HIP stream for the three planes' part.  The planes' bits and the 3-D table's
        bits share nothing but their inputs: with a stream given, the planes' forward is enqueued there (forked from the
        current stream behind the STE of the tables) and the 3-D part on the current stream next to it; autograd runs
        each node's backward on its forward's stream, so one backward call runs the two halves side by side as well.
        Same values: the partial sums are added in the order of the one-stream pass.  The caller orders its stream
        after `stream_2D` once the backward has been enqueued (the trainer's `_context_pass`).

        `planes` (extension) = (bits, parameter count) of the three planes' tables computed elsewhere — the training
        step's captured graph of the planes' half (cnc_amd._planes_graph) — : only the 3-D half runs here, and the
        planes' bits enter the totals as given (no gradient through them: the graph has back-propagated their share)."""
        axes = ("xy", "xz", "yz")
        if planes is not None:          # (on a refresh step the caller has rebuilt the planes' structures: `refresh_planes`)
            with _range("ctx/ste_params"):
                params_q_xyz = self.get_STE_params(Encoding_xyz)
            return self._bits_3D_and_total(Encoding_xyz, params_q_xyz, binary_vxl, sample_num, planes[0], planes[1], None,
                                           None, sync_MB)
        with _range("ctx/ste_params"):
            params_q_xy = self.get_STE_params(Encoding_xy)
            params_q_xz = self.get_STE_params(Encoding_xz)
            params_q_yz = self.get_STE_params(Encoding_yz)
            params_q_xyz = self.get_STE_params(Encoding_xyz)
        ttl_bit_sum, ttl_num_sum = 0, 0

        refresh = self.refresh_planes(binary_vxl, step, params_q_xy)
        idx_coords2, binary_2D = self.idx_coords2_tmp, self._binary_2D

        fork_2D = None
        if stream_2D is not None and params_q_xyz.is_cuda and refresh is False:
            # (a refresh step builds the planes' structures with host round trips inside the loop below: one stream)
            fork_2D = torch.cuda.current_stream(params_q_xyz.device)
            stream_2D.wait_stream(fork_2D)
            for t in (params_q_xy, params_q_xz, params_q_yz, params_q_xyz):
                t.record_stream(stream_2D)
        with (torch.cuda.stream(stream_2D) if fork_2D is not None else contextlib.nullcontext()):
            finest_3D = params_q_xyz[self._off3_host[-2]:self._off3_host[-1]]
            ttl_bit_sum, ttl_num_sum = self._bits_2D(Encoding_xy, Encoding_xz, Encoding_yz, params_q_xy, params_q_xz, params_q_yz,
                                                     finest_3D, binary_vxl, binary_2D, idx_coords2, refresh)

        return self._bits_3D_and_total(Encoding_xyz, params_q_xyz, binary_vxl, sample_num, ttl_bit_sum, ttl_num_sum, fork_2D,
                                       stream_2D, sync_MB)

    def refresh_planes(self, binary_vxl, step, probe):
        """What the planes' half of the pass is built on — the dimension-wise vote plan, the three projections of the
        occupancy grid, the planes' vertex lists and slot orders — rebuilt on every `step_update`-th step (and the
        projections whenever the grid tensor is another one).  Returns whether this step rebuilt them.  `probe`: any
        tensor of the planes' tables' device and dtype (what `_plane_batch_ok` looks at)."""
        axes = ("xy", "xz", "yz")
        refresh = step % self.step_update == 0
        if refresh and binary_vxl is not None:
            # Everything a refresh rebuilds (vote plan: three radix sorts and five gathers over ~2e7 vertices; the planes'
            # vertex lists and slot orders) is a function of the occupancy grid alone.  The estimator re-thresholds its
            # EMA every `step_update` steps, but once surfaces have formed most refreshes change no cell at all: compare
            # with the grid the structures were built from (2 MB, one small kernel + the sync a refresh pays anyway) and
            # keep them when nothing flipped.
            last = getattr(self, "_occ_built_from", None)
            how = (self.planned_votes, self.fused_segments, self.use_dimension_wise, self.fused_heads,
                   self.plane_batched)                                                              # what gets built
            self.refresh_stats["refreshes"] += 1
            if (self.skip_unchanged_refresh and last is not None and how == self._occ_built_how
                    and last.shape == binary_vxl.shape and last.device == binary_vxl.device
                    and bool(torch.equal(last, binary_vxl))):
                refresh = False
                self.refresh_stats["skipped"] += 1
            else:
                self._occ_built_from, self._occ_built_how = binary_vxl.clone(), how
                self.structures_version += 1        # what a captured graph of the planes' half is valid for
        if refresh and self.use_dimension_wise:
            occ = binary_vxl.squeeze(0)
            R_fine = self.dimension_wise_resolution
            t = (R_fine - 2) // self.binary_vxl_len
            if (self.planned_votes and self.fused_segments and occ.is_cuda and occ.dim() == 3 and t >= 1
                    and tuple(occ.shape) == (self.binary_vxl_len,) * 3 and occ.dtype in (torch.bool, torch.uint8)
                    and R_fine == self.binary_vxl_len * t + 2 and R_fine <= 1024 and self.binary_vxl_len <= 128
                    and t == getattr(self, "_idx_coord_t", t)):
                # the plan straight from the occupancy grid: no vertex list, no sort by pixel (the list itself is only
                # made if the unplanned fallback below asks for it)
                self.vote_plan = _backend.VotePlan.from_occupancy(occ.contiguous(), t, R_fine, 2 ** self.log2_hashmap_size)
                self.idx_coords2_tmp = None
            else:
                self.idx_coords2_tmp = self.get_idx_coords2(binary_vxl)
                # the vertex list is fixed until the next refresh: sort it once for the vote kernels
                self.vote_plan = (_backend.VotePlan(self.idx_coords2_tmp.to(torch.int16).contiguous(),
                                                    self.dimension_wise_resolution, 2 ** self.log2_hashmap_size)
                                  if self.planned_votes else None)
        planes_changed = True
        if refresh or getattr(self, "_binary_2D_src", None) is not binary_vxl:
            # the projections (and, keyed on them, the encoders' summed-area tables) live until the occupancy changes.
            # A refresh that flips cells of the 3-D grid usually leaves its three PROJECTIONS as they were (a cell appears
            # or goes behind another one): everything the planes' half is built on below depends on the projections only, so
            # it is kept then — with the OLD projection tensors, which is what the caches keyed on them look at.
            new = [self._project(binary_vxl, a) for a in axes]
            old = getattr(self, "_binary_2D", None)
            how2 = (self.plane_batched, self.fused_heads, self.fused_segments, self._plane_batch_ok(probe))
            if (self.skip_unchanged_refresh and old is not None and getattr(self, "_planes_built_how", None) == how2
                    and getattr(self, "_planes_built_from", None) is old        # (what the structures were built from)
                    and all(o.shape == n_.shape and o.device == n_.device for o, n_ in zip(old, new))
                    and (self._plane_cat[0] is not None or self.batched_inputs_list is not None)
                    and bool(torch.equal(torch.stack(old), torch.stack(new)))):
                planes_changed = False
                if refresh:
                    self.refresh_stats["planes_kept"] = self.refresh_stats.get("planes_kept", 0) + 1
            else:
                self._binary_2D = new
                self._planes_built_how = how2
            self._binary_2D_src = binary_vxl
        binary_2D = self._binary_2D
        if refresh and planes_changed:
            self._planes_built_from = binary_2D
            # vertex lists, slot order and the slots' cumulative counts are fixed until the next refresh; the coded rows the
            # level-by-level branch of `_bits_2D` caches belong to the OLD lists (callers that run `_bits_2D` with
            # refresh=False behind this rebuild — the planes' graph / thread — would otherwise keep using them)
            self._rows_2D_cat = [None, None, None]
            if self._plane_batch_ok(probe) and self._refresh_plane_cats(binary_2D):
                self.batched_inputs_list = None           # the per-level lists: only the level-by-level loop wants them
            else:
                self.batched_inputs_list = self._slot_lists_2D(binary_2D)
                self._plane_cat = [None, None, None]

        return refresh

    def _bits_2D(self, Encoding_xy, Encoding_xz, Encoding_yz, params_q_xy, params_q_xz, params_q_yz, finest_3D,
                 binary_vxl, binary_2D, idx_coords2, refresh):
        """(bits of the three planes' tables, their parameter count): utils_bpp_acc.py:560-617.  `finest_3D`: the
        binarised finest level of the 3-D table (the dimension-wise votes' source)."""
        axes = ("xy", "xz", "yz")
        ttl_bit_sum, ttl_num_sum = 0, 0
        pn_fracs = None
        if self.use_dimension_wise and self.vote_plan is not None and finest_3D.shape[0] <= self.vote_plan.hashmap_size:
            with _range("ctx/pn_frac"):
                pn_fracs = self.get_pn_embed_frac_planes(finest_3D, self.vote_plan)
        for k, (Ec, p_q) in enumerate(zip((Encoding_xy, Encoding_xz, Encoding_yz),
                                          (params_q_xy, params_q_xz, params_q_yz))):
            with _range("ctx/pn_frac"):
                if pn_fracs is not None:
                    pn_frac = pn_fracs[k]
                else:
                    if self.use_dimension_wise and idx_coords2 is None and self.vote_plan is None:
                        idx_coords2 = self.idx_coords2_tmp = self.get_idx_coords2(binary_vxl)
                    pn_frac = (self.get_pn_embed_frac(finest_3D, idx_coords2, axis=axes[k], plan=self.vote_plan)
                               if self.use_dimension_wise else None)
            with _range("ctx/level_Pg"):
                Pg_all, bits_all = self.level_stats(p_q, self._off2_host)
            if self._plane_batch_ok(p_q) and (self._plane_cat[k] is not None or self.batched_inputs_list is not None):
                # the coded levels of the plane in one pass (their context windows all start at level 0)
                ttl_bit_sum = ttl_bit_sum + self._plane_bits(k, Ec, p_q, Pg_all, bits_all, binary_2D[k], pn_frac, refresh)
                ttl_num_sum += p_q.numel()
                continue
            if self.batched_inputs_list is None:
                self.batched_inputs_list = self._slot_lists_2D(binary_2D)
            batches = iter(self.batched_inputs_list[k])
            with _range("ctx/level_Pg"):
                # one unbind each instead of a select per level: a select's backward is a zero-filled [L] vector plus
                # a copy — 43 five-microsecond kernels per step for the 27 selects of the four tables
                Pg_all, bits_all = Pg_all.unbind(0), bits_all.unbind(0)
            # the coded levels of a plane share one rate kernel (their rows are distinct table rows, the bits add up):
            # one gather / one table-sized gradient per plane instead of one per level
            one_rate = self.fused_heads and p_q.is_cuda
            rows_of, means_of = [], []
            for n in range(self.n_levels_2D):
                Pg_n, bits_n = Pg_all[n], bits_all[n]
                if self._coded_2D(n):
                    points_n, order, rows, unique_cnt, cum = next(batches)
                    with _range("ctx/2D_mean"):
                        mean = self._mean_2D(Ec, n, points_n, Pg_n, binary_2D[k], pn_frac, order, unique_cnt, cum=cum)
                    if one_rate:
                        rows_of.append(rows)
                        means_of.append(mean)
                        continue
                    with _range("ctx/2D_entropy"):
                        bits_n = self._bits(p_q, rows, mean)
                ttl_bit_sum = ttl_bit_sum + bits_n
            if rows_of:
                with _range("ctx/2D_entropy"):
                    if refresh or self._rows_2D_cat[k] is None:
                        self._rows_2D_cat[k] = torch.cat(rows_of)          # fixed until the next refresh
                    ttl_bit_sum = ttl_bit_sum + self._bits(p_q, self._rows_2D_cat[k], torch.cat(means_of))
            ttl_num_sum += p_q.numel()
        return ttl_bit_sum, ttl_num_sum

    def _bits_3D_and_total(self, Encoding_xyz, params_q_xyz, binary_vxl, sample_num, bits_2D, ttl_num_sum, fork_2D, stream_2D,
                           sync_MB):
        """The 3-D table's bits (a random contiguous window of hash slots per level, :619-634) + the planes' -> (bits per
        parameter, estimated MB)."""
        later = []           # the 3-D terms, added to the planes' bits at the end in the one-stream pass's order
        if sample_num is not None:
            sample_num_levels = self._sample_allocation(sample_num)
            ttl_sample_valid = sum(int(sample_num_levels[n].item()) for n in range(self.n_levels) if self._coded_3D(n))
        else:
            sample_num_levels, ttl_sample_valid = self.sample_num_levels, self.ttl_sample_num_valid_levels
        v0s = torch.round((self.hashparams_num_levels - sample_num_levels) * self.rand_like(self.utils_rand)).to(torch.long)
        v1s = v0s + sample_num_levels
        p0s = self.unique_count_cumsum_list[self.utils_nlevel_idx, v0s]
        p1s = self.unique_count_cumsum_list[self.utils_nlevel_idx, v1s]

        with _range("ctx/level_Pg"):
            Pg_all, bits_all = self.level_stats(params_q_xyz, self._off3_host)
        # the window bounds of every level in ONE device->host copy
        v0s, v1s, p0s, p1s = torch.stack([v0s, v1s, p0s, p1s]).tolist()
        coded = [n for n in range(self.n_levels) if self._coded_3D(n)]
        if len(coded) < self.n_levels:
            if self._noncoded_3D is None or self._noncoded_3D.device != bits_all.device:
                self._noncoded_3D = torch.tensor([0.0 if n in coded else 1.0 for n in range(self.n_levels)],
                                                 dtype=bits_all.dtype, device=bits_all.device)
            later.append(torch.dot(bits_all, self._noncoded_3D))      # the zero-order levels, one op
        fused = self.fused_heads and params_q_xyz.is_cuda and len(coded) <= 16
        L = self.max_context_layer_num
        if coded and fused:
            with _range("ctx/3D_gather"):
                # vertices, positions, level / resolution per vertex, slot counts and table rows of every coded
                # level's window, concatenated: one kernel
                pts_orig, pts_n, lvl_ids, res_pts, cnts, rows_3D = _ctxk.window_gather(
                    [dict(pos=self.pos_grid_sorted_list[n][p0s[n]:p1s[n]], cnt=self.unique_count_list[n, v0s[n]:v1s[n]],
                          val=self.unique_value_list[n][v0s[n]:v1s[n]], level=n, res=self._res3_host[n],
                          row0=self._off3_host[n]) for n in coded], self.dev)
            with _range("ctx/3D_query"):
                mask, overlap = self._query(pts_orig, binary_vxl, resolution_list=res_pts)
                idx = torch.nonzero(mask).squeeze(1)          # the vertices next to occupied space (one sync)
            with _range("ctx/3D_slot_masks"):
                # positions, levels, window starts and clamped overlaps of the masked vertices: one kernel
                # (cnc_ctx_compact) instead of three index_selects, a subtraction, a clamp and two casts
                pts_m, lvl_m, min_lvl, picked = _ctxk.compact_masked(idx, pts_n, lvl_ids, overlap.contiguous(), L)
                mask_cnt, mask_exist, overlap_w = self._slot_masks(mask, overlap, cnts, idx, picked=picked)
            with _range("ctx/3D_encode"):
                context = Encoding_xyz.forward_diff_levels(pts_m, min_lvl, L, binary_vxl=binary_vxl.squeeze(), PV=1001)
            with _range("ctx/3D_mlp_fuse"):
                # the input row is [context | Pg of the vertex's level]: the column is read from Pg_all by level
                mean_pts = _ctxk.context_mlp(self.context_model_3D, context, None, Pg_all, lvl_m)
                mean = self._fuse_3D(mean_pts, mask_cnt, overlap_w)
            with _range("ctx/3D_entropy"):
                coded_rows = rows_3D[mask_exist] if mask_exist.dtype == torch.bool else rows_3D.index_select(0, mask_exist)
                bits = self._bits(params_q_xyz, coded_rows, mean)
            later.append(bits / ttl_sample_valid * self.ttl_hashparams_num_valid_levels)
        elif coded:
            pts_orig, pts_n, Pg_cols, lvl_ids, cnts, values_q = [], [], [], [], [], []
            for n in coded:
                po = self.pos_grid_sorted_list[n][p0s[n]:p1s[n]]
                pts_orig.append(po)
                pts_n.append((po - 0.5) / self.scales_list[n, :])
                Pg_cols.append(Pg_all[n].reshape(1, 1).repeat(po.shape[0], 1))
                lvl_ids.append(torch.full((po.shape[0],), n, dtype=torch.long, device=self.dev))
                cnts.append(self.unique_count_list[n, v0s[n]:v1s[n]])
                values_q.append(self.unique_value_list[n][v0s[n]:v1s[n]] + self._off3_host[n])   # table rows
            pts_orig, pts_n, Pg_cols = torch.cat(pts_orig), torch.cat(pts_n), torch.cat(Pg_cols)
            lvl_ids, cnts = torch.cat(lvl_ids), torch.cat(cnts)
            rows_3D = torch.cat(values_q)
            mask, overlap = self.query_binary_vxl_qlist(pts_orig, binary_vxl, lvl_ids, return_overlap_area=True)
            mask_cnt, mask_exist, overlap_w = self._slot_masks(mask, overlap, cnts)
            context = Encoding_xyz.forward_diff_levels(pts_n[mask], lvl_ids[mask].to(torch.int) - L, L,
                                                       binary_vxl=binary_vxl.squeeze(), PV=1001)
            context = torch.cat([context, Pg_cols[mask]], dim=-1)
            mean = self._fuse_3D(self.context_model_3D(context), mask_cnt, overlap_w)
            bits = self._bits(params_q_xyz, rows_3D[mask_exist], mean)
            later.append(bits / ttl_sample_valid * self.ttl_hashparams_num_valid_levels)

        if fork_2D is not None:
            fork_2D.wait_stream(stream_2D)
            if isinstance(bits_2D, torch.Tensor):
                bits_2D.record_stream(fork_2D)
        ttl_bit_sum = 0 if bits_2D is None else bits_2D
        for t in later:
            ttl_bit_sum = ttl_bit_sum + t
        ttl_num_sum += params_q_xyz.numel()
        bits_per_param = ttl_bit_sum / ttl_num_sum
        # second value: the estimate in MB as a Python float like the reference (a device->host sync), or — with
        # sync_MB=False — the 0-dim device tensor, for callers that only read it when they log
        est_MB = ttl_bit_sum.detach() * (1.0 / 8388608.0)        # / 8 / 1024 / 1024: a power of two, one op, same value
        return bits_per_param, (est_MB.item() if sync_MB else est_MB)

    # ------------------------------------------------------------------------------- encode
    def encode_binary_vxl_mixPg_3D2D(self, Encoding_xyz, Encoding_xy, Encoding_xz, Encoding_yz,
                                     binary_vxl=None, filename_prefix="b"):
        """Arithmetic-code all four tables into {prefix}_{axis}{n}.b / {prefix}_3D{n}[_{chunk}].b.
        Returns (Pgs_dict, estimated MB, coded MB) (utils_bpp_acc.py:709-865)."""
        Pgs_dict = {}
        params_q_xy = self.get_STE_params(Encoding_xy)
        params_q_xz = self.get_STE_params(Encoding_xz)
        params_q_yz = self.get_STE_params(Encoding_yz)
        params_q_xyz = self.get_STE_params(Encoding_xyz)
        F = self.n_features
        ttl_bit_sum = 0
        coders = CoderPool()
        streams = []            # futures: every stream is an independent file, coded while the next is prepared

        idx_coords2 = self.get_idx_coords2(binary_vxl) if self.use_dimension_wise else None
        finest_3D = params_q_xyz[self.offsets_list[-2]:self.offsets_list[-1]]
        for Ec, p_q, axis in zip((Encoding_xy, Encoding_xz, Encoding_yz),
                                 (params_q_xy, params_q_xz, params_q_yz), ("xy", "xz", "yz")):
            binary_2D = self._project(binary_vxl, axis)
            pn_frac = self.get_pn_embed_frac(finest_3D, idx_coords2, axis=axis) if self.use_dimension_wise else None
            for n in range(self.n_levels_2D):
                Pg_n, bits_n, _ = self.get_BiRF_wentropy_leveln(p_q, n, self.offsets_list_2D)
                Pgs_dict[axis + str(n)] = Pg_n
                fname = f"{filename_prefix}_{axis}{n}.b"
                if not self._coded_2D(n):
                    xs = p_q[self.offsets_list_2D[n]:self.offsets_list_2D[n + 1]].reshape(-1)
                    ps = Pg_n.reshape(1).expand(xs.numel())
                else:
                    points_n, order, rows, unique_cnt = self._sorted_slots_2D(binary_2D, n)
                    mean = self._mean_2D(Ec, n, points_n, Pg_n, binary_2D, pn_frac, order, unique_cnt, detach_pn=True)
                    values_q = p_q[rows, :]
                    bits_n = torch.sum(self.entropy_model(values_q, mean))
                    xs = values_q.reshape(-1)
                    ps = torch.clamp(mean, min=1e-6, max=1 - 1e-6).reshape(-1)
                streams.append(coders.encode(xs, ps, fname))
                ttl_bit_sum = ttl_bit_sum + bits_n

        for n in range(self.n_levels):
            Pg_n, bits_n, _ = self.get_BiRF_wentropy_leveln(params_q_xyz, n)
            Pgs_dict["3D" + str(n)] = Pg_n
            if not self._coded_3D(n):
                xs = params_q_xyz[self.offsets_list[n]:self.offsets_list[n + 1]].reshape(-1)
                ps = Pg_n.reshape(1).expand(xs.numel())
                streams.append(coders.encode(xs, ps, f"{filename_prefix}_3D{n}.b"))
                ttl_bit_sum = ttl_bit_sum + bits_n
                continue
            for sn, v0, v1 in self._chunks_3D(n):
                mean, mask_exist, rows = self._level_chunk_3D(Encoding_xyz, n, v0, v1, Pg_n, binary_vxl, None)
                values_q = params_q_xyz[rows][mask_exist]
                ttl_bit_sum = ttl_bit_sum + torch.sum(self.entropy_model(values_q, mean))
                ps = torch.clamp(mean, min=1e-6, max=1 - 1e-6).reshape(-1)
                streams.append(coders.encode(values_q.reshape(-1), ps, f"{filename_prefix}_3D{n}_{sn}.b"))
        encode_bits = sum(f.result() for f in streams)
        coders.shutdown()
        return Pgs_dict, ttl_bit_sum.item() / 8.0 / 1024 / 1024, encode_bits / 8.0 / 1024 / 1024

    # ------------------------------------------------------------------------------- decode
    def decode_binary_vxl_mixPg_3D2D(self, Encoding_xyz, Encoding_xy, Encoding_xz, Encoding_yz,
                                     params_q_xyz_rec, params_q_xy_rec, params_q_xz_rec,
                                     params_q_yz_rec, binary_vxl=None, Pgs_dict=None, filename_prefix="b"):
        """Sequential inverse of encode: 3-D levels coarse to fine (each level's context reads the
        already decoded lower levels), then the three planes (which need the decoded finest 3-D
        level).  Rows never coded keep the caller's initial value (utils_bpp_acc.py:867-999)."""
        F = self.n_features
        coders = CoderPool()
        dev = params_q_xyz_rec.device
        with torch.no_grad():
            # 3-D: level n needs levels < n decoded, but the chunks of ONE level only read lower levels:
            # their probabilities are computed first, then the chunk streams are decoded concurrently
            for n in range(self.n_levels):
                Pg_n = Pgs_dict["3D" + str(n)]
                if not self._coded_3D(n):
                    rows = int(self.offsets_list[n + 1] - self.offsets_list[n])
                    sout = coders.decode(Pg_n.reshape(1).expand(rows * F), f"{filename_prefix}_3D{n}.b").result()
                    params_q_xyz_rec[self.offsets_list[n]:self.offsets_list[n + 1]] = sout.to(dev).view(rows, F)
                    continue
                pending = []
                for sn, v0, v1 in self._chunks_3D(n):
                    mean, mask_exist, rows = self._level_chunk_3D(Encoding_xyz, n, v0, v1, Pg_n,
                                                                  binary_vxl, params_q_xyz_rec)
                    ps = torch.clamp(mean, min=1e-6, max=1 - 1e-6).reshape(-1)
                    pending.append((coders.decode(ps, f"{filename_prefix}_3D{n}_{sn}.b"), rows[mask_exist], mean.shape))
                for fut, dst, shape in pending:
                    params_q_xyz_rec[dst] = fut.result().to(dev).view(*shape)

            # planes: level n of a plane needs its own levels < n (and the finest 3-D level); the three planes
            # are independent of each other, so each level is decoded for xy / xz / yz concurrently
            idx_coords2 = self.get_idx_coords2(binary_vxl) if self.use_dimension_wise else None
            finest_3D = params_q_xyz_rec[self.offsets_list[-2]:self.offsets_list[-1]]
            planes = []
            for Ec, rec, axis in zip((Encoding_xy, Encoding_xz, Encoding_yz),
                                     (params_q_xy_rec, params_q_xz_rec, params_q_yz_rec), ("xy", "xz", "yz")):
                binary_2D = self._project(binary_vxl, axis)
                pn_frac = self.get_pn_embed_frac(finest_3D, idx_coords2, axis=axis) if self.use_dimension_wise else None
                planes.append((Ec, rec, axis, binary_2D, pn_frac))
            for n in range(self.n_levels_2D):
                pending = []
                for Ec, rec, axis, binary_2D, pn_frac in planes:
                    Pg_n = Pgs_dict[axis + str(n)]
                    fname = f"{filename_prefix}_{axis}{n}.b"
                    if not self._coded_2D(n):
                        rows = int(self.offsets_list_2D[n + 1] - self.offsets_list_2D[n])
                        dst = slice(int(self.offsets_list_2D[n]), int(self.offsets_list_2D[n + 1]))
                        pending.append((coders.decode(Pg_n.reshape(1).expand(rows * F), fname), rec, dst))
                        continue
                    points_n, order, rows, unique_cnt = self._sorted_slots_2D(binary_2D, n)
                    mean = self._mean_2D(Ec, n, points_n, Pg_n, binary_2D, pn_frac, order, unique_cnt,
                                         outspace_params=rec, detach_pn=True)
                    ps = torch.clamp(mean, min=1e-6, max=1 - 1e-6).reshape(-1)
                    pending.append((coders.decode(ps, fname), rec, rows))
                for fut, rec, dst in pending:
                    rec[dst] = fut.result().to(dev).view(-1, F)
        coders.shutdown()
        return params_q_xyz_rec, params_q_xy_rec, params_q_xz_rec, params_q_yz_rec
