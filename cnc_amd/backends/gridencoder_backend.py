"""Drop-in for the reference's `_gridencoder` extension (gridencoder/src/bindings.cpp:5-9).

Same function names, positional arguments and error behaviour as gridencoder/src/gridencoder.h:12-54:
the caller owns every buffer, results are written in place, nothing is returned, RuntimeError on
non-CUDA / non-contiguous / wrong-dtype tensors.  Kernels run on the current torch stream.
"""
from __future__ import annotations

import os

import torch

from .. import _lib
from .._lib import check, check_contiguous, check_cuda, ptr, stream

_FLOATING = (torch.float32, torch.float16, torch.float64)


def _check_floating(t, name):
    if t.dtype not in _FLOATING:
        raise RuntimeError(f"{name} must be a floating tensor")


def _check_int(t, name):
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")


def _require_f32(t, name):
    # gridencoder.cu:790 also dispatches half/double tables; the CNC drivers never enable
    # autocast (train_CNC_nerf_synthetic.py:211,362 use GradScaler only), so only fp32 is built.
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: libcnc_hip builds the fp32 path only (got {t.dtype})")


def occupancy_sat(binary_vxl):
    """(extension) summed-volume table of a bool occupancy grid [Rb]^D -> int32 [(Rb+1)^D]:
    sat[a,b,c] = number of set cells with indices < (a,b,c)."""
    occ = binary_vxl.to(torch.int32)
    D = occ.dim()
    sat = torch.zeros([n + 1 for n in occ.shape], dtype=torch.int32, device=occ.device)
    inner = occ
    for d in range(D):
        inner = torch.cumsum(inner, dim=d, dtype=torch.int32)
    sat[tuple(slice(1, None) for _ in range(D))] = inner
    return sat.contiguous()


def _check_sat(occ_sat, binary_vxl):
    if occ_sat is None or binary_vxl is None:
        return None
    if (occ_sat.dtype != torch.int32 or not occ_sat.is_cuda or not occ_sat.is_contiguous()
            or tuple(occ_sat.shape) != tuple(n + 1 for n in binary_vxl.shape)):
        raise RuntimeError("occ_sat must be a contiguous CUDA int32 tensor of shape (Rb+1)^D")
    return occ_sat


def occupancy_vertex_bits(binary_vxl, occ_sat, resolutions, max_vertices=1 << 26):
    """(extension) per-level vertex bit planes of the occupancy mask (cnc_grid_vertex_bits): for every level whose
    R^D vertices number at most `max_vertices`, one bit per vertex = the per-corner box test of kernel_grid
    (gridencoder.cu:221-276).  Returns (words int32 [W], offsets int32 [L]: first word of level l's plane, -1 = none)."""
    D = binary_vxl.dim()
    Rb = binary_vxl.shape[-1]
    L = _lib.lib()
    offs, total = [], 0
    for R in resolutions:
        R = int(R)
        if R >= 3 and R ** D <= max_vertices:
            offs.append(total)
            total += int(L.cnc_grid_vertex_bits_words(D, R))
        else:
            offs.append(-1)
    words = torch.empty(max(total, 2), dtype=torch.int32, device=binary_vxl.device)
    sat = _check_sat(occ_sat, binary_vxl)
    for R, o in zip(resolutions, offs):
        if o >= 0:
            check(L.cnc_grid_vertex_bits(ptr(sat), D, int(Rb), int(R), words.data_ptr() + 4 * o,
                                         stream(binary_vxl.device)), "grid_vertex_bits")
    # the offsets depend on the resolutions only: one host->device copy (a synchronising one) per (resolutions, device),
    # not per occupancy refresh and encoder
    key = (tuple(offs), str(binary_vxl.device))
    cached = _VB_OFFSETS.get(key)
    if cached is None:
        cached = _VB_OFFSETS[key] = torch.tensor(offs, dtype=torch.int32, device=binary_vxl.device)
    return words, cached


_VB_OFFSETS = {}


def _vb(vertex_bits, binary_vxl, n_levels):
    """(words pointer, offsets pointer) of an optional `vertex_bits` pair; the offsets must cover the call's levels."""
    if vertex_bits is None or binary_vxl is None:
        return None, None
    words, offs = vertex_bits
    if (words.dtype != torch.int32 or offs.dtype != torch.int32 or not words.is_cuda or not offs.is_cuda
            or not words.is_contiguous() or not offs.is_contiguous() or offs.numel() < n_levels):
        raise RuntimeError("vertex_bits must be (int32 CUDA words, int32 CUDA offsets [>= n_levels])")
    return words.data_ptr(), offs.data_ptr()


def _common_checks(named):
    for name, t in named:
        check_cuda(t, name)
    for name, t in named:
        check_contiguous(t, name)


def grid_encode_forward(inputs, embeddings, offsets_list, resolutions_list, outputs, N, num_dim,
                        n_features, n_levels, max_level, Rb, PV, dy_dx=None, binary_vxl=None,
                        min_level_id=None, *, ste_binary=False, occ_sat=None, out_ld=0, out_col=0, vertex_bits=None):
    _common_checks([("inputs", inputs), ("embeddings", embeddings), ("offsets_list", offsets_list),
                    ("resolutions_list", resolutions_list), ("outputs", outputs)])
    _check_floating(inputs, "inputs")
    _check_floating(embeddings, "embeddings")
    _check_int(offsets_list, "offsets_list")
    _check_int(resolutions_list, "resolutions_list")
    _check_floating(outputs, "outputs")
    for name, t in (("inputs", inputs), ("embeddings", embeddings), ("outputs", outputs)):
        _require_f32(t, name)
    if n_features not in (1, 2, 4, 8, 16, 32):
        raise RuntimeError("GridEncoding: n_fearures must be 1, 2, 4, 8, 16 or 32.")
    if num_dim not in (1, 2, 3):
        raise RuntimeError("GridEncoding: num_dim must be 1, 2, 3.")
    if binary_vxl is not None:
        binary_vxl = binary_vxl.contiguous()
    rc = _lib.lib().cnc_grid_encode_forward(
        ptr(inputs), ptr(embeddings), ptr(offsets_list), ptr(resolutions_list), ptr(outputs),
        int(N), int(num_dim), int(n_features), int(n_levels), int(Rb), float(PV), ptr(dy_dx),
        ptr(binary_vxl), ptr(min_level_id), _lib.CNC_FLAG_STE_BINARY if ste_binary else 0,
        ptr(_check_sat(occ_sat, binary_vxl)), *_vb(vertex_bits, binary_vxl, 0 if min_level_id is not None else n_levels),
        int(out_ld), int(out_col), stream(inputs.device))
    check(rc, "grid_encode_forward")


def grid_encode_backward(grad, inputs, embeddings, offsets_list, resolutions_list, grad_embeddings,
                         N, num_dim, n_features, n_levels, max_level, Rb, dy_dx=None,
                         grad_inputs=None, binary_vxl=None, min_level_id=None, *, ste_binary=False,
                         ste_clip_count=None, occ_sat=None, grad_ld=0, grad_col=0, binned=None,
                         interleave_levels=False, overlap_streams=True, vertex_bits=None, cell_merge=False, cell_carry=False):
    """gridencoder.h:24-36.  `binned` (extension) = (n_binned, level_rows) from `plan_binned_levels`:
    take that many finest levels off the global-atomic path (cnc_grid_encode_backward_binned).
    `interleave_levels` (extension, same result): CNC_FLAG_LEVELS_FINEST_FIRST for the plain entry —
    for calls whose tables are small enough to stay cached while all levels are in flight.
    `cell_merge` / `cell_carry` (extension, same result to fp32 summation order): CNC_FLAG_CELL_MERGE / _CARRY — the
    masked / per-point-level calls of a training step's context pass (grid_encode_cells.hip)."""
    _common_checks([("grad", grad), ("inputs", inputs), ("embeddings", embeddings),
                    ("offsets_list", offsets_list), ("resolutions_list", resolutions_list),
                    ("grad_embeddings", grad_embeddings)])
    _check_floating(grad, "grad")
    _check_floating(inputs, "inputs")
    _check_floating(embeddings, "embeddings")
    _check_int(offsets_list, "offsets_list")
    _check_int(resolutions_list, "resolutions_list")
    _check_floating(grad_embeddings, "grad_embeddings")
    for name, t in (("grad", grad), ("inputs", inputs), ("embeddings", embeddings),
                    ("grad_embeddings", grad_embeddings)):
        _require_f32(t, name)
    if n_features not in (1, 2, 4, 8, 16, 32):
        raise RuntimeError("GridEncoding: n_fearures must be 1, 2, 4, 8, 16 or 32.")
    if num_dim not in (1, 2, 3):
        raise RuntimeError("GridEncoding: num_dim must be 1, 2, 3.")
    if binary_vxl is not None:
        binary_vxl = binary_vxl.contiguous()
    if binned is not None and binned[0] > 0 and binary_vxl is None and min_level_id is None \
            and dy_dx is None and grad_inputs is None:
        n_binned, level_rows = int(binned[0]), int(binned[1])
        L = _lib.lib()
        nbytes = int(L.cnc_grid_encode_backward_binned_workspace(int(N), n_binned, level_rows))
        flags = (_lib.CNC_FLAG_STE_BINARY if ste_binary else 0) | (_lib.CNC_FLAG_BIN_LANE_STORES if _BIN_LANE_STORES else 0)
        if overlap_streams and _OVERLAP_ENABLED:
            # coarse levels on the caller's stream, the finest ones on side streams the library owns through a plan
            # object: fork, join and the split into groups live behind the C ABI (grid_encode_overlap.hip)
            cur = torch.cuda.current_stream(grad.device).cuda_stream
            nbytes = int(L.cnc_grid_encode_backward_overlapped_workspace(int(N), n_binned, level_rows))
            ws = _workspace(grad.device, nbytes, (cur, 0))
            rc = L.cnc_grid_encode_backward_overlapped(
                _plan(grad.device, cur), ptr(grad), ptr(inputs), ptr(embeddings), ptr(offsets_list), ptr(resolutions_list),
                ptr(grad_embeddings), int(N), int(num_dim), int(n_features), int(n_levels), flags, ptr(ste_clip_count),
                int(grad_ld), int(grad_col), n_binned, level_rows, ptr(ws),
                nbytes, stream(grad.device))      # (the size THIS call asked for, not what an earlier, larger call left
            check(rc, "grid_encode_backward_overlapped")     # behind: the library deepens its bins with the scratch it is given)
            return
        ws = _workspace(grad.device, nbytes, (torch.cuda.current_stream(grad.device).cuda_stream, 0))
        rc = L.cnc_grid_encode_backward_binned(
            ptr(grad), ptr(inputs), ptr(embeddings), ptr(offsets_list), ptr(resolutions_list),
            ptr(grad_embeddings), int(N), int(num_dim), int(n_features), int(n_levels), flags,
            ptr(ste_clip_count), int(grad_ld), int(grad_col), n_binned, level_rows, ptr(ws), nbytes,
            stream(grad.device))
        check(rc, "grid_encode_backward_binned")
        return
    rc = _lib.lib().cnc_grid_encode_backward(
        ptr(grad), ptr(inputs), ptr(embeddings), ptr(offsets_list), ptr(resolutions_list),
        ptr(grad_embeddings), int(N), int(num_dim), int(n_features), int(n_levels), int(Rb),
        ptr(dy_dx), ptr(grad_inputs), ptr(binary_vxl), ptr(min_level_id),
        (_lib.CNC_FLAG_STE_BINARY if ste_binary else 0)
        | (_lib.CNC_FLAG_LEVELS_FINEST_FIRST if interleave_levels else 0)
        | (_lib.CNC_FLAG_CELL_MERGE if cell_merge else 0)
        | (_lib.CNC_FLAG_CELL_CARRY if cell_merge and cell_carry else 0), ptr(ste_clip_count),
        ptr(_check_sat(occ_sat, binary_vxl)), *_vb(vertex_bits, binary_vxl, 0 if min_level_id is not None else n_levels),
        int(grad_ld), int(grad_col), stream(grad.device))
    check(rc, "grid_encode_backward")


_WORKSPACES = {}
_PLANS = {}
_OVERLAP_ENABLED = os.environ.get("CNC_BWD_OVERLAP", "1") != "0"   # measurement switch (profiles/)
_BIN_LANE_STORES = os.environ.get("CNC_BWD_BIN_LANE_STORES", "0") == "1"   # measurement switch: the round-2 bin pass


def _plan(device, caller_stream):
    """One cnc_backward_plan (two side streams + events, owned by the library) per (device, caller stream)."""
    import ctypes as C
    key = (device.type, device.index, caller_stream)
    h = _PLANS.get(key)
    if h is None:
        out = C.c_void_p()
        with torch.cuda.device(device):
            check(_lib.lib().cnc_backward_plan_create(C.byref(out)), "backward_plan_create")
        h = _PLANS[key] = out
    return h


def _workspace(device, nbytes, which=0):
    """Scratch for the binned backward, grown on demand and reused (stream-ordered reuse is safe:
    every call clears what it reads).  `which`: one buffer per concurrently running level group."""
    key = (device.type, device.index, which)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _WORKSPACES[key] = ws
    return ws


def plan_binned_levels(resolutions, offsets, num_dim, n_features, n_points, min_resolution=400,
                       min_points=1 << 16, min_work=3 << 19):
    """Which finest levels the binned backward should take: (n_binned, level_rows) or None.

    `resolutions` / `offsets` are HOST sequences (the library never reads device tables on the host).
    A level qualifies when its cells are finer than typical sample spacing (resolution >=
    min_resolution: below that, consecutive ray samples and the samples of neighbouring rays share cells and
    the run-merging atomic kernel is the cheaper one) and its table has at least 2^16 rows; the qualifying
    levels must be the last ones.  min_resolution: 288 until round 3 (7 binned levels of the 16-level bench grid);
    with the sorted bin pass and the overlapped call the R = 296 level is cheaper on the merge kernel
    (bench call 1.008 -> 0.980 ms per 2^20 samples; 200 / 288 / 400 / 560: 1.082 / 1.012 / 0.989 / 1.093 ms)."""
    if num_dim != 3 or n_features not in (2, 4, 8) or n_points < min_points or n_points >= 1 << 24:
        return None
    if "CNC_BIN_MIN_RES" in os.environ:          # measurement switch
        min_resolution = int(os.environ["CNC_BIN_MIN_RES"])
    res = [int(r) for r in resolutions]
    off = [int(o) for o in offsets]
    n, rows = 0, 0
    for l in range(len(res) - 1, -1, -1):
        r = off[l + 1] - off[l]
        if res[l] < min_resolution or r < (1 << 16) or r > (1 << 20):
            break
        n += 1
        rows = max(rows, r)
    # The bin pass works in blocks of 4096 samples per level and the coarse half moves to the run-merging kernel: with
    # few binned levels or few samples that is a handful of blocks on 256 CUs and the plain atomic kernel wins — measured
    # (ms per backward call, samples along rays): 6 binned levels 2^17 / 2^18 / 2^19 samples 0.29 / 0.46 / 0.72 against
    # 0.29 / 0.52 / 0.97 atomic; ONE binned level (the 12-level training grid) 2^18 / 2^20 samples 0.34 / 0.97 against
    # 0.25 / 0.86 — in the training step 0.93 against 0.34 ms.  Hence: samples x binned levels >= `min_work` = 1.5 M.
    if n_points * n < min_work and "CNC_BIN_MIN_RES" not in os.environ:
        return None
    return (n, rows) if n else None


def pack_sign_bits(embeddings, bits=None, clip_count=None):
    """(extension) sign bit plane of a [rows, F] fp32 table: uint8 [ceil(rows*F/8)], bit = value >= 0.
    clip_count: optional CUDA int32/uint32 tensor [1], set to the number of entries with |v| > 1."""
    _common_checks([("embeddings", embeddings)])
    _require_f32(embeddings, "embeddings")
    rows, F = embeddings.shape
    n_bytes = (rows * F + 7) // 8
    if bits is None:
        bits = torch.empty(n_bytes, dtype=torch.uint8, device=embeddings.device)
    elif bits.numel() != n_bytes or bits.dtype != torch.uint8 or not bits.is_cuda:
        raise RuntimeError("bits must be a CUDA uint8 tensor of ceil(rows*F/8) bytes")
    rc = _lib.lib().cnc_pack_sign_bits(ptr(embeddings), ptr(bits), int(rows), int(F),
                                       ptr(clip_count), stream(embeddings.device))
    check(rc, "pack_sign_bits")
    return bits


def grid_encode_forward_bits(inputs, bits, offsets_list, resolutions_list, outputs, N, num_dim,
                             n_features, n_levels, Rb, binary_vxl=None, min_level_id=None, occ_sat=None,
                             out_ld=0, out_col=0, vertex_bits=None):
    """(extension) grid_encode_forward on the bit plane of a binarised table; same outputs as
    grid_encode_forward(..., ste_binary=True) on the fp32 table."""
    _common_checks([("inputs", inputs), ("bits", bits), ("offsets_list", offsets_list),
                    ("resolutions_list", resolutions_list), ("outputs", outputs)])
    _check_int(offsets_list, "offsets_list")
    _check_int(resolutions_list, "resolutions_list")
    _require_f32(inputs, "inputs")
    _require_f32(outputs, "outputs")
    if bits.dtype != torch.uint8:
        raise RuntimeError("bits must be a uint8 tensor")
    if n_features not in (1, 2, 4, 8, 16, 32):
        raise RuntimeError("GridEncoding: n_fearures must be 1, 2, 4, 8, 16 or 32.")
    if num_dim not in (1, 2, 3):
        raise RuntimeError("GridEncoding: num_dim must be 1, 2, 3.")
    if binary_vxl is not None:
        binary_vxl = binary_vxl.contiguous()
    rc = _lib.lib().cnc_grid_encode_forward_bits(
        ptr(inputs), ptr(bits), ptr(offsets_list), ptr(resolutions_list), ptr(outputs), int(N),
        int(num_dim), int(n_features), int(n_levels), int(Rb), ptr(binary_vxl), ptr(min_level_id),
        ptr(_check_sat(occ_sat, binary_vxl)), *_vb(vertex_bits, binary_vxl, 0 if min_level_id is not None else n_levels),
        int(out_ld), int(out_col), stream(inputs.device))
    check(rc, "grid_encode_forward_bits")


def cnt_np_embed(inputs, embeddings_clip, outputs, N, resolution, n_features, hashmap_size, axis):
    _common_checks([("inputs", inputs), ("embeddings_clip", embeddings_clip), ("outputs", outputs)])
    _check_floating(embeddings_clip, "embeddings_clip")
    _check_floating(outputs, "outputs")
    if inputs.dtype != torch.int16:
        raise RuntimeError("expected scalar type Short for inputs")   # data_ptr<short>() mismatch
    _require_f32(embeddings_clip, "embeddings_clip")
    _require_f32(outputs, "outputs")
    if n_features not in (1, 2, 4, 8, 16, 32):
        raise RuntimeError("GridEncoding: n_features must be 1, 2, 4, 8, 16 or 32.")
    rc = _lib.lib().cnc_cnt_np_embed(ptr(inputs), ptr(embeddings_clip), ptr(outputs), int(N),
                                     int(resolution), int(n_features), int(hashmap_size), int(axis),
                                     stream(inputs.device))
    check(rc, "cnt_np_embed")


def cnt_np_embed_backward(inputs, embeddings_clip, outputs_sum, grad, grad_embeddings, N,
                          resolution, n_features, hashmap_size, axis):
    _common_checks([("inputs", inputs), ("embeddings_clip", embeddings_clip),
                    ("outputs_sum", outputs_sum), ("grad", grad),
                    ("grad_embeddings", grad_embeddings)])
    for name, t in (("embeddings_clip", embeddings_clip), ("outputs_sum", outputs_sum),
                    ("grad", grad), ("grad_embeddings", grad_embeddings)):
        _check_floating(t, name)
        _require_f32(t, name)
    if inputs.dtype != torch.int16:
        raise RuntimeError("expected scalar type Short for inputs")
    if n_features not in (1, 2, 4, 8, 16, 32):
        raise RuntimeError("GridEncoding: n_features must be 1, 2, 4, 8, 16 or 32.")
    rc = _lib.lib().cnc_cnt_np_embed_backward(ptr(inputs), ptr(embeddings_clip), ptr(outputs_sum),
                                              ptr(grad), ptr(grad_embeddings), int(N),
                                              int(resolution), int(n_features), int(hashmap_size),
                                              int(axis), stream(inputs.device))
    check(rc, "cnt_np_embed_backward")


class VotePlan:
    """(extension) Static plan for `cnt_np_embed` on one vertex list: the list sorted by pixel of each
    projection plane (forward) and by table row of the finest level (backward).  Built once per
    occupancy refresh; see cnc_amd/csrc/cnt_votes.hip."""

    def __init__(self, inputs_i16, resolution, hashmap_size):
        _common_checks([("inputs", inputs_i16)])
        if inputs_i16.dtype != torch.int16:
            raise RuntimeError("expected scalar type Short for inputs")
        N, dev = inputs_i16.shape[0], inputs_i16.device
        self.resolution, self.hashmap_size = int(resolution), int(hashmap_size)
        self.n_pixels = (self.resolution - 2) ** 2
        self._masks = self._masks_key = self._masks_src = None
        L = _lib.lib()
        # one kernel: table row and the three planes' pixels of every vertex; the vertices cnt_np_embed skips are keyed
        # past the last row / pixel, so a sort leaves them behind the last segment and nothing is compacted; the same
        # kernel says which pixel lists are already ascending (the xy plane of an (x, y, z)-sorted list)
        rows = torch.empty(N, dtype=torch.int32, device=dev)
        pix = [torch.empty(N, dtype=torch.int32, device=dev) for _ in range(3)]
        unsorted = torch.zeros(3, dtype=torch.int32, device=dev)
        check(L.cnc_cnt_np_plan3(ptr(inputs_i16), N, self.resolution, self.hashmap_size, ptr(rows), ptr(pix[0]),
                                 ptr(pix[1]), ptr(pix[2]), ptr(unsorted), stream(dev)), "cnt_np_plan3")
        unsorted = unsorted.tolist() if N else [0, 0, 0]
        # forward: rows ordered by pixel, per plane (32-bit keys: half the radix passes of int64)
        self.rows_by_pixel, self.pixel_seg = [], []
        for axis in range(3):
            p = pix[axis]
            if not unsorted[axis]:
                rows_sorted, p_sorted = rows, p
            else:
                p_sorted, order = torch.sort(p, stable=True)
                rows_sorted = rows[order]
            self.rows_by_pixel.append(rows_sorted.contiguous())
            self.pixel_seg.append(self._segments(p_sorted, self.n_pixels))
        # backward: pixels ordered by row (one order for the three planes)
        rows_sorted, order = torch.sort(rows, stable=True)
        self._pixels_by_row = [p[order].contiguous() for p in pix]
        self.xyz_by_row = None
        self.row_seg = self._segments(rows_sorted, self.hashmap_size)

    @classmethod
    def from_occupancy(cls, occupancy, t, resolution, hashmap_size):
        """The same plan as `VotePlan(get_idx_coords2(binary_vxl), ...)` without the vertex list: per-pixel counts of the
        vertex set decided line by line from the bit-packed occupancy (cnc_vote_plan_count), their running sums, rows
        written in each plane's pixel-major order (cnc_vote_plan_fill: what a stable sort by pixel of the (x, y,
        z)-ordered list gives), ONE sort by table row of the packed vertices for the backward.  occupancy: bool / uint8
        [Rb, Rb, Rb], Rb <= 128; resolution = Rb t + 2 <= 1024."""
        _common_checks([("occupancy", occupancy)])
        Rb = occupancy.shape[-1]
        if occupancy.dim() != 3 or occupancy.shape != (Rb, Rb, Rb) or occupancy.dtype not in (torch.bool, torch.uint8) \
                or int(resolution) != Rb * int(t) + 2 or int(resolution) > 1024 or Rb > 128:
            raise RuntimeError("VotePlan.from_occupancy: occupancy [Rb, Rb, Rb] of bytes and resolution = Rb t + 2 <= 1024")
        self = cls.__new__(cls)
        dev, R = occupancy.device, int(resolution)
        self.resolution, self.hashmap_size = R, int(hashmap_size)
        self.n_pixels = (R - 2) ** 2
        self._masks = self._masks_key = self._masks_src = None
        L, st = _lib.lib(), stream(dev)
        bits = torch.empty(3 * Rb * Rb * 4, dtype=torch.int32, device=dev)
        counts = torch.empty((3, self.n_pixels), dtype=torch.int32, device=dev)
        check(L.cnc_vote_plan_count(ptr(occupancy), Rb, int(t), ptr(bits), ptr(counts), st), "vote_plan_count")
        # running sums per plane from ONE flat scan (a [3, P] scan along P is three serial rows: 0.5 ms)
        flat = torch.cumsum(counts.view(-1), 0, dtype=torch.int32).view(3, self.n_pixels)
        seg = torch.zeros((3, self.n_pixels + 1), dtype=torch.int32, device=dev)
        seg[:, 1:] = flat
        seg[1:, 1:] -= flat[:-1, -1:]                       # every plane holds the same n vertices
        n = int(seg[0, -1].item())                          # the one sync: sizes the lists
        rows = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3)]
        xyz = torch.empty(n, dtype=torch.int32, device=dev)
        self.pixel_seg = [seg[0], seg[1], seg[2]]
        if n:
            check(L.cnc_vote_plan_fill(ptr(bits), Rb, int(t), self.hashmap_size, ptr(seg), ptr(rows[0]), ptr(rows[1]),
                                       ptr(rows[2]), ptr(xyz), st), "vote_plan_fill")
        self.rows_by_pixel = rows
        rows_sorted, order = torch.sort(rows[0], stable=True)      # rows[0] is in (x, y, z) order, as xyz is
        self.xyz_by_row = torch.index_select(xyz, 0, order)       # (the generic `xyz[order]` gather took 0.22 ms for 3.5 M)
        self.row_seg = self._segments(rows_sorted, self.hashmap_size)
        self._pixels_by_row = None
        return self

    @property
    def pixels_by_row(self):
        """Per plane, the vertices' pixels ordered by table row — kept by the list constructor, derived from the packed
        vertices (only the single-plane backward asks for them) by `from_occupancy`."""
        if self._pixels_by_row is None:
            q, S = self.xyz_by_row, self.resolution - 2
            x, y, z = (q & 1023) - 1, ((q >> 10) & 1023) - 1, ((q >> 20) & 1023) - 1
            self._pixels_by_row = [(x * S + y).contiguous(), (x * S + z).contiguous(), (y * S + z).contiguous()]
        return self._pixels_by_row

    @staticmethod
    def _segments(sorted_keys, n):
        """seg[k] = number of keys < k for k = 0 .. n (int32), from the SORTED key list: a binary search per
        boundary instead of a histogram of 10^7 keys with atomics (0.5 ms each, four per occupancy refresh)."""
        bounds = torch.arange(n + 1, dtype=sorted_keys.dtype, device=sorted_keys.device)
        return torch.searchsorted(sorted_keys, bounds, right=False).to(torch.int32)


def cnt_np_embed_planned(plan, embeddings_clip, outputs, n_features, axis):
    """outputs [res-2, res-2, F, 2] written with cnt_np_embed's counts (no zero-fill needed)."""
    _common_checks([("embeddings_clip", embeddings_clip), ("outputs", outputs)])
    _require_f32(embeddings_clip, "embeddings_clip")
    _require_f32(outputs, "outputs")
    if outputs.numel() != plan.n_pixels * n_features * 2:
        raise RuntimeError("cnt_np_embed_planned: tensor sizes do not match the plan")
    L = _lib.lib()
    # the votes (embedding > 0.9) as one word per table row, packed once per table version and shared by the three
    # projections of a step: the per-vertex gather is then 4 bytes from a 2 MB array instead of a 4 F byte row
    key = (embeddings_clip.data_ptr(), embeddings_clip._version, tuple(embeddings_clip.shape))
    if plan._masks_key != key:
        n_rows = min(plan.hashmap_size, embeddings_clip.shape[0])
        masks = torch.empty(n_rows, dtype=torch.int32, device=embeddings_clip.device)
        check(L.cnc_cnt_vote_masks(ptr(embeddings_clip), n_rows, int(n_features), ptr(masks),
                                   stream(outputs.device)), "cnt_vote_masks")
        plan._masks, plan._masks_key, plan._masks_src = masks, key, embeddings_clip
    rc = L.cnc_cnt_np_embed_planned_masked(ptr(plan.rows_by_pixel[axis]), ptr(plan.pixel_seg[axis]),
                                           ptr(plan._masks), ptr(outputs), plan.n_pixels,
                                           int(n_features), stream(outputs.device))
    check(rc, "cnt_np_embed_planned")


def cnt_np_embed_planned_backward(plan, embeddings_clip, grad_over_sum, grad_embeddings, n_features, axis):
    _common_checks([("embeddings_clip", embeddings_clip), ("grad_over_sum", grad_over_sum),
                    ("grad_embeddings", grad_embeddings)])
    for name, t in (("embeddings_clip", embeddings_clip), ("grad_over_sum", grad_over_sum),
                    ("grad_embeddings", grad_embeddings)):
        _require_f32(t, name)
    if grad_over_sum.numel() != plan.n_pixels * n_features * 2 or grad_embeddings.shape != embeddings_clip.shape:
        raise RuntimeError("cnt_np_embed_planned_backward: tensor sizes do not match the plan")
    rc = _lib.lib().cnc_cnt_np_embed_planned_backward(ptr(plan.pixels_by_row[axis]), ptr(plan.row_seg),
                                                      ptr(embeddings_clip), ptr(grad_over_sum),
                                                      ptr(grad_embeddings),
                                                      min(plan.hashmap_size, embeddings_clip.shape[0]),
                                                      int(n_features), stream(grad_embeddings.device))
    check(rc, "cnt_np_embed_planned_backward")


def cnt_np_embed_planned_backward3(plan, embeddings_clip, grads_over_sum, grad_embeddings, n_features):
    """The three planes' `cnt_np_embed_planned_backward` in one pass; `grad_embeddings` is written, not accumulated."""
    ts = [("embeddings_clip", embeddings_clip), ("grad_embeddings", grad_embeddings)] + \
         [(f"grad_over_sum[{i}]", g) for i, g in enumerate(grads_over_sum)]
    _common_checks(ts)
    for name, t in ts:
        _require_f32(t, name)
    if len(grads_over_sum) != 3 or any(g.numel() != plan.n_pixels * n_features * 2 for g in grads_over_sum) \
            or grad_embeddings.shape != embeddings_clip.shape or embeddings_clip.shape[0] > plan.hashmap_size:
        raise RuntimeError("cnt_np_embed_planned_backward3: tensor sizes do not match the plan")
    if plan.xyz_by_row is not None:        # one packed vertex per entry instead of three pixels
        check(_lib.lib().cnc_cnt_np_embed_planned_backward3_xyz(
            ptr(plan.xyz_by_row), ptr(plan.row_seg), ptr(embeddings_clip), ptr(grads_over_sum[0]), ptr(grads_over_sum[1]),
            ptr(grads_over_sum[2]), ptr(grad_embeddings), embeddings_clip.shape[0], int(n_features), plan.resolution,
            stream(grad_embeddings.device)), "cnt_np_embed_planned_backward3_xyz")
        return
    rc = _lib.lib().cnc_cnt_np_embed_planned_backward3(
        ptr(plan.pixels_by_row[0]), ptr(plan.pixels_by_row[1]), ptr(plan.pixels_by_row[2]), ptr(plan.row_seg),
        ptr(embeddings_clip), ptr(grads_over_sum[0]), ptr(grads_over_sum[1]), ptr(grads_over_sum[2]),
        ptr(grad_embeddings), embeddings_clip.shape[0], int(n_features), stream(grad_embeddings.device))
    check(rc, "cnt_np_embed_planned_backward3")
