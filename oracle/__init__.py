"""CPU oracle for the CNC hot path — TEST INFRASTRUCTURE ONLY.

numpy front-end over ``libcnc_oracle.so`` (built from ``oracle/cnc_oracle.c`` + ``range_coder.c``
by ``oracle/Makefile``).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; nothing under ``cnc_amd/`` does.

Every wrapper mirrors one reference kernel; see the C file for the file:line each follows.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcnc_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("cnc_oracle.c", "range_coder.c", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_grid_index.restype = C.c_uint32
        _lib.orc_openmp_max_threads.restype = C.c_int
        _lib.orc_rc_encode.restype = C.c_int64
        _lib.orc_rc_decode.restype = C.c_int
        _lib.orc_rc_encode2.restype = C.c_int64
        _lib.orc_rc_decode2.restype = C.c_int
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


class Segments(C.Structure):
    _fields_ = [("vals", C.c_void_p), ("chunk_starts", C.c_void_p), ("chunk_cnts", C.c_void_p),
                ("ray_indices", C.c_void_p), ("is_left", C.c_void_p), ("is_right", C.c_void_p),
                ("is_valid", C.c_void_p)]


def max_threads() -> int:
    return int(lib().orc_openmp_max_threads())


# ----------------------------------------------------------------------------- encoder
def grid_index(pos: np.ndarray, hashmap_size: int, resolution: int) -> np.ndarray:
    """pos [..., D] non-negative ints -> row index (uint32)."""
    pos = np.ascontiguousarray(pos, dtype=np.uint32)
    D = pos.shape[-1]
    flat = pos.reshape(-1, D)
    rows = np.empty(flat.shape[0], dtype=np.uint32)
    lib().orc_grid_index_many(C.c_uint32(D), _p(flat), C.c_uint32(flat.shape[0]),
                              C.c_uint32(hashmap_size), C.c_uint32(resolution), _p(rows))
    return rows.reshape(pos.shape[:-1])


def grid_encode_forward(inputs, embeddings, offsets, resolutions, n_levels_calc=None,
                        binary_vxl=None, min_level_id=None, ste_binary=False, threads=1):
    """Returns outputs [L, N, F] float32 (kernel_grid layout)."""
    inputs = _c(inputs, np.float32)
    emb = _c(embeddings, np.float32)
    offsets = _c(offsets, np.int32)
    resolutions = _c(resolutions, np.int32)
    N, D = inputs.shape
    F = emb.shape[1]
    L = int(n_levels_calc) if n_levels_calc is not None else resolutions.shape[0]
    Rb = 128
    vxl = None
    if binary_vxl is not None:
        vxl = _c(binary_vxl, np.uint8)
        Rb = vxl.shape[-1]
    mli = _c(min_level_id, np.int32)
    out = np.empty((L, N, F), dtype=np.float32)
    lib().orc_grid_encode_forward(_p(inputs), _p(emb), _p(offsets), _p(resolutions), _p(out),
                                  C.c_uint32(N), C.c_uint32(D), C.c_uint32(F), C.c_uint32(L),
                                  C.c_uint32(Rb), _p(vxl), _p(mli), C.c_int(int(ste_binary)),
                                  C.c_int(threads))
    return out


def grid_encode_backward(grad, inputs, embeddings, offsets, resolutions, binary_vxl=None,
                         min_level_id=None, ste_binary=False, threads=1, want_acc64=False):
    """grad [L, N, F] -> grad_embeddings [rows, F] (float32, serial accumulation order)
    and, if want_acc64, the same sums accumulated in float64."""
    grad = _c(grad, np.float32)
    inputs = _c(inputs, np.float32)
    emb = _c(embeddings, np.float32)
    offsets = _c(offsets, np.int32)
    resolutions = _c(resolutions, np.int32)
    L, N, F = grad.shape
    D = inputs.shape[1]
    Rb = 128
    vxl = None
    if binary_vxl is not None:
        vxl = _c(binary_vxl, np.uint8)
        Rb = vxl.shape[-1]
    mli = _c(min_level_id, np.int32)
    g = np.zeros_like(emb)
    acc = np.zeros(emb.shape, dtype=np.float64) if want_acc64 else None
    lib().orc_grid_encode_backward(_p(grad), _p(inputs), _p(emb), _p(offsets), _p(resolutions),
                                   _p(g), _p(acc), C.c_uint32(N), C.c_uint32(D), C.c_uint32(F),
                                   C.c_uint32(L), C.c_uint32(Rb), _p(vxl), _p(mli),
                                   C.c_int(int(ste_binary)), C.c_int(threads))
    return (g, acc) if want_acc64 else g


def grid_dy_dx(inputs, embeddings, offsets, resolutions, n_levels_calc=None, min_level_id=None,
               ste_binary=False):
    """dy_dx [N, L, D, F] of kernel_grid's dy_dx branch (gridencoder.cu:319-395)."""
    inputs = _c(inputs, np.float32)
    emb = _c(embeddings, np.float32)
    offsets = _c(offsets, np.int32)
    resolutions = _c(resolutions, np.int32)
    N, D = inputs.shape
    F = emb.shape[1]
    L = int(n_levels_calc) if n_levels_calc is not None else resolutions.shape[0]
    mli = _c(min_level_id, np.int32)
    out = np.empty((N, L, D, F), dtype=np.float32)
    lib().orc_grid_dy_dx(_p(inputs), _p(emb), _p(offsets), _p(resolutions), _p(out), C.c_uint32(N),
                         C.c_uint32(D), C.c_uint32(F), C.c_uint32(L), _p(mli), C.c_int(int(ste_binary)))
    return out


def input_backward(grad, dy_dx):
    """kernel_input_backward (gridencoder.cu:588-614): grad [L, N, F], dy_dx [N, L, D, F] -> [N, D]."""
    grad = _c(grad, np.float32)
    dy_dx = _c(dy_dx, np.float32)
    L, N, F = grad.shape
    D = dy_dx.shape[2]
    out = np.empty((N, D), dtype=np.float32)
    lib().orc_input_backward(_p(grad), _p(dy_dx), _p(out), C.c_uint32(N), C.c_uint32(D), C.c_uint32(F),
                             C.c_uint32(L))
    return out


def cnt_np_embed(inputs, embeddings, resolution, hashmap_size, axis):
    inputs = _c(inputs, np.int16)
    emb = _c(embeddings, np.float32)
    F = emb.shape[1]
    s = resolution - 2
    out = np.zeros((s, s, F, 2), dtype=np.float32)
    lib().orc_cnt_np_embed(_p(inputs), _p(emb), _p(out), C.c_uint32(inputs.shape[0]),
                           C.c_uint32(resolution), C.c_uint32(F), C.c_uint32(hashmap_size),
                           C.c_uint32(axis))
    return out


def cnt_np_embed_backward(inputs, embeddings, outputs_sum, grad, resolution, hashmap_size, axis,
                          want_acc64=False):
    inputs = _c(inputs, np.int16)
    emb = _c(embeddings, np.float32)
    outputs_sum = _c(outputs_sum, np.float32)
    grad = _c(grad, np.float32)
    F = emb.shape[1]
    g = np.zeros_like(emb)
    acc = np.zeros(emb.shape, dtype=np.float64) if want_acc64 else None
    lib().orc_cnt_np_embed_backward(_p(inputs), _p(emb), _p(outputs_sum), _p(grad), _p(g), _p(acc),
                                    C.c_uint32(inputs.shape[0]), C.c_uint32(resolution),
                                    C.c_uint32(F), C.c_uint32(hashmap_size), C.c_uint32(axis))
    return (g, acc) if want_acc64 else g


# ----------------------------------------------------------------------------- aligner
def query_mask(points, binary_vxl, resolution=None, resolution_list=None, contraction=3):
    """`contraction` (3 = the oracle's reading): which multiply-add pairs of aligner_kernel.cu:57,71,233 are ONE fused
    operation, as nvcc's default -fmad=true would make them (bit 0: the cell's upper edge, bit 1: the accumulation)."""
    points = _c(points, np.int16)
    vxl = _c(binary_vxl, np.uint8)
    N, D = points.shape
    assert vxl.ndim == D
    mask = np.zeros(N, dtype=np.int16)
    overlap = np.zeros(N, dtype=np.int32)
    rl = _c(resolution_list, np.int64)
    lib().orc_query_mask_contraction(_p(points), C.c_uint32(D), _p(vxl), C.c_int(vxl.shape[0]), _p(mask),
                                     _p(overlap), C.c_int(0 if resolution is None else int(resolution)),
                                     _p(rl), C.c_uint32(N), C.c_int(int(contraction)))
    return mask, overlap


def align_and_pack_forward(feat, cnt, V=0.0):
    feat = _c(feat, np.float32)
    cnt = _c(cnt, np.int64)
    cumsum = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    N, M, F = cnt.shape[0], int(cnt.max()) if cnt.size else 0, feat.shape[1]
    packed = np.zeros((N, M, F), dtype=np.float32)
    lib().orc_align_and_pack_forward(_p(feat), _p(cnt), _p(cumsum), _p(packed), C.c_uint32(N),
                                     C.c_uint32(M), C.c_uint32(F), C.c_float(V))
    return packed


def align_and_pack_backward(dpacked, cnt, T):
    dpacked = _c(dpacked, np.float32)
    cnt = _c(cnt, np.int64)
    cumsum = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    N, M, F = dpacked.shape
    dfeat = np.zeros((T, F), dtype=np.float32)
    lib().orc_align_and_pack_backward(_p(dpacked), _p(cnt), _p(cumsum), _p(dfeat), C.c_uint32(N),
                                      C.c_uint32(M), C.c_uint32(F))
    return dfeat


# ----------------------------------------------------------------------------- marcher
def ray_aabb_intersect(rays_o, rays_d, aabbs, near=-np.inf, far=np.inf, miss=np.inf):
    rays_o = _c(rays_o, np.float32)
    rays_d = _c(rays_d, np.float32)
    aabbs = _c(aabbs, np.float32)
    n, m = rays_o.shape[0], aabbs.shape[0]
    t0 = np.empty((n, m), dtype=np.float32)
    t1 = np.empty((n, m), dtype=np.float32)
    hits = np.empty((n, m), dtype=np.uint8)
    lib().orc_ray_aabb_intersect(_p(rays_o), _p(rays_d), _p(aabbs), C.c_int32(n), C.c_int32(m),
                                 C.c_float(near), C.c_float(far), C.c_float(miss), _p(t0), _p(t1),
                                 _p(hits))
    return t0, t1, hits.astype(bool)


def traverse_grids(rays_o, rays_d, binaries, aabbs, near_planes=None, far_planes=None,
                   step_size=1e-3, cone_angle=0.0, traverse_steps_limit=None, over_allocate=False,
                   rays_mask=None, t_sorted=None, t_indices=None, hits=None):
    """Host logic of nerfacc/cuda/csrc/grid.cu:356-510 (two passes, or one with over-allocation)
    around the restated kernel.  Returns (intervals, samples, terminate_planes) as dicts of numpy
    arrays with the RaySegmentsSpec field names."""
    rays_o = _c(rays_o, np.float32)
    rays_d = _c(rays_d, np.float32)
    binaries = _c(binaries, np.uint8)
    aabbs = _c(aabbs, np.float32)
    n = rays_o.shape[0]
    m = binaries.shape[0]
    near_planes = np.zeros(n, np.float32) if near_planes is None else _c(near_planes, np.float32)
    far_planes = np.full(n, np.inf, np.float32) if far_planes is None else _c(far_planes, np.float32)
    rays_mask = np.ones(n, np.uint8) if rays_mask is None else _c(rays_mask, np.uint8)
    limit = -1 if traverse_steps_limit is None else int(traverse_steps_limit)
    if t_sorted is None or t_indices is None or hits is None:
        t0, t1, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
        cat = np.concatenate([t0, t1], axis=-1)
        t_indices = np.argsort(cat, axis=-1, kind="stable").astype(np.int64)
        t_sorted = np.take_along_axis(cat, t_indices, axis=-1)
    t_sorted = _c(t_sorted, np.float32)
    t_indices = _c(t_indices, np.int64)
    hits = _c(hits, np.uint8)
    term = np.empty(n, np.float32)

    def alloc(cnts, masks, valid):
        total = int(cnts.sum())
        d = dict(chunk_cnts=cnts, chunk_starts=(np.cumsum(cnts) - cnts).astype(np.int64),
                 vals=np.zeros(total, np.float32), ray_indices=np.zeros(total, np.int64))
        if masks:
            d["is_left"] = np.zeros(total, np.uint8)
            d["is_right"] = np.zeros(total, np.uint8)
        if valid:
            d["is_valid"] = np.zeros(total, np.uint8)
        return d

    def view(d):
        s = Segments()
        for k in ("vals", "chunk_starts", "chunk_cnts", "ray_indices", "is_left", "is_right", "is_valid"):
            setattr(s, k, d[k].ctypes.data if k in d else None)
        return s

    def launch(mask, first, iv, sm, tp):
        siv, ssm = view(iv), view(sm)
        lib().orc_traverse_grids(_p(rays_o), _p(rays_d), _p(mask), C.c_int32(n), _p(binaries),
                                 C.c_int32(m), C.c_int32(binaries.shape[1]),
                                 C.c_int32(binaries.shape[2]), C.c_int32(binaries.shape[3]),
                                 _p(aabbs), _p(hits), _p(t_sorted), _p(t_indices), _p(near_planes),
                                 _p(far_planes), C.c_float(step_size), C.c_float(cone_angle),
                                 C.c_int32(limit), C.c_int32(first), C.byref(siv), C.byref(ssm),
                                 _p(tp))

    if over_allocate:
        assert limit > 0
        iv = alloc((np.full(n, limit * 2, np.int64) * rays_mask.astype(np.int64)), True, False)
        sm = alloc((np.full(n, limit, np.int64) * rays_mask.astype(np.int64)), False, True)
        launch(rays_mask, 0, iv, sm, term)
        for d in (iv, sm):
            d["chunk_starts"] = (np.cumsum(d["chunk_cnts"]) - d["chunk_cnts"]).astype(np.int64)
    else:
        iv = dict(chunk_cnts=np.empty(n, np.int64))
        sm = dict(chunk_cnts=np.empty(n, np.int64))
        launch(None, 1, iv, sm, None)
        iv = alloc(iv["chunk_cnts"], True, False)
        sm = alloc(sm["chunk_cnts"], False, True)
        launch(None, 0, iv, sm, term)
    for d in (iv, sm):
        for k in ("is_left", "is_right", "is_valid"):
            if k in d:
                d[k] = d[k].astype(bool)
    return iv, sm, term


# ----------------------------------------------------------------------------- scans
def segmented_scan(inputs, chunk_starts, chunk_cnts, exclusive, prod=False, reverse=False,
                   normalize=False):
    inputs = _c(inputs, np.float32)
    cs = _c(chunk_starts, np.int64)
    cc = _c(chunk_cnts, np.int64)
    out = np.zeros_like(inputs)
    lib().orc_segmented_scan(_p(cs), _p(cc), _p(inputs), _p(out), C.c_uint32(cs.shape[0]),
                             C.c_int(int(exclusive)), C.c_int(int(prod)), C.c_int(int(reverse)),
                             C.c_int(int(normalize)))
    return out


def prod_backward(inputs, outputs, grad_outputs, chunk_starts, chunk_cnts, exclusive):
    inputs = _c(inputs, np.float32)
    outputs = _c(outputs, np.float32)
    g = _c(grad_outputs, np.float32)
    cs = _c(chunk_starts, np.int64)
    cc = _c(chunk_cnts, np.int64)
    gi = np.zeros_like(inputs)
    lib().orc_prod_backward(_p(cs), _p(cc), _p(inputs), _p(outputs), _p(g), _p(gi),
                            C.c_uint32(cs.shape[0]), C.c_int64(inputs.shape[0]),
                            C.c_int(int(exclusive)))
    return gi


# ----------------------------------------------------------------------------- volume rendering
def render_weight_from_density(t_starts, t_ends, sigmas, chunk_starts, chunk_cnts, prefix_trans=None):
    """nerfacc/volrend.py:258-266,363 on flattened samples: float32 op by op, the running optical depth
    through the restated exclusive_sum tile tree (scan.cu), exp in float32.  (weights, trans, alphas)."""
    ts, te, sg = (_c(a, np.float32) for a in (t_starts, t_ends, sigmas))
    sdt = (sg * (te - ts)).astype(np.float32)
    alphas = (np.float32(1.0) - np.exp(-sdt, dtype=np.float32)).astype(np.float32)
    before = segmented_scan(sdt, chunk_starts, chunk_cnts, exclusive=True)
    trans = np.exp(-before, dtype=np.float32)
    if prefix_trans is not None:
        trans = (trans * _c(prefix_trans, np.float32)).astype(np.float32)
    return (trans * alphas).astype(np.float32), trans, alphas


def composite(weights, rgbs, t_starts, t_ends, ray_indices, n_rays, render_bkgd=None, finalize=True):
    """accumulate_along_rays x3 + the tail of `rendering` (volrend.py:116-140): the sums are taken in
    float64 (the reference's atomics have no order), then rounded.  (colors, opacity, depth) float32."""
    w = np.asarray(weights, np.float64)
    ri = np.asarray(ray_indices, np.int64)
    mid = ((_c(t_starts, np.float32) + _c(t_ends, np.float32)) / np.float32(2.0)).astype(np.float64)
    col = np.zeros((n_rays, 3)); op = np.zeros((n_rays, 1)); dp = np.zeros((n_rays, 1))
    np.add.at(col, ri, w[:, None] * np.asarray(rgbs, np.float32).astype(np.float64))
    np.add.at(op, ri, w[:, None])
    np.add.at(dp, ri, (w * mid)[:, None])
    if finalize:
        dp = dp / np.maximum(op, np.finfo(np.float32).eps)
        if render_bkgd is not None:
            col = col + np.asarray(render_bkgd, np.float64)[None, :] * (1.0 - op)
    return col.astype(np.float32), op.astype(np.float32), dp.astype(np.float32)


def render_visibility(trans, alphas, early_stop_eps=1e-4, alpha_thre=0.0):
    """volrend.py:425-475."""
    vis = np.asarray(trans) >= np.float32(early_stop_eps)
    if alpha_thre > 0:
        vis &= np.asarray(alphas) >= np.float32(alpha_thre)
    return vis


# ----------------------------------------------------------------------------- range coder
def rc_encode(p_one: np.ndarray, symbols: np.ndarray, prob_is_cdf1: bool = False) -> bytes:
    """Binary arithmetic coding of symbols in {0,1} with P(sym=1)=p_one, CDF [0, 1-p, 1]
    (examples/utils_bpp_acc.py:77-93 -> torchac.encode_float_cdf).  prob_is_cdf1: the array holds
    the CDF's middle column 1-p itself (what torchac receives) instead of p."""
    p = _c(p_one, np.float32).reshape(-1)
    s = _c(symbols, np.int16).reshape(-1)
    cap = p.shape[0] // 4 + 64
    while True:
        buf = np.empty(cap, dtype=np.uint8)
        n = lib().orc_rc_encode2(_p(p), C.c_int(int(prob_is_cdf1)), _p(s), C.c_int64(p.shape[0]),
                                 _p(buf), C.c_int64(cap))
        if n >= 0:
            return buf[:n].tobytes()
        cap *= 2


def rc_decode(p_one: np.ndarray, stream: bytes, prob_is_cdf1: bool = False) -> np.ndarray:
    p = _c(p_one, np.float32).reshape(-1)
    buf = np.frombuffer(stream, dtype=np.uint8)
    out = np.empty(p.shape[0], dtype=np.int16)
    rc = lib().orc_rc_decode2(_p(p), C.c_int(int(prob_is_cdf1)), C.c_int64(p.shape[0]), _p(buf),
                              C.c_int64(buf.shape[0]), _p(out))
    assert rc == 0
    return out
