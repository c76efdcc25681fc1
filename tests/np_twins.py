"""Second, independent pins for the oracle (test infrastructure, like oracle/ itself).

Vectorised NumPy restatements written directly from the reference's CUDA sources — NOT from
oracle/cnc_oracle.c and sharing no code with it — so that an error in the C restatement cannot hide
behind "the HIP kernel agrees with the oracle":

  * `grid_corners`, `grid_encode_forward`, `grid_encode_backward64`   gridencoder/src/gridencoder.cu:46-316,411-584
  * `cnt_np_embed`                                                    gridencoder/src/gridencoder.cu:873-915
  * `query_mask_3D`                                                   my_cuda_backen/aligner_kernel.cu:161-242
  * `align_and_pack_forward`                                          my_cuda_backen/aligner_kernel.cu:421-434
  * `traverse_grids` (one grid, the CNC case)                         nerfacc/cuda/csrc/grid.cu:71-318 +
                                                                      include/utils_grid.cuh:59-149

Arithmetic: everything float32, operation by operation, as the CUDA source spells it.  Where a C++
double literal promotes an expression it is evaluated in float64 and rounded once.  Where nvcc's
default -fmad=true contracts `a*b + c` (same-type operands) the fused result is emulated as
float32(float64(a)*float64(b) + float64(c)): the product of two float32 is exact in float64; the sum is
rounded to 53 bits and then to 24, which differs from a true fma only when the 53-bit value sits
exactly on a float32 rounding boundary (never observed on the seeded inputs of the tests).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
PRIMES = np.array([1, 2654435761, 805459861], dtype=np.uint32)


def fma32(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


# ----------------------------------------------------------------------------------------------
# hash-grid encoder
# ----------------------------------------------------------------------------------------------
def grid_index(q, hashmap_size, R):
    """q uint32 [..., D] -> row (without the *F).  gridencoder.cu:62-87."""
    q = np.asarray(q, np.uint32)
    D = q.shape[-1]
    stride = 1
    index = np.zeros(q.shape[:-1], np.uint32)
    d = 0
    while d < D and stride <= hashmap_size:
        index = (index + q[..., d] * np.uint32(stride & 0xFFFFFFFF)).astype(np.uint32)
        stride = (stride * R) & 0xFFFFFFFF          # uint32 wrap like the kernel
        d += 1
    if stride > hashmap_size:
        h = np.zeros(q.shape[:-1], np.uint32)
        for i in range(D):
            h ^= (q[..., i] * PRIMES[i]).astype(np.uint32)
        index = h
    return index % np.uint32(hashmap_size)


def _box_any(q, R, vxl):
    """Occupancy test of the (2/(R-2))-wide box around vertex q.  gridencoder.cu:219-276."""
    Rb = vxl.shape[-1]
    D = q.shape[-1]
    s = f32(1.0 / (float(f32(R)) - 2.0))
    pn = ((q.astype(f32).astype(np.float64) - 0.5) * np.float64(s)).astype(f32)
    lo = np.clip((pn - s) * f32(Rb), f32(0), f32(Rb - 1)).astype(np.int64)
    hi = np.clip((pn + s) * f32(Rb), f32(0), f32(Rb - 1)).astype(np.int64)
    # inclusive box sums through a summed-volume table (an OR over the box == sum > 0)
    sat = vxl.astype(np.int64)
    for ax in range(D):
        sat = np.cumsum(sat, axis=ax)
    sat = np.pad(sat, [(1, 0)] * D)
    tot = np.zeros(q.shape[:-1], np.int64)
    for corner in range(1 << D):
        idx = tuple(np.where((corner >> ax) & 1, hi[..., ax] + 1, lo[..., ax]) for ax in range(D))
        sign = (-1) ** (D - bin(corner).count("1"))
        tot += sign * sat[idx]
    return tot > 0


def grid_corners(x, R, hashmap_size, vxl=None):
    """Per point and corner: table row, normalised weight w/sum(w_valid), validity; plus the
    in-range flag of the point.  gridencoder.cu:143-291."""
    x = np.asarray(x, f32)
    N, D = x.shape
    inside = ~np.any((x < 0) | (x > 1), axis=1)
    pos = ((x * f32(R - 2)).astype(np.float64) + 0.5).astype(f32)
    g = np.floor(pos).astype(np.uint32)
    fr = pos - g.astype(f32)
    nC = 1 << D
    rows = np.zeros((N, nC), np.uint32)
    w = np.ones((N, nC), f32)
    valid = np.ones((N, nC), bool)
    for c in range(nC):
        q = np.empty((N, D), np.uint32)
        for d in range(D):
            if (c >> d) & 1:
                w[:, c] = w[:, c] * fr[:, d]
                q[:, d] = np.minimum(g[:, d] + np.uint32(1), np.uint32(R - 1))
            else:
                w[:, c] = w[:, c] * (f32(1) - fr[:, d])
                q[:, d] = g[:, d]
        border = np.any((q == 0) | (q == R - 1), axis=1)
        m = np.ones(N, bool) if vxl is None else _box_any(q, R, vxl)
        valid[:, c] = ~border & m
        rows[:, c] = grid_index(q, hashmap_size, R)
    wn = np.zeros(N, f32)
    for c in range(nC):                              # float32 running sum in corner order
        wn = np.where(valid[:, c], wn + w[:, c], wn)
    wn = np.where(wn == 0, f32(1e-9), wn)
    wn_re = (1.0 / wn.astype(np.float64)).astype(f32)
    valid &= inside[:, None]
    return rows, w * wn_re[:, None], valid, inside


def grid_encode_forward(x, emb, offsets, resolutions, vxl=None, ste_binary=False):
    """[L, N, F] float32.  The accumulate is the kernel's `results += (w*wn_re) * e` with fmad."""
    emb = np.asarray(emb, f32)
    N, F = x.shape[0], emb.shape[1]
    L = len(resolutions)
    out = np.zeros((L, N, F), f32)
    for l in range(L):
        hs = int(offsets[l + 1] - offsets[l])
        rows, wr, valid, _ = grid_corners(x, int(resolutions[l]), hs, vxl)
        tab = emb[int(offsets[l]): int(offsets[l]) + hs]
        if ste_binary:
            tab = np.where(tab >= 0, f32(1), f32(-1))
        acc = np.zeros((N, F), f32)
        for c in range(rows.shape[1]):
            e = tab[rows[:, c]]
            acc = np.where(valid[:, c, None], fma32(wr[:, c, None], e, acc), acc)
        out[l] = acc
    return out


def grid_entry_counts(x, offsets, resolutions):
    """Number of (sample, corner) contributions each table ROW receives (all levels) — the `n` of the
    float32 summation-error bound (n - 1) * eps * sum|terms|."""
    cnt = np.zeros(int(offsets[-1]), np.int64)
    for l in range(len(resolutions)):
        o0 = int(offsets[l])
        hs = int(offsets[l + 1]) - o0
        rows, _, valid, _ = grid_corners(x, int(resolutions[l]), hs)
        cnt[o0: o0 + hs] += np.bincount(rows[valid].astype(np.int64), minlength=hs)
    return cnt


def grid_encode_backward64(grad, x, emb, offsets, resolutions, ste_binary=False):
    """(float64 sums of the float32 contributions w*wn_re*g, float64 sums of their magnitudes,
    number of contributions) per table entry — gridencoder.cu:556-581; the order-free reference any
    scatter implementation must agree with up to float32 summation error."""
    emb = np.asarray(emb, f32)
    grad = np.asarray(grad, f32)
    rows_total, F = emb.shape
    acc = np.zeros((rows_total, F), np.float64)
    mag = np.zeros((rows_total, F), np.float64)
    cnt = np.zeros(rows_total, np.int64)
    for l in range(len(resolutions)):
        o0 = int(offsets[l])
        hs = int(offsets[l + 1]) - o0
        rows, wr, valid, _ = grid_corners(x, int(resolutions[l]), hs)
        for c in range(rows.shape[1]):
            v = valid[:, c]
            r = rows[v, c].astype(np.int64)
            contrib = (wr[v, c, None] * grad[l][v]).astype(f32)          # float32 product, as in the kernel
            cnt[o0: o0 + hs] += np.bincount(r, minlength=hs)
            for ch in range(F):
                acc[o0: o0 + hs, ch] += np.bincount(r, weights=contrib[:, ch].astype(np.float64), minlength=hs)
                mag[o0: o0 + hs, ch] += np.bincount(r, weights=np.abs(contrib[:, ch]).astype(np.float64), minlength=hs)
    if ste_binary:
        keep = (emb >= -1) & (emb <= 1)
        acc *= keep
        mag *= keep
    return acc, mag, cnt


def cnt_np_embed(points, emb, R, hashmap_size, axis):
    """[R-2, R-2, F, 2] vote counts.  gridencoder.cu:881-914."""
    p = np.asarray(points).astype(np.int64)
    F = emb.shape[1]
    q = p.astype(np.uint32)
    row = grid_index(q, hashmap_size, R).astype(np.int64)
    ok = np.all((p > 0) & (p < R - 1), axis=1)
    a, b = {0: (0, 1), 1: (0, 2), 2: (1, 2)}[axis]
    S = R - 2
    pix = (p[ok, a] - 1) * S + (p[ok, b] - 1)
    vals = emb[row[ok]]
    out = np.zeros((S * S, F, 2), f32)
    for ch in range(F):
        pos = vals[:, ch] > 0.9
        out[:, ch, 0] = np.bincount(pix[pos], minlength=S * S)
        out[:, ch, 1] = np.bincount(pix[~pos], minlength=S * S)
    return out.reshape(S, S, F, 2)


# ----------------------------------------------------------------------------------------------
# aligner
# ----------------------------------------------------------------------------------------------
def query_mask_3D(points, vxl, resolution):
    """(mask int16 [N], overlap int32 [N]); `resolution` scalar or per-point array (the _qlist
    variant).  aligner_kernel.cu:161-242 / :244-330."""
    p = np.asarray(points).astype(np.int64)
    N = p.shape[0]
    Rb = vxl.shape[0]
    Rb_re = f32(1.0 / float(f32(Rb)))
    res = np.broadcast_to(np.asarray(resolution), (N,)).astype(f32)
    s = (1.0 / (res.astype(np.float64) - 2.0)).astype(f32)                                  # [N]
    pn = ((p.astype(f32).astype(np.float64) - 0.5) * s[:, None].astype(np.float64)).astype(f32)   # [N,3]
    lo_f, hi_f = pn - s[:, None], pn + s[:, None]
    lo = np.clip(lo_f * f32(Rb), f32(0), f32(Rb - 1)).astype(np.int64)
    hi = np.clip(hi_f * f32(Rb), f32(0), f32(Rb - 1)).astype(np.int64)
    span = int((hi - lo).max()) + 1 if N else 0
    m = np.zeros(N, bool)
    area = np.zeros(N, f32)

    def overlap(idx, d):
        right = np.minimum(fma32(idx.astype(f32), Rb_re, Rb_re), hi_f[:, d])
        left = np.maximum(idx.astype(f32) * Rb_re, lo_f[:, d])
        return right - left

    for da in range(span):
        ia = lo[:, 0] + da
        va = ia <= hi[:, 0]
        oa = overlap(ia, 0)
        for db in range(span):
            ib = lo[:, 1] + db
            vb = va & (ib <= hi[:, 1])
            ob = overlap(ib, 1)
            for dc in range(span):
                ic = lo[:, 2] + dc
                v = vb & (ic <= hi[:, 2])
                if not v.any():
                    continue
                oc = overlap(ic, 2)
                cell = np.zeros(N, bool)
                cell[v] = vxl[ia[v], ib[v], ic[v]]
                m |= cell
                area = np.where(cell, fma32(oa * ob, oc, area), area)
    area = area * f32(Rb) * f32(Rb) * f32(Rb)
    return m.astype(np.int16), (area * f32(1000)).astype(np.int32)


def align_and_pack_forward(feat, cnt, V=0.0):
    """packed[i, j] = feat[cumsum[i] + j] for j < cnt[i], else V.  aligner_kernel.cu:421-434."""
    cnt = np.asarray(cnt, np.int64)
    starts = np.cumsum(cnt) - cnt
    M = int(cnt.max()) if cnt.size else 0
    j = np.arange(M)[None, :]
    take = j < cnt[:, None]
    src = np.where(take, starts[:, None] + j, 0)
    out = np.where(take[..., None], feat[src], f32(V)).astype(f32)
    return out


# ----------------------------------------------------------------------------------------------
# occupancy-grid marcher, one grid (CNC never uses more)
# ----------------------------------------------------------------------------------------------
def traverse_grids(rays_o, rays_d, binaries, aabb, t_min, t_max, hits, near, far, step_size, limit=-1):
    """All rays advance together, one DDA cell per outer iteration; a ray that is done drops out.
    Returns dict(counts [n_rays], ray [S], t_mid [S], t_left [S], t_right [S], first [S] (the sample
    opens a new interval run), terminate [n_rays]).  cone_angle = 0 (CNC), step_size > 0."""
    o = np.asarray(rays_o, f32)
    d = np.asarray(rays_d, f32)
    n = o.shape[0]
    res = np.array(binaries.shape[-3:], np.int64)
    occ = np.asarray(binaries).reshape(-1, *res)[0]
    bmin, bmax = np.asarray(aabb[:3], f32), np.asarray(aabb[3:], f32)
    near = np.broadcast_to(np.asarray(near, f32), (n,)).copy()
    far = np.broadcast_to(np.asarray(far, f32), (n,)).copy()
    eps = f32(1e-6)
    dt = np.clip(f32(0), f32(step_size), f32(1e10))             # clamp(t*0, step, 1e10)
    half = dt * f32(0.5)

    tmin = np.maximum(np.asarray(t_min, f32), near)
    tmax = np.minimum(np.asarray(t_max, f32), far)
    alive = np.asarray(hits, bool) & ~(tmin >= tmax)
    t_last = near.copy()

    def advance(tl, target, mask):
        """while !(t_last + dt/2 >= target) t_last += dt  — repeated float32 adds, per lane."""
        tl = tl.copy()
        todo = mask & ~(tl + half >= target)
        while todo.any():
            tl[todo] = tl[todo] + dt
            todo = mask & ~(tl + half >= target)
        return tl

    t_last = advance(t_last, tmin, alive)
    inv = (f32(1) / d).astype(f32)
    voxel = (bmax - bmin) / res.astype(f32)
    rs = fma32(d, (tmin + eps)[:, None], o)
    re = fma32(d, (tmax - eps)[:, None], o)
    cur = np.clip((((rs - bmin) / (bmax - bmin)) * res.astype(f32)).astype(np.int32), 0, res - 1).astype(np.int64)
    fin = np.clip((((re - bmin) / (bmax - bmin)) * res.astype(f32)).astype(np.int32), 0, res - 1).astype(np.int64)
    start_idx = cur + (d > 0)
    txyz = fma32(bmin + fma32(start_idx.astype(f32), voxel, -rs), inv, tmin[:, None])
    zero = d == 0
    tdist = np.where(zero, tmax[:, None], txyz).astype(f32)
    sgn = np.where(zero, f32(0), np.where(d > 0, f32(1), f32(-1))).astype(f32)
    step_i = sgn.astype(np.int64)
    delta = np.where(zero, tmax[:, None], voxel * inv * sgn).astype(f32)
    over = fin + step_i

    counts = np.zeros(n, np.int64)
    continuous = np.zeros(n, bool)
    rec = {k: [] for k in ("ray", "mid", "left", "right", "first")}
    active = alive.copy()
    if limit > 0:
        active &= counts < limit
    while active.any():
        a = np.nonzero(active)[0]
        t_trav = np.minimum(np.minimum(tdist[a, 0], np.minimum(tdist[a, 1], tdist[a, 2])), tmax[a])
        filled = occ[cur[a, 0], cur[a, 1], cur[a, 2]]
        # empty cells: march past them
        e = a[~filled]
        if e.size:
            m = np.zeros(n, bool)
            m[e] = True
            tt = np.zeros(n, f32)
            tt[a] = t_trav
            t_last = advance(t_last, tt, m)
            continuous[e] = False
        # occupied cells: emit samples until the mid point passes t_trav
        s_idx = a[filled]
        s_trav = t_trav[filled]
        going = np.ones(s_idx.size, bool)
        if limit > 0:
            going &= counts[s_idx] < limit
        while going.any():
            k = s_idx[going]
            tv = s_trav[going]
            stop = t_last[k] + half >= tv
            kk, tvv = k[~stop], tv[~stop]
            t_next = t_last[kk] + dt
            rec["ray"].append(kk)
            rec["mid"].append((t_next + t_last[kk]) * f32(0.5))
            rec["left"].append(t_last[kk].copy())
            rec["right"].append(t_next)
            rec["first"].append(~continuous[kk])
            counts[kk] += 1
            continuous[kk] = True
            t_last[kk] = t_next
            nxt = np.zeros(s_idx.size, bool)
            pos = np.nonzero(going)[0][~stop]
            cont = ~(t_next >= tvv)
            if limit > 0:
                cont &= counts[kk] < limit
            nxt[pos[cont]] = True
            going = nxt
        # single_traversal: strict < picks x, then y, else z
        tx, ty, tz = tdist[a, 0], tdist[a, 1], tdist[a, 2]
        ax = np.where((tx < ty) & (tx < tz), 0, np.where(ty < tz, 1, 2))
        cur[a, ax] += step_i[a, ax]
        tdist[a, ax] = tdist[a, ax] + delta[a, ax]
        out = cur[a, ax] == over[a, ax]
        active[a[out]] = False
        if limit > 0:
            active &= counts < limit
    if rec["ray"]:
        ray = np.concatenate(rec["ray"])
        order = np.argsort(ray, kind="stable")
        cat = {k: np.concatenate(v)[order] for k, v in rec.items()}
    else:
        cat = {k: np.zeros(0, f32) for k in rec}
        cat["ray"] = np.zeros(0, np.int64)
    return dict(counts=counts, ray=cat["ray"], t_mid=cat["mid"].astype(f32), t_left=cat["left"].astype(f32),
                t_right=cat["right"].astype(f32), first=cat["first"].astype(bool), terminate=t_last)
