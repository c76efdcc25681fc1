"""The encoder backward calls of one training step (scratch/bwd_calls.npz from tools/dump_bwd_calls.py), counted on the CPU:
per call and level slot the corner rows a per-corner scatter would send, what merging runs of equal rows along the list
leaves, and the DISTINCT (level, row) targets per block of 256 / 1,024 / 4,096 consecutive points — the ceiling of what any
block-local merge can save (profiles/r06_step_backward_calls.md).     python tools/count_bwd_calls.py [calls.npz]"""
import numpy as np, sys
z = np.load(sys.argv[1] if len(sys.argv) > 1 else 'scratch/bwd_calls.npz')
PR = np.array([1, 2654435761, 805459861], dtype=np.uint64)
def rows_of(q, hs, R, D):
    # q [n, D] uint64
    stride = 1; idx = np.zeros(len(q), np.uint64); 
    for d in range(D):
        if stride <= hs:
            idx += q[:, d] * np.uint64(stride); stride *= R
    if stride > hs:
        idx = np.zeros(len(q), np.uint64)
        for d in range(D):
            idx ^= (q[:, d] * PR[d]) & np.uint64(0xFFFFFFFF)
    return (idx % np.uint64(hs)).astype(np.int64)
ncalls = len([k for k in z.files if k.endswith('_N')])
for ci in range(ncalls):
    g = lambda k: z[f'c{ci}_{k}'] if f'c{ci}_{k}' in z.files else None
    N, D, L = int(g('N')), int(g('D')), int(g('L'))
    x = g('inputs'); offs = g('offsets'); res = g('resolutions'); mli = g('mli'); nz = g('nz')
    print(f"== call {ci}: N={N} D={D} L={L} masked={g('vxl') is not None} perpoint={mli is not None} res={res.tolist()} nzfrac={nz.mean():.3f}")
    tot = dict(corner=0, runs=0, d256=0, d1024=0, d4096=0)
    for s in range(L):
        lvl = (mli + s) if mli is not None else np.full(N, s)
        keep = nz[:, s] > 0
        inr = np.all((x >= 0) & (x <= 1), axis=1)
        keep &= inr
        R = res[lvl].astype(np.int64); hs = (offs[lvl + 1] - offs[lvl]).astype(np.int64)
        p = (x * (R - 2)[:, None].astype(np.float32)).astype(np.float32) + np.float32(0.5)
        cell = np.floor(p).astype(np.int64)
        # cell key incl. level
        key = lvl.astype(np.int64)
        for d in range(D): key = key * 4096 + cell[:, d]
        key = np.where(keep, key, -1)
        # runs
        head = np.ones(N, bool); head[1:] = key[1:] != key[:-1]
        head &= key >= 0
        C = 1 << D
        nk = int(keep.sum())
        out = [nk * C, int(head.sum()) * C]
        for B in (256, 1024, 4096):
            blk = np.arange(N) // B
            kk = key[keep] ; bb = blk[keep]
            u = np.unique(bb * (1 << 48) + kk)
            out.append(len(u) * C)
        # distinct rows over the whole call (compulsory)
        ulv = np.unique(lvl)
        drows = 0
        for l in ulv:
            m = keep & (lvl == l)
            if not m.any(): continue
            Rl = int(res[l]); hsl = int(offs[l+1]-offs[l])
            cs = np.unique(cell[m], axis=0)
            rr = []
            for c in range(C):
                q = cs.copy()
                for d in range(D):
                    if (c >> d) & 1: q[:, d] = np.minimum(q[:, d] + 1, Rl - 1)
                rr.append(rows_of(q.astype(np.uint64), hsl, Rl, D))
            drows += len(np.unique(np.concatenate(rr)))
        print(f"  slot {s}: pts {nk:8d} corner-rows {out[0]/1e6:7.2f}M  run-merged {out[1]/1e6:7.2f}M  distinct/256 {out[2]/1e6:7.2f}M /1024 {out[3]/1e6:7.2f}M /4096 {out[4]/1e6:7.2f}M  distinct rows in call {drows/1e6:6.2f}M" + (f"  levels {np.bincount(lvl[keep], minlength=12).tolist()}" if mli is not None else f" R={res[s]}"))
        for k, v in zip(tot, out): tot[k] += v
    print("  TOTAL (M corner-rows):", {k: round(v/1e6, 2) for k, v in tot.items()}, " at 21G/s ->", {k: round(v/21e9*1e3, 3) for k, v in tot.items()}, "ms")
