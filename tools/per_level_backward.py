"""Scratch: cost of ONE level of the backward on each of the three paths (run-aggregated atomics, cross-ray merge kernel,
bin + owner), marched-ray samples of the bench scene, 2^20 samples, 16L x 2^19 x F8."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_binned import marched, timeit, emb, o_t, r_t, clip, be, F, dev
from cnc_amd.synthetic import RES_16L
N = 1 << 20
x = marched(N); N = x.shape[0]
g = torch.randn((16, N, F), device=dev)
ge = torch.zeros_like(emb)
print(f"N = {N}")
print("level   R     atomics   merge    binned   (ms, one level alone)")
for l in range(16):
    gl = g[l:l + 1].contiguous()
    o, r = o_t[l:l + 2], r_t[l:l + 1]
    rows = int(o_t[l + 1] - o_t[l])
    a = timeit(lambda: be.grid_encode_backward(gl, x, emb, o, r, ge, N, 3, F, 1, 0, 128, None, None, None, None, ste_binary=True, ste_clip_count=clip))
    m = timeit(lambda: be.grid_encode_backward(gl, x, emb, o, r, ge, N, 3, F, 1, 0, 128, None, None, None, None, ste_binary=True, ste_clip_count=clip, interleave_levels=True))
    b = float("nan")
    if (1 << 16) <= rows <= (1 << 20):
        b = timeit(lambda: be.grid_encode_backward(gl, x, emb, o, r, ge, N, 3, F, 1, 0, 128, None, None, None, None, ste_binary=True, ste_clip_count=clip, binned=(1, rows), overlap_streams=False))
    print(f"{l:3d} {RES_16L[l]:6d}   {a:7.3f}  {m:7.3f}  {b:7.3f}")
