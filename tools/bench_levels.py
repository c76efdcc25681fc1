"""Scratch: cumulative cost of the atomic backward kernel over the first k levels (marched rays)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_binned import marched, timeit, emb, o_t, r_t, clip, be, F, dev
N = 1 << 20
x = marched(N); N = x.shape[0]
g = torch.randn((16, N, F), device=dev)
ge = torch.zeros_like(emb)
prev = 0
for k in range(1, 17):
    ms = timeit(lambda: be.grid_encode_backward(g, x, emb, o_t, r_t, ge, N, 3, F, k, 0, 128, None, None, None, None, ste_binary=True, ste_clip_count=clip))
    print(f"levels 0..{k-1}: {ms:.3f} ms  (+{ms-prev:.3f})")
    prev = ms
