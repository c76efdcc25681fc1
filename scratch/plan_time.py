import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from cnc_amd.backends import gridencoder_backend as ge
from cnc_amd.synthetic import ball_binaries
dev = torch.device("cuda:0")
occ = ball_binaries(128, device=dev)[0].bool()
t = 4
m = occ
for axis in range(3):
    up = m.repeat_interleave(t, dim=axis); n = up.shape[axis]
    shape = list(up.shape); shape[axis] = n + 2
    out = torch.zeros(shape, dtype=torch.bool, device=dev)
    for s in range(3): out.narrow(axis, s, n).logical_or_(up)
    m = out
verts = torch.nonzero(m).to(torch.int16).contiguous()
print("vertices", verts.shape)
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("VotePlan ms", timeit(lambda: ge.VotePlan(verts, 514, 2 ** 19)))
N = verts.shape[0]
k = torch.randint(0, 512 * 512, (N,), device=dev, dtype=torch.int32)
print("sort i32 stable", timeit(lambda: torch.sort(k, stable=True)))
print("sort i32", timeit(lambda: torch.sort(k)))
o = torch.sort(k, stable=True)[1]
print("gather", timeit(lambda: k[o]))
ks = torch.sort(k)[0]
b = torch.arange(512 * 512 + 1, device=dev, dtype=torch.int32)
print("searchsorted", timeit(lambda: torch.searchsorted(ks, b)))
k2 = torch.randint(0, 2 ** 17, (300000,), device=dev, dtype=torch.int32)
print("sort 300k i32 stable", timeit(lambda: torch.sort(k2, stable=True)))
k3 = k2.long()
print("sort 300k i64 stable", timeit(lambda: torch.sort(k3, stable=True)))
ks2 = torch.sort(k2)[0]
print("unique_consecutive 300k", timeit(lambda: torch.unique_consecutive(ks2, return_counts=True)))
