"""Scratch: point-major bit-plane forward at training-batch sizes (N = 2^18, 12 levels 3-D + 4 levels 2-D, ld = 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cnc_amd.backends import gridencoder_backend as be
from cnc_amd import synthetic
dev = torch.device("cuda:0")
def grid(res, T, D):
    offs = [0]
    for R in res: offs.append(offs[-1] + int(np.ceil(min(2 ** T, R ** D) / 8) * 8))
    return torch.tensor(offs, dtype=torch.int32, device=dev), torch.tensor(res, dtype=torch.int32, device=dev), offs[-1]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
F = 8
o3, r3, rows3 = grid(list(synthetic.RES_3D_REF), 19, 3)
o2, r2, rows2 = grid(list(synthetic.RES_2D_REF), 17, 2)
emb3 = torch.sign(torch.rand((rows3, F), device=dev) * 2 - 1); bits3 = be.pack_sign_bits(emb3)
emb2 = torch.sign(torch.rand((rows2, F), device=dev) * 2 - 1); bits2 = be.pack_sign_bits(emb2)
for N in (1 << 16, 1 << 18, 1 << 20):
    x = torch.rand((N, 3), device=dev); x2 = x[:, :2].contiguous()
    feat = torch.empty((N, 256), device=dev); lm = torch.empty((12, N, F), device=dev)
    L3, L2 = len(synthetic.RES_3D_REF), len(synthetic.RES_2D_REF)
    a = timeit(lambda: be.grid_encode_forward_bits(x, bits3, o3, r3, feat, N, 3, F, L3, 128, None, None, None, out_ld=256, out_col=0))
    b = timeit(lambda: be.grid_encode_forward_bits(x, bits3, o3, r3, lm, N, 3, F, L3, 128))
    c = timeit(lambda: be.grid_encode_forward_bits(x2, bits2, o2, r2, feat, N, 2, F, L2, 128, None, None, None, out_ld=256, out_col=96))
    print(f"N=2^{N.bit_length()-1}: 3-D 12 levels point-major {a:7.1f} us, level-major {b:7.1f} us; 2-D 4 levels point-major {c:6.1f} us   (uniform points)")
