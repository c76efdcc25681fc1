import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from cnc_amd.mlp import FusedMLPForward, Linear
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for dims in ((255,160,80),(95,160,160,3)):
    layers=[]
    for i in range(len(dims)-1):
        layers.append(Linear(dims[i],dims[i+1]))
        if i < len(dims)-2: layers.append(nn.ReLU(inplace=True))
    seq=nn.Sequential(*layers).to(dev); fused=FusedMLPForward(seq); fused32=FusedMLPForward(seq, rows_per_wave=32)
    fl = 2*sum(dims[i]*dims[i+1] for i in range(len(dims)-1))
    for N in (1<<16, 1<<18, 1<<20, 1<<22):
        xb=torch.randn(N,(dims[0]+3)//4*4,device=dev); x=xb[:,:dims[0]]
        from cnc_amd import _lib
        with torch.no_grad():
            a=t(lambda: seq(x))
            b=t(lambda: fused(x))
            d=t(lambda: fused32(x))
            err=(fused32(x)-seq(x)).abs().max().item()
        # (the one-tile / 96-row / LDS-shared-weights variants of rounds 2-3 were measured with this probe and removed in
        # round 6 together with the process-wide switch that selected them: docs/engineering_log.md 4.6)
        print(f"{dims} N=2^{N.bit_length()-1}: torch {a:.3f} ms ({fl*N/a/1e9:.1f} TF)   1-wave {b:.3f} ms ({fl*N/b/1e9:.1f} TF)   64-row-L1 {d:.3f} ms ({fl*N/d/1e9:.1f} TF)  max err {err:.1e}")
