import torch, time
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for N in (1<<18, 1<<20):
  for (K, M) in ((255,160),(160,80),(95,160),(160,160),(160,3),(25,32),(32,32),(32,8)):
    X = torch.randn(N, K, device=dev); dY = torch.randn(N, M, device=dev); W = torch.randn(M, K, device=dev)
    fl = 2*N*K*M
    a = t(lambda: X @ W.t())                      # forward
    b = t(lambda: dY @ W)                         # dX
    c = t(lambda: dY.t() @ X)                     # dW as autograd does
    outs = {}
    for S in (16, 64, 256):
        if N % S: continue
        d = t(lambda: torch.bmm(dY.view(S, N//S, M).transpose(1,2), X.view(S, N//S, K)).sum(0))
        outs[S] = d
    torch.backends.cuda.matmul.allow_tf32 = False
    print(f"N=2^{N.bit_length()-1} K={K:3d} M={M:3d}: fwd {a:.3f} ms ({fl/a/1e9:.1f} TF)  dX {b:.3f} ({fl/b/1e9:.1f} TF)  dW {c:.3f} ({fl/c/1e9:.1f} TF)  " + "  ".join(f"splitK{S} {v:.3f} ({fl/v/1e9:.1f} TF)" for S,v in outs.items()))
