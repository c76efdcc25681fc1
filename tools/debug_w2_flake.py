import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_field_fused import _field, _inputs, CONFIGS
cuda = torch.device("cuda")
f = _field(cuda, CONFIGS["f8_full"], seed=3)
x, d = _inputs(cuda, 70001, seed=70001)
with torch.no_grad():
    f.fused_field = False
    ref, _ = f(x, d)
    f.fused_field = True
    outs = [f(x, d)[0].clone() for _ in range(int(os.environ.get('REPS', '30')))]
torch.cuda.synchronize()
for k, o in enumerate(outs):
    bad = ((o - ref).abs().max(dim=1).values > 1e-4).nonzero().flatten()
    if bad.numel():
        rows = bad.tolist()
        print(k, "bad rows:", len(rows), "tiles:", sorted(set(r // 32 for r in rows))[:10], "rows in tile:", sorted(set(r % 32 for r in rows)),
              "max err", float((o - ref).abs().max()))
print("done")
