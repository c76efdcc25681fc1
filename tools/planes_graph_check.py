"""Scratch: the planes' graph against the op-by-op entropy pass on the same state: gradients and bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd.trainer import TrainConfig, Trainer
from cnc_amd._planes_graph import PlanesGraph

toy = "--toy" in sys.argv
if toy:
    cfg = TrainConfig(lmbda=2e-3, Pg_level=5, Pg_level_2D=3, log2_hashmap_size=12, log2_hashmap_size_2D=9,
                      sample_num=3000, max_context_layer_num=3, n_features=2, n_neurons=32,
                      resolutions_list=(10, 14, 18, 26, 34), resolutions_list_2D=(18, 34, 66),
                      skip_levels_3D=(0, 1, 2), skip_levels_2D=(0,), max_steps=150, init_batch_size=512,
                      target_sample_batch_size=1 << 14, grid_resolution=16, render_step_size=2e-2,
                      milestones=(100, 130), warmup_iters=20, test_views=2, image_size=48, out_dir="/tmp/bits_t", log_every=50)
else:
    cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=2000, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
pg = tr.planes_graph
LAST = 20 if "--graphed" in sys.argv else 37
if "--graphed" not in sys.argv:
    tr.planes_graph = None
for step in range(LAST):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
step = LAST
print("before the check: captures", pg.captures, "replays", pg.replays)
params = list(tr.field.parameters()) + list(tr.context.parameters())
names = [n for n, _ in tr.field.named_parameters()] + ["ctx." + n for n, _ in tr.context.named_parameters()]

def run(use_graph, reps=1):
    tr.planes_graph = pg if use_graph else None
    out = None
    for _ in range(reps):
        for p in params:
            p.grad = None
        for s in (tr.sink_render, tr.sink_ctx):
            s.zero()
        for enc in tr.field.mlp_base._encoders():
            enc._bit_plane(enc.params)
        torch.manual_seed(5)
        main = torch.cuda.current_stream()
        tr._ensure_planes_graph(step, None)
        bpp, mb, done, _ = tr._context_pass(step, main.record_event())
        main.wait_event(done)
        tr.sink_ctx.flush()
        if tr._planes_replayed:
            tr.planes_graph.flush()
        torch.cuda.synchronize()
        out = (float(bpp), float(mb), [None if p.grad is None else p.grad.clone() for p in params])
    return out

a = run(False)
b = run(True)
c = run(True, reps=3)       # replays of the same graph
print("bpp", a[0], b[0], c[0], " MB", a[1], b[1], c[1], " captures", pg.captures, "replays", pg.replays)
for which, r in (("graph", b), ("graph x3", c)):
    worst = 0.0
    for n, x, y in zip(names, a[2], r[2]):
        if (x is None) != (y is None):
            print(which, n, "None mismatch", x is None, y is None)
            continue
        if x is None:
            continue
        d, s = float((x - y).abs().max()), float(x.abs().max())
        if not (d <= 1e-5 * max(s, 1e-30)):
            print(which, n, "diff", d, "scale", s, "finite", bool(torch.isfinite(y).all()))
        worst = max(worst, d / max(s, 1e-30))
    print(which, "worst relative difference", worst)
print("pairs", [(tuple(g.shape), float(g.abs().max()), rows) for _, g, rows in pg.pairs])
print("sink flags", tr.sink_ctx._tables_used, pg._tables_used, "arena max", [float(v.abs().max()) for v in tr.sink_ctx.table_views])
# eager decomposition: autograd part vs sink part
tr.planes_graph = None
for p in params:
    p.grad = None
for s in (tr.sink_render, tr.sink_ctx):
    s.zero()
torch.manual_seed(5)
main = torch.cuda.current_stream()
bpp, mb, done, _ = tr._context_pass(step, main.record_event())
main.wait_event(done)
torch.cuda.synchronize()
enc = tr.field.mlp_base._encoders()
print("eager autograd part", [None if e.params.grad is None else float(e.params.grad.abs().max()) for e in enc])
print("eager sink part", [float(v.abs().max()) for v in tr.sink_ctx.table_views])
# replay stress: the same graph, nothing else running, many times
import math
tr.planes_graph = pg
bad = 0
vals = set()
with torch.cuda.stream(tr.ctx_stream_2D):
    for i in range(300):
        tr.sink_ctx.zero()
        pg.graph.replay()
        torch.cuda.synchronize()
        v = float(pg.bits)
        vals.add(v)
        if not math.isfinite(v) or not all(bool(torch.isfinite(g).all()) for _, g, _ in pg.pairs):
            bad += 1
print("300 replays alone: non-finite", bad, "distinct bits values", len(vals), sorted(vals)[:3])
# ... with the render pass's kernels running next to them on another stream
x = torch.rand(1 << 18, 3, device="cuda") * 2 - 1
d = torch.nn.functional.normalize(torch.randn(1 << 18, 3, device="cuda"), dim=-1)
bad = 0
for i in range(200):
    tr.sink_ctx.zero()
    with torch.cuda.stream(tr.ctx_stream_2D):
        tr.ctx_stream_2D.wait_stream(torch.cuda.current_stream())
        pg.graph.replay()
    with torch.no_grad():
        tr.field(x, d)
        tmp = torch.empty(1 << 24, device="cuda").normal_()      # allocator traffic on the main stream
        del tmp
    torch.cuda.synchronize()
    v = float(pg.bits)
    if not math.isfinite(v):
        bad += 1
print("200 replays next to other work: non-finite", bad)
