"""Scratch: A/B of a Trainer switch inside ONE process (boxes differ by more than the effects measured here):
alternating blocks of steps with the switch off / on, wall time per step of each block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnc_amd.trainer import TrainConfig, Trainer

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=2000, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
what = sys.argv[1] if len(sys.argv) > 1 else "bucket"

def switch(on):
    if what == "bucket":
        tr.field.row_bucket_min = 4096 if on else 1 << 30
    elif what == "ctxthread":
        tr.ctx_thread = on
    elif what == "ctxstream":
        if not hasattr(tr, "_cs"):
            tr._cs = tr.ctx_stream
        tr.ctx_stream = tr._cs if on else None
        tr.ctx_thread = False
    elif what == "chain":
        tr.field.fused_chain, tr.field._chain_supported = on, None
    elif what == "stream2d":
        if not hasattr(tr, "_s2"):
            tr._s2 = tr.ctx_stream_2D
        tr.ctx_stream_2D = tr._s2 if on else None
    elif what.startswith("switch"):         # switch0.0005: the interpreter's thread switch interval (s), on = that value
        sys.setswitchinterval(float(what[6:]) if on else 0.005)
    elif what == "pgraph":
        if not hasattr(tr, "_pg"):
            tr._pg = tr.planes_graph
        tr.planes_graph = tr._pg if on else None
    elif what == "premarch":
        tr.premarch = on
    elif what == "prefetch":
        tr.prefetch = on
    elif what == "wgrad":
        tr.field.fused_wgrad = on
    elif what == "train":
        tr.field.fused_train = on
    elif what == "cells":                   # the context pass's scatters: cell-merging kernel / run kernel
        from cnc_amd import gridencoder
        gridencoder._CELL_MERGE = on
        if tr.planes_graph is not None:
            tr.planes_graph.drop()
    elif what == "tableadam":               # the tables' update: pieces summed inside cnc_table_adam / flushed into .grad
        tr.fused_table_adam = on
    elif what == "windows":                 # the sampler's depth windows: sample counts left on the device / read per window
        tr.estimator._COUNTED_WINDOWS = on
    elif what == "vbits":
        for e in tr.field.mlp_base._encoders():
            e.vertex_bits = on
    else:
        os.environ[what] = "1" if on else "0"

step = 0
for _ in range(240):
    tr.train_step(step, want_stats=False); step += 1
res = {False: [], True: []}
for rep in range(8):
    for on in (False, True):
        switch(on)
        for _ in range(8):
            tr.train_step(step, want_stats=False); step += 1
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            tr.train_step(step, want_stats=False); step += 1
        torch.cuda.synchronize()
        res[on].append((time.perf_counter() - t0) / 40 * 1e3)
for on in (False, True):
    r = sorted(res[on])
    print(f"{what} {'on ' if on else 'off'}: median {r[len(r)//2]:.2f} ms  min {r[0]:.2f}  max {r[-1]:.2f}   {['%.2f' % x for x in res[on]]}")
