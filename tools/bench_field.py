#!/usr/bin/env python
"""Time the gradient-free radiance field at N samples: fused kernel vs chain (bench.py's `field_entries`, alone).
    python tools/bench_field.py [--n 1048576] [--only fused|chain] [--mode density|rgb|both] [--reps 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--mode", default="both")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--features", type=int, default=8)
    a = ap.parse_args()
    from cnc_amd.field import NGPRadianceField_mygrid_2D3D
    dev = torch.device("cuda")
    torch.manual_seed(1)
    f = NGPRadianceField_mygrid_2D3D(aabb=[-1.5] * 3 + [1.5] * 3, n_features_per_level=a.features, n_neurons=160,
                                     resolutions_list=(18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514),
                                     log2_hashmap_size=19, resolutions_list_2D=(130, 258, 514, 1026),
                                     log2_hashmap_size_2D=17).to(dev)
    with torch.no_grad():
        for e in f.mlp_base._encoders():
            e.params.uniform_(-1, 1)
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.rand(a.n, 3, device=dev, generator=g) * 3.0 - 1.5
    d = torch.nn.functional.normalize(torch.randn(a.n, 3, device=dev, generator=g), dim=-1)

    def timed(fn):
        ts = []
        for it in range(a.reps + 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            if it >= 3:
                ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    with torch.no_grad():
        for name, fused in (("fused", True), ("chain", False)):
            if a.only and a.only != name:
                continue
            f.fused_field = fused
            if a.mode in ("density", "both"):
                print(f"{name:6s} density   {timed(lambda: f.query_density(x)):8.4f} ms / {a.n} samples", flush=True)
            if a.mode in ("rgb", "both"):
                print(f"{name:6s} rgb       {timed(lambda: f(x, d)):8.4f} ms / {a.n} samples", flush=True)


if __name__ == "__main__":
    main()
