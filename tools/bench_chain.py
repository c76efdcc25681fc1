#!/usr/bin/env python
"""Gradient pass of the radiance field at N samples: forward + backward through `_FieldChain` (one kernel for the
input-gradient chain) against the layer-by-layer autograd path; per-kernel times from the torch profiler.
    python tools/bench_chain.py [--n 262144] [--profile]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 18)
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    from cnc_amd.field import NGPRadianceField_mygrid_2D3D
    dev = torch.device("cuda")
    torch.manual_seed(1)
    f = NGPRadianceField_mygrid_2D3D(aabb=[-1.5] * 3 + [1.5] * 3, n_features_per_level=8, n_neurons=160,
                                     resolutions_list=(18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514),
                                     log2_hashmap_size=19, resolutions_list_2D=(130, 258, 514, 1026),
                                     log2_hashmap_size_2D=17).to(dev)
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.rand(a.n, 3, device=dev, generator=g) * 3.0 - 1.5
    d = torch.nn.functional.normalize(torch.randn(a.n, 3, device=dev, generator=g), dim=-1)
    wr = torch.randn(a.n, 3, device=dev, generator=g) * 1e-4
    wd = torch.randn(a.n, 1, device=dev, generator=g) * 1e-5

    def step():
        for p in f.parameters():
            p.grad = None
        rgb, den = f(x, d)
        ((rgb * wr).sum() + (den * wd).sum()).backward()

    # (chain kernel, saving fused forward, weight-gradient kernel)
    for chain, train, wgrad in ((True, True, True), (True, True, False), (True, False, True), (True, False, False), (False, False, False)):
        f.fused_chain, f.fused_train, f.fused_wgrad, f._chain_supported = chain, train, wgrad, None
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); step(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"chain={chain} train={train} wgrad={wgrad}: {sorted(ts)[len(ts) // 2]:.3f} ms per forward + backward at N = {a.n}", flush=True)
        if a.profile and (chain, train, wgrad) == (True, True, True):
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
            print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=22, max_name_column_width=70))


if __name__ == "__main__":
    main()
