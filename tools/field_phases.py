#!/usr/bin/env python
"""Where the two-wave fused field kernel's wave time goes (needs a build with CNC_HIP_EXTRA_FLAGS=-DCNC_W2_PROF:
    CNC_HIP_EXTRA_FLAGS=-DCNC_W2_PROF python -m cnc_amd.build --force && python tools/field_phases.py).
Shader-clock ticks between marks, summed over the waves; printed as a share of the total and as ms of one wave slot."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = ["fill (gather + features -> LDS)", "weights issued + barrier", "A reads + weights wait + MFMA issue",
         "density epilogue", "h1 -> planes", "layer 2", "geo / SH scatter", "head 1", "-> planes", "head 2", "-> planes",
         "head 3 + store"]


def main():
    from cnc_amd.field import NGPRadianceField_mygrid_2D3D
    dev = torch.device("cuda")
    torch.manual_seed(1)
    f = NGPRadianceField_mygrid_2D3D(aabb=[-1.5] * 3 + [1.5] * 3, n_features_per_level=8, n_neurons=160,
                                     resolutions_list=(18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514),
                                     log2_hashmap_size=19, resolutions_list_2D=(130, 258, 514, 1026),
                                     log2_hashmap_size_2D=17).to(dev)
    with torch.no_grad():
        for e in f.mlp_base._encoders():
            e.params.uniform_(-1, 1)
    n = 1 << 20
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.rand(n, 3, device=dev, generator=g) * 3.0 - 1.5
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
    with torch.no_grad():
        for mode in ("density", "rgb"):
            fn = (lambda: f.query_density(x)) if mode == "density" else (lambda: f(x, d))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            guard = f._field_fused._buffers["guard"]
            guard[8:].zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            words = guard[8:8 + 26].view(torch.int64).tolist()
            waves, total = words[12], sum(words[:12])
            print(f"{mode}: {e0.elapsed_time(e1):.3f} ms, {waves} waves, {total / max(waves, 1):.0f} ticks per wave")
            for k, name in enumerate(NAMES):
                if words[k]:
                    print(f"    {name:40s} {100.0 * words[k] / total:5.1f} %   {words[k] / max(waves, 1):10.0f} ticks / wave")


if __name__ == "__main__":
    main()
