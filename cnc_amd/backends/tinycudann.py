"""`tinycudann` stand-in for the one thing the reference takes from it (examples/radiance_fields/ngp.py:412-425):

    tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "Composite", "nested": [
        {"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": 4}]})

— a module with `n_input_dims`, `n_output_dims` (16) that maps directions given in [0, 1]^3 to the 16 real spherical
harmonics of degree < 4.  tiny-cuda-nn is an unpinned third-party git install (README.md:56) and absent here, so its
numerics are PARITY UNPINNED; what is reproduced is its documented behaviour: inputs are mapped back to [-1, 1], the
polynomial form is evaluated in fp32, and the output tensor is half precision unless `dtype=torch.float32` is asked
for (tcnn's default output precision on GPUs with fp16 support — the reference then `cat`s it with fp32 features,
which promotes the rounded values back to fp32).  Every other encoding / network type raises.
"""
from __future__ import annotations

import torch

from ..field import SHEncoding


def _spherical_harmonics_of(config: dict, n_input_dims: int):
    otype = str(config.get("otype", "")).lower()
    if otype == "composite":
        nested = config.get("nested", [])
        if len(nested) != 1:
            raise NotImplementedError("tinycudann stand-in: only a Composite of ONE SphericalHarmonics encoding is built")
        if int(nested[0].get("n_dims_to_encode", n_input_dims)) != n_input_dims:
            raise NotImplementedError("tinycudann stand-in: the nested encoding must cover all input dimensions")
        return _spherical_harmonics_of(nested[0], n_input_dims)
    if otype != "sphericalharmonics":
        raise NotImplementedError(f"tinycudann stand-in: encoding otype {config.get('otype')!r} is outside the CNC path")
    if int(config.get("degree", 4)) != 4 or n_input_dims != 3:
        raise NotImplementedError("tinycudann stand-in: SphericalHarmonics is built for degree 4 on 3 input dims")
    return 16


class Encoding(torch.nn.Module):
    def __init__(self, n_input_dims: int, encoding_config: dict, seed: int = 1337, dtype=None):
        super().__init__()
        self.n_input_dims = int(n_input_dims)
        self.n_output_dims = _spherical_harmonics_of(dict(encoding_config), self.n_input_dims)
        self.encoding_config = encoding_config
        self.seed = seed
        self.dtype = torch.float16 if dtype is None else dtype
        self._sh = SHEncoding(fp16_round=False)
        self.register_parameter("params", torch.nn.Parameter(torch.zeros(0), requires_grad=False))   # tcnn modules own one

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("tinycudann: input must be a CUDA tensor")
        return self._sh(x.float()).to(self.dtype)


class Network(torch.nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("tinycudann stand-in: fully fused networks are outside the CNC path (its MLPs are nn.Linear)")


NetworkWithInputEncoding = Network
