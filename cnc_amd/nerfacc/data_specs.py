"""RaySamples / RayIntervals records (reference: nerfacc/data_specs.py:12-180)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import cuda as _C


def _packed(spec):
    if spec.chunk_starts is None or spec.chunk_cnts is None:
        return None
    return torch.stack([spec.chunk_starts, spec.chunk_cnts], -1)


@dataclass
class RaySamples:
    """Samples along rays: batched `vals` (n_rays, n_samples) or flattened `vals` (all_samples,)
    with `packed_info` (n_rays, 2) = (start, count) and/or `ray_indices`."""
    vals: torch.Tensor
    packed_info: Optional[torch.Tensor] = None
    ray_indices: Optional[torch.Tensor] = None
    is_valid: Optional[torch.Tensor] = None

    def _to_cpp(self):
        spec = _C.RaySegmentsSpec()
        spec.vals = self.vals.contiguous()
        if self.packed_info is not None:
            spec.chunk_starts = self.packed_info[:, 0].contiguous()
            spec.chunk_cnts = self.packed_info[:, 1].contiguous()
        if self.ray_indices is not None:
            spec.ray_indices = self.ray_indices.contiguous()
        return spec

    @classmethod
    def _from_cpp(cls, spec):
        return cls(vals=spec.vals, packed_info=_packed(spec), ray_indices=spec.ray_indices,
                   is_valid=spec.is_valid)

    @property
    def device(self) -> torch.device:
        return self.vals.device


@dataclass
class RayIntervals:
    """Interval edges along rays; `is_left` / `is_right` flag which edges open / close a sample
    interval (an edge shared by two consecutive intervals carries both)."""
    vals: torch.Tensor
    packed_info: Optional[torch.Tensor] = None
    ray_indices: Optional[torch.Tensor] = None
    is_left: Optional[torch.Tensor] = None
    is_right: Optional[torch.Tensor] = None

    def _to_cpp(self):
        spec = _C.RaySegmentsSpec()
        spec.vals = self.vals.contiguous()
        if self.packed_info is not None:
            spec.chunk_starts = self.packed_info[:, 0].contiguous()
            spec.chunk_cnts = self.packed_info[:, 1].contiguous()
        for k in ("ray_indices", "is_left", "is_right"):
            v = getattr(self, k)
            if v is not None:
                setattr(spec, k, v.contiguous())
        return spec

    @classmethod
    def _from_cpp(cls, spec):
        return cls(vals=spec.vals, packed_info=_packed(spec), ray_indices=spec.ray_indices,
                   is_left=spec.is_left, is_right=spec.is_right)

    @property
    def device(self) -> torch.device:
        return self.vals.device
