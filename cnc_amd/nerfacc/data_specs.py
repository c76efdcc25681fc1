"""The two sample records of the nerfacc API (reference: nerfacc/data_specs.py:12-180): `RaySamples` — sample
positions along rays — and `RayIntervals` — the edges of the sample intervals.  Either *batched*
(`vals` is (n_rays, n)) or *flattened* (`vals` is 1-D and `packed_info` (n_rays, 2) = (start, count) and / or
`ray_indices` say which ray an entry belongs to).  `_to_cpp` / `_from_cpp` convert to and from the extension's
`RaySegmentsSpec` (here: cnc_amd.backends.nerfacc_cuda.RaySegmentsSpec)."""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Optional

import torch

from . import cuda as _C

# record field -> RaySegmentsSpec attribute, for the optional per-entry arrays
_PER_ENTRY = ("ray_indices", "is_left", "is_right", "is_valid")


def _spec_of(record):
    """RaySegmentsSpec holding the record's tensors (contiguous); packed_info is split into its columns."""
    spec = _C.RaySegmentsSpec()
    spec.vals = record.vals.contiguous()
    info = record.packed_info
    if info is not None:
        spec.chunk_starts, spec.chunk_cnts = (info[:, k].contiguous() for k in (0, 1))
    for name in _PER_ENTRY:
        t = getattr(record, name, None)
        if t is not None:
            setattr(spec, name, t.contiguous())
    return spec


def _record_of(cls, spec):
    """Inverse of `_spec_of` for the fields `cls` declares; (starts, counts) are re-joined into packed_info."""
    have = {f.name for f in fields(cls)}
    kw = {name: getattr(spec, name) for name in _PER_ENTRY if name in have}
    joined = None
    if spec.chunk_starts is not None and spec.chunk_cnts is not None:
        joined = torch.stack((spec.chunk_starts, spec.chunk_cnts), dim=-1)
    return cls(vals=spec.vals, packed_info=joined, **kw)


class _Record:
    def _to_cpp(self):
        return _spec_of(self)

    @classmethod
    def _from_cpp(cls, spec):
        return _record_of(cls, spec)

    @property
    def device(self) -> torch.device:
        return self.vals.device


@dataclass
class RaySamples(_Record):
    """`vals`: sample distances; `is_valid` marks the used slots of an over-allocated march."""
    vals: torch.Tensor
    packed_info: Optional[torch.Tensor] = None
    ray_indices: Optional[torch.Tensor] = None
    is_valid: Optional[torch.Tensor] = None


@dataclass
class RayIntervals(_Record):
    """`vals`: interval edges; `is_left` / `is_right` flag the edges that open / close a sample interval (an edge
    shared by two consecutive intervals carries both)."""
    vals: torch.Tensor
    packed_info: Optional[torch.Tensor] = None
    ray_indices: Optional[torch.Tensor] = None
    is_left: Optional[torch.Tensor] = None
    is_right: Optional[torch.Tensor] = None
