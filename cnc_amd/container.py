"""Single-file bitstream container.

The reference leaves its result scattered: one `.b` file per level/chunk (utils_bpp_acc.py:722,752,
793,854), the 24 level frequencies `Pgs_dict` only in memory (:710), the occupancy grid and the
13-bit MLP only *estimated* (train_CNC_nerf_synthetic.py:53-68,508-556) and the context models not
stored at all.  This module writes everything a decoder needs into ONE file, so "size (KB)" is a
real file size:

    magic "CNC1" | u32 header_len | header JSON | payload sections (offsets in the header)

sections: every arithmetic-coded table stream, the occupancy grid (arithmetic-coded with its own
frequency), the radiance-field MLP tensors uniformly quantised to `mlp_bits` (13) bits and
bit-packed, the context-model weights in fp32.
"""
from __future__ import annotations

import io
import json
import os
import struct
import tempfile
from typing import Dict

import numpy as np
import torch

from .context import decoder, encoder

MAGIC = b"CNC1"


def _pack_bits(values: np.ndarray, bits: int) -> bytes:
    """values: uint32 < 2^bits -> little-endian bit stream."""
    v = values.astype(np.uint64).reshape(-1)
    out = np.zeros((v.size * bits + 7) // 8 + 8, dtype=np.uint8)
    pos = np.arange(v.size, dtype=np.uint64) * np.uint64(bits)
    for b in range(bits):                       # bit-plane at a time (vectorised)
        bit = ((v >> np.uint64(b)) & np.uint64(1)).astype(np.uint8)
        p = pos + np.uint64(b)
        np.bitwise_or.at(out, (p >> np.uint64(3)).astype(np.int64), bit << (p & np.uint64(7)).astype(np.uint8))
    return out[: (v.size * bits + 7) // 8].tobytes()


def _unpack_bits(buf: bytes, n: int, bits: int) -> np.ndarray:
    raw = np.frombuffer(buf, dtype=np.uint8)
    allbits = np.unpackbits(raw, bitorder="little")[: n * bits].reshape(n, bits).astype(np.uint32)
    return (allbits << np.arange(bits, dtype=np.uint32)).sum(axis=1).astype(np.uint32)


def quantize_tensor(p: torch.Tensor, bits: int):
    """(p - min) // interval, interval = (max - min) / (2^bits - 1) + 1e-6
    (quantize_params, train_CNC_nerf_synthetic.py:30-50)."""
    lo, hi = float(p.min()), float(p.max())
    interval = (hi - lo) / (2 ** bits - 1) + 1e-6
    q = torch.div(p - lo, interval, rounding_mode="floor").clamp_(0, 2 ** bits - 1)
    return q.to(torch.int64).cpu().numpy().astype(np.uint32), lo, interval


def write_container(path: str, *, meta: Dict, table_streams: Dict[str, bytes], binaries: torch.Tensor,
                    field_mlp: Dict[str, torch.Tensor], context_state: Dict[str, torch.Tensor],
                    mlp_bits: int = 13) -> int:
    """Returns the file size in bytes."""
    sections, payload = {}, io.BytesIO()

    def add(name, blob, **extra):
        sections[name] = dict(offset=payload.tell(), size=len(blob), **extra)
        payload.write(blob)

    for name, blob in table_streams.items():
        add("table/" + name, blob)
    # occupancy grid: Bernoulli(Pg) arithmetic code, what get_binary_vxl_size only estimates
    occ = binaries.reshape(-1).to(torch.float32).cpu()
    pg = float(occ.mean().clamp(1e-6, 1 - 1e-6))
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "occ.b")
        encoder(occ * 2 - 1, torch.full_like(occ, pg), f)
        add("occupancy", open(f, "rb").read(), shape=list(binaries.shape), pg=pg)
    for name, p in field_mlp.items():
        q, lo, interval = quantize_tensor(p.detach().float(), mlp_bits)
        add("mlp/" + name, _pack_bits(q, mlp_bits), shape=list(p.shape), lo=lo, interval=interval, bits=mlp_bits)
    for name, p in context_state.items():
        add("ctx/" + name, p.detach().float().cpu().numpy().tobytes(), shape=list(p.shape))
    header = json.dumps(dict(meta=meta, sections=sections)).encode()
    with open(path, "wb") as fo:
        fo.write(MAGIC)
        fo.write(struct.pack("<I", len(header)))
        fo.write(header)
        fo.write(payload.getvalue())
    return os.path.getsize(path)


def read_container(path: str, device="cpu"):
    """Returns (meta, table_streams, binaries, field_mlp (dequantised), context_state)."""
    with open(path, "rb") as fi:
        assert fi.read(4) == MAGIC, "not a CNC1 container"
        (hlen,) = struct.unpack("<I", fi.read(4))
        header = json.loads(fi.read(hlen))
        blob = fi.read()
    sec = header["sections"]

    def get(name):
        s = sec[name]
        return blob[s["offset"]: s["offset"] + s["size"]], s

    tables, mlp, ctx = {}, {}, {}
    binaries = None
    for name in sec:
        data, s = get(name)
        if name.startswith("table/"):
            tables[name[6:]] = data
        elif name == "occupancy":
            n = int(np.prod(s["shape"]))
            with tempfile.TemporaryDirectory() as td:
                f = os.path.join(td, "occ.b")
                open(f, "wb").write(data)
                x = decoder(torch.full((n,), s["pg"], dtype=torch.float32), f)
            binaries = (x > 0).view(*s["shape"]).to(device)
        elif name.startswith("mlp/"):
            n = int(np.prod(s["shape"]))
            q = _unpack_bits(data, n, s["bits"]).astype(np.float32)
            mlp[name[4:]] = (torch.from_numpy(q) * s["interval"] + s["lo"]).view(*s["shape"]).to(device)
        elif name.startswith("ctx/"):
            arr = np.frombuffer(data, dtype=np.float32).copy()
            ctx[name[4:]] = torch.from_numpy(arr).view(*s["shape"]).to(device)
    return header["meta"], tables, binaries, mlp, ctx
