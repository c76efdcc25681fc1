"""`torchac` stand-in: the four entry points of torchac 0.9.3 (requirements.txt:32) over libcnc_codec.so.

The reference calls two of them (examples/utils_bpp_acc.py:87,108):

    byte_stream = torchac.encode_float_cdf(output_cdf, sym, check_input_bounds=True)
    sym_out     = torchac.decode_float_cdf(output_cdf, byte_stream)

with `output_cdf = cat([0, 1 - p, 1], -1)` float32 of shape [..., 3] and `sym` int16 in {0, 1} — both HOST tensors
(torchac is a CPU extension; the reference `.cpu()`s its inputs first).  The float CDF is turned into the coder's
16-bit integers the way torchac publishes it (`_convert_to_int_and_normalize`): `round(cdf * (2^16 - (Lp - 1)))`
in float32, cast to int16, plus `arange(Lp)` so that neighbouring entries differ by at least one.  torchac itself is a
third-party package absent from the reference tree: byte-level parity with it is UNPINNED (include/cnc_codec.h).
"""
from __future__ import annotations

import numpy as np
import torch

from .._codec import lib as _codec_lib

PRECISION = 16
__version__ = "0.9.3+cnc"


def _convert_to_int_and_normalize(cdf_float: torch.Tensor, needs_normalization: bool) -> torch.Tensor:
    Lp = cdf_float.shape[-1]
    top = float(2 ** PRECISION - (Lp - 1)) if needs_normalization else float(2 ** PRECISION)
    cdf = cdf_float.mul(top).round().to(torch.int32).to(torch.int16)     # 65536 wraps to 0, as in torchac
    if needs_normalization:
        cdf = cdf + torch.arange(Lp, dtype=torch.int16, device=cdf.device)
    return cdf


def _host(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.is_cuda:
        raise ValueError(f"{what} must be on CPU")        # torchac's own message
    return t


def _check_sym(cdf, sym):
    if sym.dtype != torch.int16:
        raise ValueError(f"sym must be int16, got {sym.dtype}")
    if tuple(cdf.shape[:-1]) != tuple(sym.shape):
        raise ValueError(f"Invalid shapes of cdf={tuple(cdf.shape)}, sym={tuple(sym.shape)}. The first m elements "
                         "of cdf.shape must be equal to sym.shape, and cdf should only have one more dimension.")


def encode_int16_normalized_cdf(cdf_int: torch.Tensor, sym: torch.Tensor) -> bytes:
    cdf_int, sym = _host(cdf_int, "cdf_int"), _host(sym, "sym")
    _check_sym(cdf_int, sym)
    if cdf_int.dtype != torch.int16:
        raise ValueError(f"cdf_int must be int16, got {cdf_int.dtype}")
    Lp = cdf_int.shape[-1]
    c = cdf_int.contiguous().view(-1, Lp)
    s = sym.contiguous().view(-1)
    n = s.numel()
    L = _codec_lib()
    cap = int(L.cnc_rc_bound(n))
    buf = np.empty(cap, dtype=np.uint8)
    nbytes = L.cnc_rc_encode_cdf16(c.data_ptr(), s.data_ptr(), n, Lp, buf.ctypes.data, cap)
    if nbytes == -2:
        raise ValueError(f"sym has values outside [0, {Lp - 2}]")
    if nbytes < 0:
        raise RuntimeError("range coder: output buffer too small")
    return buf[:nbytes].tobytes()


def decode_int16_normalized_cdf(cdf_int: torch.Tensor, byte_stream: bytes) -> torch.Tensor:
    cdf_int = _host(cdf_int, "cdf_int")
    if cdf_int.dtype != torch.int16:
        raise ValueError(f"cdf_int must be int16, got {cdf_int.dtype}")
    Lp = cdf_int.shape[-1]
    c = cdf_int.contiguous().view(-1, Lp)
    n = c.shape[0]
    out = torch.empty(n, dtype=torch.int16)
    stream = np.frombuffer(byte_stream, dtype=np.uint8)
    rc = _codec_lib().cnc_rc_decode_cdf16(c.data_ptr(), n, Lp, stream.ctypes.data, stream.shape[0], out.data_ptr())
    if rc != 0:
        raise ValueError("invalid CDF")
    return out.view(cdf_int.shape[:-1])


def encode_float_cdf(cdf_float: torch.Tensor, sym: torch.Tensor, needs_normalization: bool = True,
                     check_input_bounds: bool = False) -> bytes:
    """Bytes of `sym` under the per-symbol CDFs `cdf_float` ([..., Lp], first entry 0, last entry 1)."""
    cdf_float, sym = _host(cdf_float, "cdf_float"), _host(sym, "sym")
    if check_input_bounds:
        if cdf_float.min() < 0:
            raise ValueError(f"cdf_float.min() == {cdf_float.min()}, should be >=0.!")
        if cdf_float.max() > 1:
            raise ValueError(f"cdf_float.max() == {cdf_float.max()}, should be <=1.!")
        Lp = cdf_float.shape[-1]
        if sym.max() >= Lp - 1:
            raise ValueError(f"sym.max() == {sym.max()}, should be <=Lp - 1.!")
    return encode_int16_normalized_cdf(_convert_to_int_and_normalize(cdf_float, needs_normalization), sym)


def decode_float_cdf(cdf_float: torch.Tensor, byte_stream: bytes, needs_normalization: bool = True) -> torch.Tensor:
    """int16 symbols of shape cdf_float.shape[:-1]."""
    cdf_float = _host(cdf_float, "cdf_float")
    return decode_int16_normalized_cdf(_convert_to_int_and_normalize(cdf_float, needs_normalization), byte_stream)
