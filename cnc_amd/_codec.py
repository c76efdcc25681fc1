"""ctypes loader of libcnc_codec.so (include/cnc_codec.h): the host-side entropy coder."""
from __future__ import annotations

import ctypes as C
import os

_lib = None

SIGNATURES = {
    "cnc_rc_bound": (C.c_int64, [C.c_int64]),
    "cnc_rc_encode_pm1": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "cnc_rc_decode_pm1": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "cnc_rc_encode_cdf16": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64]),
    "cnc_rc_decode_cdf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
}


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcnc_codec.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: build it with `python -m cnc_amd.build`")
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib
