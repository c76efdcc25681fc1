"""cnc_amd — MI355X (gfx950) implementation of the CNC hot path.

Hash-grid encoder, occupancy-grid ray marcher / segmented scans, and the context-model aligner,
as hand-written HIP kernels behind a C ABI (include/cnc_hip.h, cnc_amd/libcnc_hip.so), with
host-side mirrors of the reference's Python/extension interface on top.
"""
from __future__ import annotations

import sys

__version__ = "0.1.0"


def install_dropins() -> None:
    """Register the host mirrors under the reference's import names, so the reference's own
    Python (`import _gridencoder as _backend`, `import pack_and_align`, `from . import cuda as _C`)
    resolves to this package:

        _gridencoder        -> cnc_amd.backends.gridencoder_backend
        pack_and_align      -> cnc_amd.backends.pack_and_align
        gridencoder         -> cnc_amd.gridencoder   (exports GridEncoder)
        nerfacc             -> cnc_amd.nerfacc
    """
    import importlib

    from .backends import gridencoder_backend, pack_and_align

    sys.modules.setdefault("_gridencoder", gridencoder_backend)
    sys.modules.setdefault("pack_and_align", pack_and_align)
    for alias, target in (("gridencoder", "cnc_amd.gridencoder"), ("nerfacc", "cnc_amd.nerfacc")):
        if alias not in sys.modules:
            sys.modules[alias] = importlib.import_module(target)
