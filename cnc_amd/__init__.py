"""cnc_amd — MI355X (gfx950) implementation of the CNC hot path.

Hash-grid encoder, occupancy-grid ray marcher / segmented scans, and the context-model aligner,
as hand-written HIP kernels behind a C ABI (include/cnc_hip.h, cnc_amd/libcnc_hip.so), with
host-side mirrors of the reference's Python/extension interface on top.
"""
from __future__ import annotations

import sys

__version__ = "0.1.0"


def install_dropins(force: bool = False) -> None:
    """Register the host mirrors under the reference's import names, so that the reference's own Python
    (`import _gridencoder as _backend` ngp.py:10, `import pack_and_align`, `import torchac` utils_bpp_acc.py:4-8,
    `import tinycudann as tcnn` ngp.py:13, `from nerfacc.estimators.occ_grid import ...` utils.py:18-25) resolves to
    this package with no edit:

        _gridencoder        -> cnc_amd.backends.gridencoder_backend
        pack_and_align      -> cnc_amd.backends.pack_and_align
        torchac             -> cnc_amd.backends.torchac          (over libcnc_codec.so)
        tinycudann          -> cnc_amd.backends.tinycudann       (SphericalHarmonics degree 4 only)
        gridencoder         -> cnc_amd.gridencoder               (exports GridEncoder)
        nerfacc[.sub.mod]   -> cnc_amd.nerfacc[.sub.mod]         (every submodule, so that `from nerfacc.x import y`
                                                                  finds the SAME module objects, not second copies)

    Names already imported from elsewhere are left alone unless `force`."""
    import importlib
    import pkgutil

    def put(alias, module):
        if force or alias not in sys.modules:
            sys.modules[alias] = module

    for alias, target in (("_gridencoder", "cnc_amd.backends.gridencoder_backend"),
                          ("pack_and_align", "cnc_amd.backends.pack_and_align"),
                          ("torchac", "cnc_amd.backends.torchac"),
                          ("tinycudann", "cnc_amd.backends.tinycudann"),
                          ("gridencoder", "cnc_amd.gridencoder")):
        put(alias, importlib.import_module(target))
    pkg = importlib.import_module("cnc_amd.nerfacc")
    put("nerfacc", pkg)
    for info in pkgutil.walk_packages(pkg.__path__, prefix="cnc_amd.nerfacc."):
        put(info.name[len("cnc_amd."):], importlib.import_module(info.name))
