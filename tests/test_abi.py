"""CPU-only: the C-ABI libraries load and export every symbol the headers declare
(no compute calls without a GPU)."""
import ctypes
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cnc_[a-z0-9_A-Z]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    from cnc_amd import build
    return build.build_all()


def test_hip_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built[0])
    names = _declared("cnc_hip.h")
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"libcnc_hip.so does not export {n}"
    lib.cnc_abi_version.restype = ctypes.c_int
    assert lib.cnc_abi_version() >= 1
    lib.cnc_error_string.restype = ctypes.c_char_p
    assert lib.cnc_error_string(0) == b"ok"
    assert b"invalid" in lib.cnc_error_string(-1)


def test_no_process_wide_setters(built):
    """The header promises re-entrant, stream-ordered calls with no state between them: no `cnc_set_*` /
    `cnc_*_set_*` entry point, and no library-side global the kernels read."""
    assert not [n for n in _declared("cnc_hip.h") if "_set_" in n]
    lib = ctypes.CDLL(os.path.join(ROOT, "cnc_amd", "libcnc_hip.so"))
    for gone in ("cnc_set_persistent_share", "cnc_mlp_set_variant"):
        assert not hasattr(lib, gone)
    src = os.path.join(ROOT, "cnc_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".hpp")):
            text = open(os.path.join(src, f)).read()
            assert "static int g_" not in text and "persistent_share" not in text and "#ifdef CNC_EXP" not in text, f


def test_build_refuses_stray_extra_flags():
    import subprocess
    env = dict(os.environ, CNC_HIP_EXTRA_FLAGS="-DCNC_SOMETHING")
    env.pop("CNC_DIAG_BUILD", None)
    r = subprocess.run([sys.executable, "-c", "import cnc_amd.build"], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "CNC_DIAG_BUILD" in r.stderr


def test_ctypes_signature_table_covers_the_header(built):
    from cnc_amd import _lib
    declared = set(_declared("cnc_hip.h")) - {"cnc_error_string", "cnc_abi_version"}
    assert declared == set(_lib.SIGNATURES)
    L = _lib.lib()
    for n, argtypes in _lib.SIGNATURES.items():
        assert getattr(L, n).argtypes == argtypes


def test_codec_library_exports(built):
    lib = ctypes.CDLL(built[1])
    for n in _declared("cnc_codec.h"):
        assert hasattr(lib, n)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under cnc_amd/ may import or load it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cnc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M) or "libcnc_oracle" in txt:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_mirrors_refuse_cpu_tensors():
    """No CPU fallback: host tensors raise like the reference's CHECK_CUDA."""
    import torch
    from cnc_amd.backends import gridencoder_backend as be, nerfacc_cuda as nc, pack_and_align as pa
    x = torch.rand(4, 3)
    e = torch.rand(64, 2)
    o = torch.tensor([0, 64], dtype=torch.int32)
    r = torch.tensor([4], dtype=torch.int32)
    out = torch.empty(1, 4, 2)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        be.grid_encode_forward(x, e, o, r, out, 4, 3, 2, 1, 0, 128, 0.0, None, None, None)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        pa.query_mask_3D(torch.zeros(4, 3, dtype=torch.int16), torch.ones(4, 4, 4, dtype=torch.bool),
                         torch.zeros(4, dtype=torch.int16), torch.zeros(4, dtype=torch.int32), 18, 4)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        nc.exclusive_sum(torch.zeros(1, dtype=torch.long), torch.ones(1, dtype=torch.long), torch.ones(1), False, False)


def test_ctypes_structs_have_the_layout_of_the_header(tmp_path):
    """The structs that cross the C ABI by pointer (cnc_fused_field_t, cnc_field_pack_t): the ctypes mirrors in
    cnc_amd/_lib.py must have the size and the member offsets a C compiler gives the header's declarations — a member
    added on one side only would shift everything behind it silently."""
    import subprocess
    from cnc_amd import _lib
    members = {
        "cnc_fused_field_t": (_lib.FusedField, [f[0] for f in _lib.FusedField._fields_]),
        "cnc_field_pack_layer_t": (_lib.FieldPackLayer, [f[0] for f in _lib.FieldPackLayer._fields_]),
        "cnc_field_pack_t": (_lib.FieldPack, [f[0] for f in _lib.FieldPack._fields_]),
        "cnc_field_bwd_t": (_lib.FieldBwd, [f[0] for f in _lib.FieldBwd._fields_]),
        "cnc_field_save_t": (_lib.FieldSave, [f[0] for f in _lib.FieldSave._fields_]),
        "cnc_field_wgrad_t": (_lib.FieldWGrad, [f[0] for f in _lib.FieldWGrad._fields_]),
        "cnc_adam_table_t": (_lib.AdamTable, [f[0] for f in _lib.AdamTable._fields_]),
        "cnc_adam_tables_t": (_lib.AdamTables, [f[0] for f in _lib.AdamTables._fields_]),
        "cnc_ctx_window_t": (_lib.CtxWindow, [f[0] for f in _lib.CtxWindow._fields_]),
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include <stdint.h>', '#include "cnc_hip.h"', 'int main(void) {']
    for t, (_, names) in members.items():
        lines.append(f'  printf("{t} size %zu\\n", sizeof({t}));')
        for n in names:
            lines.append(f'  printf("{t} {n} %zu\\n", offsetof({t}, {n}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = 0
    for line in out.splitlines():
        t, name, value = line.split()
        cls = members[t][0]
        want = ctypes.sizeof(cls) if name == "size" else getattr(cls, name).offset
        assert int(value) == want, (t, name, int(value), want)
        seen += 1
    assert seen == sum(len(n) + 1 for _, n in members.values())
