"""Build libcnc_hip.so (HIP kernels + C ABI) and libcnc_codec.so (CPU range coder) in-tree.

    python -m cnc_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the .so files are git-ignored but travel with the
working tree to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(_ROOT, "include")
LIB_HIP = os.path.join(_HERE, "libcnc_hip.so")
LIB_CODEC = os.path.join(_HERE, "libcnc_codec.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",          # arithmetic policy: nothing fuses unless written as fmaf
    "-munsafe-fp-atomics",        # fp32 atomicAdd -> global_atomic_add_f32 (no CAS loop)
    "-Wall", "-Wno-unused-function",
]
# Extra compiler flags are for diagnostics builds only and must be asked for twice: a stray CNC_HIP_EXTRA_FLAGS in the
# environment must not produce a library that differs from the one the tests pinned.
if os.environ.get("CNC_HIP_EXTRA_FLAGS"):
    if os.environ.get("CNC_DIAG_BUILD") != "1":
        raise RuntimeError("CNC_HIP_EXTRA_FLAGS is set but CNC_DIAG_BUILD=1 is not: refusing to build a non-standard libcnc_hip.so")
    HIP_FLAGS += os.environ["CNC_HIP_EXTRA_FLAGS"].split()


def _newer(srcs, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def hip_sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build_hip(force=False, verbose=False):
    srcs = hip_sources()
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps += [os.path.join(INCLUDE, "cnc_hip.h"), os.path.abspath(__file__)]
    if not (force or _newer(deps, LIB_HIP)):
        return LIB_HIP
    objs = []
    procs = []
    for s in srcs:   # one hipcc per translation unit, in parallel
        o = s[:-4] + ".o"
        cmd = [HIPCC] + [f for f in HIP_FLAGS if f != "-shared"] + ["-c", "-I", INCLUDE, "-o", o, s]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_HIP] + objs
    subprocess.run(cmd, check=True)
    return LIB_HIP


def build_codec(force=False, verbose=False):
    src = os.path.join(CSRC, "range_coder.cpp")
    if not os.path.exists(src):
        return None
    if not (force or _newer([src, os.path.join(INCLUDE, "cnc_codec.h")], LIB_CODEC)):
        return LIB_CODEC
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", INCLUDE,
           "-o", LIB_CODEC, src]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_CODEC


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_codec(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
