"""Builds libdronenav.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libdronenav.so")
SOURCES = ["dn_kernels.hip", "dn_kernels_mw.hip", "dn_mlp.hip", "dn_fused.hip", "dn_capi.cpp"]
HEADERS = ["dn_internal.h", os.path.join("..", "..", "include", "dronenav.h")]
# -ffp-contract=off: no multiply-add is fused BY LICENCE -- the float32 action chain rounds every operation as numpy does, and the
# float64 part writes its fused multiply-adds out explicitly so that every kernel shape produces the same bits.
# Correctly rounded float32 divide/sqrt is hipcc's default; stated explicitly because parity relies on it.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# The fused K-step kernel loops over the whole step body; machine LICM would hoist every float64 literal of the body
# out of that loop and keep them live.  That costs the one-wave kernels, which run several waves per SIMD at large
# fleets, their occupancy (2 M drones fused: 72 -> 107 us per step) and is a gain for the multi-wave kernels, which
# are (nearly) alone on their SIMD (32768 drones, three waves: 1.48 -> 1.37 us per step).  So the multi-wave kernels
# of the plain configuration are a translation unit of their own (dn_kernels_mw.hip includes dn_kernels.hip).
NO_LICM = ["-mllvm", "-disable-machine-licm"]
FLAGS_OF = {"dn_kernels_mw.hip": []}             # everything else: NO_LICM


EXTRA = os.environ.get("DN_EXTRA_HIPCC_FLAGS", "").split()


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libdronenav.so cannot be built (there is no CPU fallback)")
    return exe


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile csrc/*.{hip,cpp} -> csrc/libdronenav.so.  Returns the library path."""
    if not force and not is_stale():
        return LIB_PATH
    objs, cmds = [], []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc()] + FLAGS + FLAGS_OF.get(src, NO_LICM) + EXTRA + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        objs.append(obj)
        cmds.append(cmd)
    procs = [subprocess.Popen(c) for c in cmds]              # the translation units are independent: compile them side by side
    codes = [p.wait() for p in procs]
    for c, rc in zip(cmds, codes):
        if rc != 0:
            raise subprocess.CalledProcessError(rc, c)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
