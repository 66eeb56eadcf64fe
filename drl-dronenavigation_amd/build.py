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
# The same ABI with the observation normaliser's OUTPUT stage in float64 (-DDN_NORM_EXACT=1, dn_kernels.hip normalize_obs_cols): the
# float32 nearest to the float64 evaluation instead of <= 3 float32 ulp.  A build of its own because the form is a compile-time one (as a
# run-time switch it costs the fused kernels ~100 spilled registers); DN_EXACT_NORM=1 makes _capi.load() take this library.
LIB_PATH_EXACT = os.path.join(CSRC, "libdronenav_exact.so")
SOURCES = ["dn_kernels.hip", "dn_kernels_mw.hip", "dn_mlp.hip", "dn_fused.hip", "dn_capi.cpp"]
STEP_SOURCES = ["dn_kernels.hip", "dn_kernels_mw.hip", "dn_fused.hip"]      # the translation units that inline the normaliser
HEADERS = ["dn_internal.h", "dn_action_sat.h", os.path.join("..", "..", "include", "dronenav.h")]
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
# The policy kernels keep their MFMA accumulators in architectural VGPRs (-amdgpu-mfma-vgpr-form): with AGPR accumulators every value of a
# tile's epilogue is first copied out with v_accvgpr_read_b32 (664 of them per wave in the four-wave kernel, 833 in the float32-grade one)
# before the vector ALU can touch it; in VGPR form the exp reads the accumulator directly and only finished operands are parked in AGPRs
# (4 + 80 moves).  Interleaved A/B at 32 768 drones: 57.2 -> 55.7 us (bf16), 154.2 -> 151.6 us (float32 grade); profiles/r06_notes.md.
MFMA_VGPR = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
FLAGS_OF = {"dn_kernels_mw.hip": [], "dn_mlp.hip": NO_LICM + MFMA_VGPR, "dn_fused.hip": NO_LICM + MFMA_VGPR}   # everything else: NO_LICM


EXTRA = os.environ.get("DN_EXTRA_HIPCC_FLAGS", "").split()


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libdronenav.so cannot be built (there is no CPU fallback)")
    return exe


def is_stale():
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    for lib in (LIB_PATH, LIB_PATH_EXACT):
        if not os.path.exists(lib):
            return True
        t = os.path.getmtime(lib)
        if any(os.path.getmtime(d) > t for d in deps):
            return True
    return False


def build_library(force=False, verbose=False):
    """Compile csrc/*.{hip,cpp} -> csrc/libdronenav.so and csrc/libdronenav_exact.so.  Returns the path of the default library."""
    if not force and not is_stale():
        return LIB_PATH
    objs, objs_exact, cmds = [], [], []
    for src in SOURCES:
        stem = os.path.join(CSRC, os.path.splitext(src)[0])
        base = [hipcc()] + FLAGS + FLAGS_OF.get(src, NO_LICM) + EXTRA
        cmds.append(base + ["-c", os.path.join(CSRC, src), "-o", stem + ".o"])
        objs.append(stem + ".o")
        if src in STEP_SOURCES:
            cmds.append(base + ["-DDN_NORM_EXACT=1", "-c", os.path.join(CSRC, src), "-o", stem + ".exact.o"])
            objs_exact.append(stem + ".exact.o")
        else:
            objs_exact.append(stem + ".o")
    if verbose:
        for c in cmds:
            print(" ".join(c))
    procs = [subprocess.Popen(c) for c in cmds]              # the translation units are independent: compile them side by side
    codes = [p.wait() for p in procs]
    for c, rc in zip(cmds, codes):
        if rc != 0:
            raise subprocess.CalledProcessError(rc, c)
    for lib, ob in ((LIB_PATH, objs), (LIB_PATH_EXACT, objs_exact)):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + ob
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
