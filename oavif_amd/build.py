"""Build the gfx950 shared library in-tree (oavif_amd/lib/liboavif_hip.so).

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "liboavif_hip.so")              # the product: C ABI of include/ssimu2_hip.h, oavif_tq.h
INSTR_LIB_PATH = os.path.join(LIB_DIR, "liboavif_hip_instr.so")  # + include/ssimu2_hip_internal.h (bench / tests only)
HOST_PATH = os.path.join(LIB_DIR, "oavif_host")   # csrc/oavif_host.c: main.zig's flow in C over the two public headers + libavif
SOURCES = ["ssimu2_hip.hip", "tq.cpp", "png_ingest.cpp"]
INSTR_SOURCES = ["ssimu2_instrument.hip", "tq.cpp", "png_ingest.cpp"]  # ssimu2_instrument.hip includes ssimu2_hip.hip


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build the gfx950 library)")


def _newer_than(t: float, paths) -> bool:
    return any(os.path.getmtime(d) > t for d in paths)


def _headers():
    inc = os.path.join(os.path.dirname(_HERE), "include")
    return [os.path.join(inc, s) for s in os.listdir(inc)]


def libs_need_build() -> bool:
    if not os.path.exists(LIB_PATH) or not os.path.exists(INSTR_LIB_PATH):
        return True
    t = min(os.path.getmtime(LIB_PATH), os.path.getmtime(INSTR_LIB_PATH))
    return _newer_than(t, [os.path.join(CSRC, s) for s in os.listdir(CSRC) if s != "oavif_host.c"] + _headers())


def host_needs_build() -> bool:
    if not os.path.exists(HOST_PATH):
        return True
    return _newer_than(os.path.getmtime(HOST_PATH), [os.path.join(CSRC, "oavif_host.c"), LIB_PATH] + _headers())


def needs_build() -> bool:
    return libs_need_build() or host_needs_build()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    # several ranks of one job may get here at once on a fresh checkout: one compiles (file
    # lock), the others wait and find the library up to date; the .so appears atomically
    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return LIB_PATH
        for target, sources in ((LIB_PATH, SOURCES), (INSTR_LIB_PATH, INSTR_SOURCES)) if force or libs_need_build() else ():
            tmp = target + f".tmp{os.getpid()}"
            cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                   "-ffp-contract=off",  # arithmetic contract: every FMA is an explicit fmaf()
                   "-fno-slp-vectorize",  # v_pk_*_f32 + operand shuffles are slower than scalar VALU here
                   "-fvisibility=hidden", "-fvisibility-inlines-hidden",  # exports = the headers' functions
                   "-Wall", "-Wno-unused-function", "-o", tmp]
            cmd += os.environ.get("OAVIF_AMD_EXTRA_HIPCC_FLAGS", "").split()  # A/B builds of experiments
            cmd += [os.path.join(CSRC, s) for s in sources]
            cmd += ["-lz"]  # png_ingest.cpp inflates with zlib
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, target)
            finally:
                if os.path.exists(tmp):
                    os.unlink(tmp)
        build_host(verbose)
    return LIB_PATH


def build_host(verbose: bool = False) -> str:
    """The C host (plain C over include/*.h; libavif is opened with dlopen at run time).  Linked against the
    product library next to it ($ORIGIN), the HIP runtime resolved through the library's own dependencies."""
    inc = os.path.join(os.path.dirname(_HERE), "include")
    tmp = HOST_PATH + f".tmp{os.getpid()}"
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-std=gnu11", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I", inc,
           os.path.join(CSRC, "oavif_host.c"), "-o", tmp, "-L", LIB_DIR, "-loavif_hip", "-ldl", "-lm", "-lpthread",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, HOST_PATH)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)
    return HOST_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
