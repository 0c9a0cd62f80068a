"""ctypes front-end of the CPU oracle (oracle/ssimu2_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of ssimu2_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by oavif_amd/.

Parity status: UNPINNED against fssimu2 0.1.1 (source absent, reference has no tests);
restates the published SSIMULACRA2 v2.1 algorithm behind the signature seen at
/root/reference/src/tq.zig:37.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BLUR_IIR = 0  # libjxl recursive Gaussian, as published
BLUR_FIR = 1  # its exact-arithmetic equivalent, 9-tap zero-padded FIR
BLUR_EXACT = 2  # the same operator accumulated in fp64 (what both fp32 forms approximate)
BLUR_IIR_FMA = 3  # BLUR_IIR with the recursion's multiply-subtract fused: a second fp32 evaluation order
BLUR_FIR_PRODFIRST = 4  # BLUR_FIR with x*x, y*y, x*y rounded first, then blurred: the published order of operations

# stage variants of the pin kit (OR_VAR_* in ssimu2_oracle.c): single stages switched to plausible alternatives
VAR_EDGE_CLAMP, VAR_EDGE_MIRROR, VAR_GAUSS9, VAR_GAUSS11 = 0x1, 0x2, 0x4, 0x8
VAR_DOWNSAMPLE_XYB, VAR_DOWNSAMPLE_FLOOR, VAR_SIZE_TEST_AFTER = 0x10, 0x20, 0x40
VAR_SRGB_POWF, VAR_CBRT_LIBM, VAR_SUMS_F32 = 0x80, 0x100, 0x200

# The pin kit's catalogue: name -> (blur mode, variant bits, stage, what differs from the published algorithm).
# The first three are the scorer's blur modes themselves (the only ones the HIP kernels implement); every other
# entry switches ONE stage, on top of the published recursion and on top of the FIR form.
PIN_VARIANTS = {
    "fir": (BLUR_FIR, 0, "blur", "the 9-tap impulse response of the published recursion, products fused into the pair sums (SSIMU2_BLUR_FIR)"),
    "recursive": (BLUR_IIR, 0, "blur", "the published fp32 recursion, scalar order (SSIMU2_BLUR_RECURSIVE)"),
    "recursive_fma": (BLUR_IIR_FMA, 0, "blur", "the recursion with its multiply-subtract fused (SSIMU2_BLUR_RECURSIVE_FMA)"),
    "fir_prodfirst": (BLUR_FIR_PRODFIRST, 0, "blur", "9-tap FIR, products rounded to fp32 first and then blurred"),
    "fir_edge_clamp": (BLUR_FIR_PRODFIRST, VAR_EDGE_CLAMP, "blur", "9-tap FIR, samples outside the plane replicate the edge"),
    "fir_edge_mirror": (BLUR_FIR_PRODFIRST, VAR_EDGE_MIRROR, "blur", "9-tap FIR, samples outside the plane mirrored"),
    "gauss9": (BLUR_FIR_PRODFIRST, VAR_GAUSS9, "blur", "true sigma-1.5 Gaussian, radius 4, normalised, zero padding"),
    "gauss11": (BLUR_FIR_PRODFIRST, VAR_GAUSS11, "blur", "true sigma-1.5 Gaussian, radius 5, normalised, zero padding"),
    "gauss9_edge_clamp": (BLUR_FIR_PRODFIRST, VAR_GAUSS9 | VAR_EDGE_CLAMP, "blur", "true Gaussian radius 4, edge replicated"),
    "gauss11_edge_mirror": (BLUR_FIR_PRODFIRST, VAR_GAUSS11 | VAR_EDGE_MIRROR, "blur", "true Gaussian radius 5, mirrored"),
}
for _base, _mode in (("recursive", BLUR_IIR), ("fir", BLUR_FIR)):
    PIN_VARIANTS.update({
        f"{_base}+downsample_xyb": (_mode, VAR_DOWNSAMPLE_XYB, "pyramid", "scales 1..5 average the previous scale's XYB planes, not linear light"),
        f"{_base}+downsample_floor": (_mode, VAR_DOWNSAMPLE_FLOOR, "pyramid", "odd sizes drop the last row / column (floor) instead of replicating it (ceil)"),
        f"{_base}+size_test_after": (_mode, VAR_SIZE_TEST_AFTER, "pyramid", "a scale is scored only if its own size is >= 8"),
        f"{_base}+srgb_powf": (_mode, VAR_SRGB_POWF, "colour", "sRGB transfer curve evaluated in fp32 (powf)"),
        f"{_base}+cbrt_libm": (_mode, VAR_CBRT_LIBM, "colour", "cube root from libm's cbrtf"),
        f"{_base}+sums_f32": (_mode, VAR_SUMS_F32, "maps", "SSIM / edge maps and their sums in fp32"),
    })

_libs: dict[bool, ctypes.CDLL] = {}


def _host_has_avx2_fma() -> bool:
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return " avx2" in flags and " fma" in flags


def omp_build_name() -> str:
    """Which OpenMP build `omp=True` uses: the AVX2 + FMA build of the same source when the host
    has both (same arithmetic: explicit fmaf only), else the portable one."""
    return "libssimu2_oracle_fast.so" if _host_has_avx2_fma() else "libssimu2_oracle_omp.so"


def build() -> None:
    """Compile the oracle (both the scalar and the OpenMP build)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def _lib(omp: bool = False) -> ctypes.CDLL:
    if omp not in _libs:
        name = omp_build_name() if omp else "libssimu2_oracle.so"
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        f32p = ctypes.POINTER(ctypes.c_float)
        f64p = ctypes.POINTER(ctypes.c_double)
        lib.or_compute_ssimu2.argtypes = [u8p, u8p, ctypes.c_uint32, ctypes.c_uint32,
                                          ctypes.c_uint32, ctypes.c_int, f64p, f64p,
                                          ctypes.POINTER(ctypes.c_int)]
        lib.or_compute_ssimu2.restype = ctypes.c_int
        lib.or_compute_ssimu2_variant.argtypes = [u8p, u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int,
                                                  ctypes.c_uint, f64p, f64p, ctypes.POINTER(ctypes.c_int)]
        lib.or_compute_ssimu2_variant.restype = ctypes.c_int
        lib.or_blur_plane.argtypes = [f32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, f32p]
        lib.or_blur_plane.restype = None
        lib.or_blur_product.argtypes = [f32p, f32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, f32p]
        lib.or_blur_product.restype = None
        lib.or_gauss_taps.argtypes = [f64p, f32p, f64p, f64p]
        lib.or_gauss_taps.restype = None
        lib.or_srgb_lut.argtypes = [f32p]
        lib.or_srgb_lut.restype = None
        lib.or_linear_to_xyb.argtypes = [f32p, ctypes.c_size_t, f32p]
        lib.or_linear_to_xyb.restype = None
        lib.or_downsample2.argtypes = [f32p, ctypes.c_size_t, ctypes.c_size_t, f32p]
        lib.or_downsample2.restype = None
        lib.or_score_from_averages.argtypes = [f64p, ctypes.c_int]
        lib.or_score_from_averages.restype = ctypes.c_double
        lib.or_cbrtf.argtypes = [ctypes.c_float]
        lib.or_cbrtf.restype = ctypes.c_float
        lib.or_set_num_threads.argtypes = [ctypes.c_int]
        lib.or_set_num_threads.restype = ctypes.c_int
        lib.or_copy_rgb_pixels.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, u8p]
        lib.or_copy_rgb_pixels.restype = None
        lib.or_weights.argtypes = [f64p]
        lib.or_weights.restype = None
        _libs[omp] = lib
    return _libs[omp]


def _u8(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _f32(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _f64(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def compute_ssimu2(ref: np.ndarray, dist: np.ndarray, blur: int = BLUR_IIR, omp: bool = False,
                   return_averages: bool = False):
    """Score `dist` against `ref`; both (h, w, 3) uint8, tightly packed.

    Mirrors `fssimu2.computeSsimu2(allocator, ref, dist, w, h, 3, null)` (tq.zig:37).
    """
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    dist = np.ascontiguousarray(dist, dtype=np.uint8)
    if ref.shape != dist.shape or ref.ndim != 3 or ref.shape[2] != 3:
        raise ValueError("ref and dist must both be (h, w, 3) uint8")
    h, w, _ = ref.shape
    score = ctypes.c_double(0.0)
    avg = np.zeros(6 * 18, dtype=np.float64)
    nsc = ctypes.c_int(0)
    rc = _lib(omp).or_compute_ssimu2(_u8(ref), _u8(dist), w, h, 3, blur, ctypes.byref(score),
                                     _f64(avg), ctypes.byref(nsc))
    if rc != 0:
        raise RuntimeError(f"or_compute_ssimu2 failed rc={rc}")
    if return_averages:
        return score.value, avg.reshape(6, 18), nsc.value
    return score.value


def compute_ssimu2_variant(ref: np.ndarray, dist: np.ndarray, blur: int, variant: int, omp: bool = False,
                           return_averages: bool = False):
    """The score with the stages named by the VAR_* bits of `variant` switched to their alternatives (pin kit only;
    variant = 0 is compute_ssimu2).  Blur variants need blur = BLUR_FIR / BLUR_FIR_PRODFIRST."""
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    dist = np.ascontiguousarray(dist, dtype=np.uint8)
    if ref.shape != dist.shape or ref.ndim != 3 or ref.shape[2] != 3:
        raise ValueError("ref and dist must both be (h, w, 3) uint8")
    h, w, _ = ref.shape
    score = ctypes.c_double(0.0)
    avg = np.zeros(6 * 18, dtype=np.float64)
    nsc = ctypes.c_int(0)
    rc = _lib(omp).or_compute_ssimu2_variant(_u8(ref), _u8(dist), w, h, blur, int(variant), ctypes.byref(score),
                                             _f64(avg), ctypes.byref(nsc))
    if rc != 0:
        raise ValueError(f"or_compute_ssimu2_variant refused blur {blur} / variant {variant:#x} (rc {rc})")
    if return_averages:
        return score.value, avg.reshape(6, 18), nsc.value
    return score.value


def pin_variant_score(ref: np.ndarray, dist: np.ndarray, name: str, omp: bool = False) -> float:
    blur, var, _stage, _what = PIN_VARIANTS[name]
    return compute_ssimu2_variant(ref, dist, blur, var, omp=omp)


def blur_plane(plane: np.ndarray, blur: int = BLUR_IIR) -> np.ndarray:
    plane = np.ascontiguousarray(plane, dtype=np.float32)
    h, w = plane.shape
    out = np.empty_like(plane)
    _lib().or_blur_plane(_f32(plane), w, h, blur, _f32(out))
    return out


def blur_product(a: np.ndarray, b: np.ndarray, blur: int = BLUR_FIR) -> np.ndarray:
    """blur(a * b) of two (h, w) float32 planes the way the score forms it in mode `blur`."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    h, w = a.shape
    out = np.empty_like(a)
    _lib().or_blur_product(_f32(a), _f32(b), w, h, blur, _f32(out))
    return out


def gauss_taps():
    t64 = np.zeros(5, np.float64)
    t32 = np.zeros(5, np.float32)
    n2 = np.zeros(3, np.float64)
    d1 = np.zeros(3, np.float64)
    _lib().or_gauss_taps(_f64(t64), _f32(t32), _f64(n2), _f64(d1))
    return t64, t32, n2, d1


def srgb_lut() -> np.ndarray:
    lut = np.zeros(256, np.float32)
    _lib().or_srgb_lut(_f32(lut))
    return lut


def linear_to_xyb(lin: np.ndarray) -> np.ndarray:
    """lin: (3, h, w) float32 linear RGB planes -> (3, h, w) positive-XYB planes."""
    lin = np.ascontiguousarray(lin, dtype=np.float32)
    out = np.empty_like(lin)
    _lib().or_linear_to_xyb(_f32(lin), lin.shape[1] * lin.shape[2], _f32(out))
    return out


def downsample2(lin: np.ndarray) -> np.ndarray:
    lin = np.ascontiguousarray(lin, dtype=np.float32)
    _, h, w = lin.shape
    out = np.empty((3, (h + 1) // 2, (w + 1) // 2), np.float32)
    _lib().or_downsample2(_f32(lin), w, h, _f32(out))
    return out


def score_from_averages(avg: np.ndarray, nscales: int) -> float:
    avg = np.ascontiguousarray(avg, dtype=np.float64).reshape(-1)
    return float(_lib().or_score_from_averages(_f64(avg), nscales))


def weights() -> np.ndarray:
    w = np.zeros(108, np.float64)
    _lib().or_weights(_f64(w))
    return w


def cbrtf(x: float) -> float:
    return float(_lib().or_cbrtf(float(x)))


def set_num_threads(n: int) -> int:
    """Thread count of the OpenMP build; returns the count in effect."""
    return int(_lib(True).or_set_num_threads(int(n)))


def copy_rgb_pixels(pixels: np.ndarray) -> np.ndarray:
    """io.zig:654-663 on an (h, w, 3|4) uint8 array with possibly padded rows -> tight (h, w, 3)."""
    assert pixels.dtype == np.uint8 and pixels.ndim == 3 and pixels.strides[2] == 1
    assert pixels.strides[1] == pixels.shape[2]
    h, w, ch = pixels.shape
    out = np.empty((h, w, 3), np.uint8)
    src = ctypes.cast(ctypes.c_void_p(pixels.ctypes.data), ctypes.POINTER(ctypes.c_uint8))
    _lib().or_copy_rgb_pixels(src, pixels.strides[0], ch, w, h, _u8(out))
    return out
