"""ctypes binding of liboavif_hip.so (C ABI: include/ssimu2_hip.h, include/oavif_tq.h).

The product path has no CPU fallback: if the library is missing or fails to load this
module raises, and every scorer call needs a gfx950 device.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OAVIF_AMD_LIB") or os.path.join(_HERE, "lib", "liboavif_hip.so")
# the same scorer sources plus the hooks of include/ssimu2_hip_internal.h: bench / scripts / a few tests
INSTR_LIB_PATH = os.environ.get("OAVIF_AMD_INSTR_LIB") or os.path.join(_HERE, "lib", "liboavif_hip_instr.so")

OK = 0
ERR_INVALID_ARG = -1
ERR_UNSUPPORTED = -2
ERR_OOM = -3
ERR_HIP = -4
ERR_NO_REFERENCE = -5
ERR_NO_DEVICE = -6

STAGE_PYRAMID, STAGE_MARCH, STAGE_FINALIZE = 0, 1, 2
NUM_SCALES = 6
STATS_PER_SCALE = 18
TQ_MAX_PASS = 12

# every symbol the public headers declare; tests/test_abi.py checks that liboavif_hip.so exports
# exactly these (and that the Zig shim / INTEGRATION.md bind nothing else)
BLUR_FIR, BLUR_RECURSIVE, BLUR_RECURSIVE_FMA = 0, 1, 2   # ssimu2_ctx_set_blur (include/ssimu2_hip.h)

EXPORTED_SYMBOLS = (
    "ssimu2_ctx_create", "ssimu2_query_device", "ssimu2_ctx_device_info", "ssimu2_host_alloc", "ssimu2_host_free",
    "ssimu2_prefetch", "ssimu2_prefetch_join", "ssimu2_ctx_destroy", "ssimu2_ctx_set_blur",
    "ssimu2_last_error",
    "ssimu2_score_rgb8", "ssimu2_set_reference", "ssimu2_score_against_reference",
    "ssimu2_score_against_reference_strided", "ssimu2_set_reference_device",
    "ssimu2_enqueue_against_reference_device", "ssimu2_score_rgb8_device",
    "ssimu2_enqueue_rgb8_device", "ssimu2_wait", "ssimu2_last_averages",
    "ssimu2_version",
    "oavif_tq_default_options", "oavif_tq_predict_q_from_score",
    "oavif_tq_interpolate_quantizer", "oavif_tq_find_target_quality", "oavif_tq_search_hip",
    "oavif_tq_find_target_quality_speculative",
    "oavif_prescale_8_to_10", "oavif_prescale_16_to_10", "oavif_prescale_16_to_8",
    "oavif_png_info_from_memory", "oavif_png_decode",
)


class TQOptions(ctypes.Structure):
    _fields_ = [("score_tgt", ctypes.c_double), ("tolerance", ctypes.c_double),
                ("max_pass", ctypes.c_uint32)]


class TQPass(ctypes.Structure):
    _fields_ = [("q", ctypes.c_uint32), ("score", ctypes.c_double)]


class TQResult(ctypes.Structure):
    _fields_ = [("q", ctypes.c_uint32), ("score", ctypes.c_double),
                ("num_pass", ctypes.c_uint32), ("buf_q", ctypes.c_int32),
                ("history_len", ctypes.c_uint32), ("history", TQPass * TQ_MAX_PASS)]


class PngInfo(ctypes.Structure):   # oavif_png_info (include/oavif_tq.h)
    _fields_ = [("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("channels", ctypes.c_uint32),
                ("hbd", ctypes.c_int), ("data_bytes", ctypes.c_size_t), ("icc_bytes", ctypes.c_size_t),
                ("bit_depth", ctypes.c_uint32), ("color_type", ctypes.c_uint32), ("interlaced", ctypes.c_uint32)]


class TQSpecOptions(ctypes.Structure):
    """oavif_tq_spec_options; struct_size = OAVIF_TQ_SPEC_OPTIONS_TAG | sizeof is the ABI guard
    (include/oavif_tq.h), filled in here."""
    TAG = 0x71530000
    _fields_ = [("struct_size", ctypes.c_uint32), ("max_fanout", ctypes.c_uint32), ("first_wave_fanout", ctypes.c_uint32)]

    def __init__(self, max_fanout: int = 1, first_wave_fanout: int = 0):
        super().__init__(self.TAG | ctypes.sizeof(TQSpecOptions), int(max_fanout), int(first_wave_fanout))


class DeviceInfo(ctypes.Structure):
    """ssimu2_device_info (include/ssimu2_hip.h): what the library read off a HIP device."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("arch", ctypes.c_char * 64),
                ("name", ctypes.c_char * 128), ("pci_bus_id", ctypes.c_char * 32), ("compute_units", ctypes.c_uint32),
                ("lds_bytes_per_cu", ctypes.c_uint32), ("lds_bytes_per_workgroup", ctypes.c_uint32),
                ("wavefront_size", ctypes.c_uint32), ("hbm_bytes", ctypes.c_uint64), ("numa_node", ctypes.c_int32),
                ("usable", ctypes.c_int32)]

    def __init__(self):
        super().__init__()
        self.struct_size = ctypes.sizeof(DeviceInfo)

    def as_dict(self) -> dict:
        return {"device": int(self.device), "arch": self.arch.decode(), "name": self.name.decode(),
                "pci_bus_id": self.pci_bus_id.decode(), "compute_units": int(self.compute_units),
                "lds_bytes_per_cu": int(self.lds_bytes_per_cu), "lds_bytes_per_workgroup": int(self.lds_bytes_per_workgroup),
                "wavefront_size": int(self.wavefront_size), "hbm_bytes": int(self.hbm_bytes),
                "numa_node": int(self.numa_node), "usable": bool(self.usable)}


class TQSpecStats(ctypes.Structure):
    _fields_ = [("waves", ctypes.c_uint32), ("probes_issued", ctypes.c_uint32),
                ("cache_hits", ctypes.c_uint32)]


# include/ssimu2_hip_internal.h: only liboavif_hip_instr.so has these
INSTR_SYMBOLS = ("ssimu2_debug_download", "ssimu2_time_device", "ssimu2_time_stage",
                 "ssimu2_time_march_rotating", "ssimu2_measure_read_stream",
                 "ssimu2_instr_set_segment_rows", "ssimu2_instr_cache_reference_blur",
                 "ssimu2_instr_rg_stop_after_scale", "ssimu2_time_blur_stage_rotating",
                 "ssimu2_instr_placed_streams", "ssimu2_time_kernels", "ssimu2_instr_use_graph")

TQ_MAX_FANOUT = 16
BATCH_PROBE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32),
                                  ctypes.c_uint32, ctypes.POINTER(ctypes.c_double))
PROBE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32,
                            ctypes.POINTER(ctypes.c_double))
CODEC_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32,
                            ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_size_t))

_lib = None
_instr = None


def lib() -> ctypes.CDLL:
    """The product library (what a caller of the C ABI links)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, False)
    return _lib


def instr_lib() -> ctypes.CDLL:
    """The instrumented build (include/ssimu2_hip_internal.h); never used by the product path."""
    global _instr
    if _instr is None:
        _instr = _load(INSTR_LIB_PATH, True)
    return _instr


def _load(path: str, instrumented: bool) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -m oavif_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the scorer.")
    # PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64 (same SONAME as /opt/rocm's).
    # Whichever HIP runtime is loaded first serves the whole process, and torch cannot
    # initialise on top of a foreign one ("No HIP GPUs are available").  Python callers
    # share device memory and streams with torch, so let torch's runtime load first.
    if os.environ.get("OAVIF_AMD_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = ctypes.CDLL(path)
    vp, u8p, f64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_double)
    u32, ci = ctypes.c_uint32, ctypes.c_int
    L.ssimu2_ctx_create.argtypes = [ci, vp, ctypes.POINTER(vp)]
    L.ssimu2_ctx_create.restype = ci
    L.ssimu2_prefetch.argtypes = [ci]
    L.ssimu2_prefetch.restype = ci
    if hasattr(L, "ssimu2_query_device"):   # absent from builds before v8 (scripts/gpu_ab.py loads those too)
        L.ssimu2_query_device.argtypes = [ci, ctypes.POINTER(DeviceInfo)]
        L.ssimu2_query_device.restype = ci
        L.ssimu2_ctx_device_info.argtypes = [vp, ctypes.POINTER(DeviceInfo)]
        L.ssimu2_ctx_device_info.restype = ci
        L.ssimu2_host_alloc.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
        L.ssimu2_host_alloc.restype = ci
        L.ssimu2_host_free.argtypes = [vp, vp]
        L.ssimu2_host_free.restype = ci
    if hasattr(L, "ssimu2_prefetch_join"):
        L.ssimu2_prefetch_join.argtypes = [ci]
        L.ssimu2_prefetch_join.restype = ci
    L.ssimu2_ctx_destroy.argtypes = [vp]
    L.ssimu2_ctx_destroy.restype = None
    if hasattr(L, "ssimu2_ctx_set_blur"):   # absent from round-1 builds (scripts/gpu_ab.py loads those too)
        L.ssimu2_ctx_set_blur.argtypes = [vp, ci]
        L.ssimu2_ctx_set_blur.restype = ci
    L.ssimu2_last_error.argtypes = [vp]
    L.ssimu2_last_error.restype = ctypes.c_char_p
    L.ssimu2_score_rgb8.argtypes = [vp, u8p, u8p, u32, u32, u32, f64p]
    L.ssimu2_score_rgb8.restype = ci
    L.ssimu2_set_reference.argtypes = [vp, u8p, u32, u32]
    L.ssimu2_set_reference.restype = ci
    L.ssimu2_score_against_reference.argtypes = [vp, u8p, f64p]
    L.ssimu2_score_against_reference.restype = ci
    L.ssimu2_score_against_reference_strided.argtypes = [vp, u8p, u32, u32, f64p]
    L.ssimu2_score_against_reference_strided.restype = ci
    L.ssimu2_set_reference_device.argtypes = [vp, vp, u32, u32]
    L.ssimu2_set_reference_device.restype = ci
    L.ssimu2_enqueue_against_reference_device.argtypes = [vp, vp]
    L.ssimu2_enqueue_against_reference_device.restype = ci
    L.ssimu2_score_rgb8_device.argtypes = [vp, vp, vp, u32, u32, f64p]
    L.ssimu2_score_rgb8_device.restype = ci
    L.ssimu2_enqueue_rgb8_device.argtypes = [vp, vp, vp, u32, u32]
    L.ssimu2_enqueue_rgb8_device.restype = ci
    L.ssimu2_wait.argtypes = [vp, f64p]
    L.ssimu2_wait.restype = ci
    L.ssimu2_last_averages.argtypes = [vp, f64p, ctypes.POINTER(ci)]
    L.ssimu2_last_averages.restype = ci
    if instrumented:
        sigs = {
            "ssimu2_measure_read_stream": [vp, ctypes.c_size_t, ci, f64p],
            "ssimu2_debug_download": [vp, ci, ci, u32, u32, ctypes.POINTER(ctypes.c_float),
                                      ctypes.POINTER(u32), ctypes.POINTER(u32)],
            "ssimu2_time_device": [vp, vp, vp, u32, u32, ci, ctypes.POINTER(ctypes.c_float), f64p],
            "ssimu2_time_stage": [vp, vp, vp, u32, u32, ci, ci, ctypes.POINTER(ctypes.c_float)],
            "ssimu2_time_march_rotating": [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ci, u32, u32, ci,
                                           ctypes.POINTER(ctypes.c_float)],
            "ssimu2_instr_set_segment_rows": [vp, ci, ci],
            "ssimu2_instr_cache_reference_blur": [vp, ci],
            "ssimu2_instr_rg_stop_after_scale": [vp, ci],
            "ssimu2_instr_placed_streams": [vp, ctypes.POINTER(ci)],
            "ssimu2_instr_use_graph": [vp, ci, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)],
            "ssimu2_time_blur_stage_rotating": [vp, ctypes.POINTER(vp), ci, u32, u32, ci,
                                                ctypes.POINTER(ctypes.c_float), f64p],
            "ssimu2_time_kernels": [vp, vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ci, u32, u32, ci, ctypes.POINTER(ctypes.c_float),
                                    ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)],
        }
        for name, argtypes in sigs.items():
            if hasattr(L, name):  # scripts/gpu_ab.py also binds older builds that lack some hooks
                getattr(L, name).argtypes = argtypes
                getattr(L, name).restype = ci
    L.ssimu2_version.argtypes = []
    L.ssimu2_version.restype = ctypes.c_char_p
    L.oavif_tq_default_options.argtypes = [ctypes.POINTER(TQOptions)]
    L.oavif_tq_default_options.restype = None
    L.oavif_tq_predict_q_from_score.argtypes = [ctypes.c_double]
    L.oavif_tq_predict_q_from_score.restype = u32
    L.oavif_tq_interpolate_quantizer.argtypes = [u32, u32, ctypes.POINTER(TQPass), u32, ctypes.c_double]
    L.oavif_tq_interpolate_quantizer.restype = u32
    L.oavif_tq_find_target_quality.argtypes = [ctypes.POINTER(TQOptions), PROBE_FN, vp,
                                               ctypes.POINTER(TQResult)]
    L.oavif_tq_find_target_quality.restype = ci
    L.oavif_tq_find_target_quality_speculative.argtypes = [
        ctypes.POINTER(TQOptions), ctypes.POINTER(TQSpecOptions), BATCH_PROBE_FN, vp,
        ctypes.POINTER(TQResult), ctypes.POINTER(TQSpecStats)]
    L.oavif_tq_find_target_quality_speculative.restype = ci
    u16p = ctypes.POINTER(ctypes.c_uint16)
    L.oavif_prescale_8_to_10.argtypes = [u8p, ctypes.c_size_t, u16p]
    L.oavif_prescale_8_to_10.restype = None
    L.oavif_prescale_16_to_10.argtypes = [u16p, ctypes.c_size_t, u16p]
    L.oavif_prescale_16_to_10.restype = None
    L.oavif_prescale_16_to_8.argtypes = [u16p, ctypes.c_size_t, u8p]
    L.oavif_prescale_16_to_8.restype = None
    if hasattr(L, "oavif_png_decode"):   # absent from older builds (scripts/gpu_ab.py loads those too)
        L.oavif_png_info_from_memory.argtypes = [u8p, ctypes.c_size_t, ctypes.POINTER(PngInfo)]
        L.oavif_png_info_from_memory.restype = ci
        L.oavif_png_decode.argtypes = [u8p, ctypes.c_size_t, vp, ctypes.c_size_t, vp, ctypes.c_size_t]
        L.oavif_png_decode.restype = ci
    L.oavif_tq_search_hip.argtypes = [ctypes.POINTER(TQOptions), vp, u8p, u32, u32, CODEC_FN, vp,
                                      ctypes.POINTER(TQResult), ctypes.POINTER(ctypes.c_size_t)]
    L.oavif_tq_search_hip.restype = ci
    return L
