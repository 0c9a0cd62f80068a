"""PNG ingest with the reference's output rules (io.loadPNG, /root/reference/src/io.zig:242-307),
through the native decoder of the C ABI (oavif_png_decode, include/oavif_tq.h: the PNG
specification over zlib; no libspng, no Pillow).

    16-bit file        -> (h, w, 4) uint16, hbd = True     (RGBA16, io.zig:270-272)
    8-bit truecolour   -> (h, w, 3) uint8                  (RGB8,   io.zig:275)
    everything else    -> (h, w, 4) uint8                  (RGBA8: gray, gray+alpha, palette, RGBA;
                                                            io.zig:276-280)
tRNS is not applied (decode flags 0 at io.zig:285: no SPNG_DECODE_TRNS), so alpha is opaque unless the
file has an alpha channel.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

ERROR_NAMES = {-1: "InvalidArgument", -2: "GetHeaderFailed", -3: "DecodeFailed", -4: "ImageSizeFailed",
               -5: "OutOfMemory"}


class PngError(Exception):
    """Carries the reference's Zig error name (io.zig:254-283)."""

    def __init__(self, code: int):
        self.code = code
        self.name = ERROR_NAMES.get(code, "DecodeFailed")
        super().__init__(self.name)


def png_info(buf: bytes) -> _lib.PngInfo:
    L = _lib.lib()
    info = _lib.PngInfo()
    arr = (ctypes.c_uint8 * len(buf)).from_buffer_copy(buf)
    rc = L.oavif_png_info_from_memory(arr, len(buf), ctypes.byref(info))
    if rc != 0:
        raise PngError(rc)
    return info


def load_png(buf: bytes):
    """-> (pixels, channels, hbd, icc): pixels (h, w, channels) uint8, or uint16 when hbd."""
    L = _lib.lib()
    info = _lib.PngInfo()
    arr = (ctypes.c_uint8 * len(buf)).from_buffer_copy(buf)
    rc = L.oavif_png_info_from_memory(arr, len(buf), ctypes.byref(info))
    if rc != 0:
        raise PngError(rc)
    try:
        out = np.empty((info.height, info.width, info.channels), np.uint16 if info.hbd else np.uint8)
        icc = np.empty(info.icc_bytes, np.uint8) if info.icc_bytes else None
    except MemoryError:
        raise PngError(-5) from None
    rc = L.oavif_png_decode(arr, len(buf), out.ctypes.data_as(ctypes.c_void_p), out.nbytes,
                            icc.ctypes.data_as(ctypes.c_void_p) if icc is not None else None,
                            info.icc_bytes)
    if rc != 0:
        raise PngError(rc)
    return out, int(info.channels), bool(info.hbd), (icc.tobytes() if icc is not None else None)
