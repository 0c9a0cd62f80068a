"""oavif_amd -- MI355X-native target-quality search path of oavif.

Scope: /root/reference/src/tq.zig and the SSIMULACRA2 scorer it calls
(fssimu2.computeSsimu2, tq.zig:37), as hand-written HIP for gfx950 behind a C ABI
(include/ssimu2_hip.h, include/oavif_tq.h).  libavif/libaom encode and decode stay on
the CPU and are the caller's.
"""
from . import _lib  # noqa: F401
from .scorer import Ssimu2, Ssimu2Error, query_device, score_many, version  # noqa: F401
from . import tq  # noqa: F401

__all__ = ["Ssimu2", "Ssimu2Error", "query_device", "score_many", "version", "tq"]
