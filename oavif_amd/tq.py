"""Host-side mirror of /root/reference/src/tq.zig over the C ABI (include/oavif_tq.h).

`find_target_quality` is `tq.findTargetQuality` (tq.zig:124-210) with the pass
(`computeScoreAtQuality`, tq.zig:21-38) injected as `probe(q) -> score`;
`search_hip` binds the scorer half of each pass to the MI355X scorer and takes the CPU
codec (encode at q -> decode -> RGB8) as a callback.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Callable, List, Tuple

import numpy as np

from . import _lib
from .scorer import Ssimu2, Ssimu2Error


@dataclass
class TQResult:
    q: int = 0
    score: float = 0.0
    num_pass: int = 0
    buf_q: int = -1
    history: List[Tuple[int, float]] = field(default_factory=list)
    last_avif_size: int = 0


def _options(score_tgt: float, tolerance: float, max_pass: int) -> _lib.TQOptions:
    return _lib.TQOptions(float(score_tgt), float(tolerance), int(max_pass))


def _result(r: _lib.TQResult) -> TQResult:
    return TQResult(q=int(r.q), score=float(r.score), num_pass=int(r.num_pass), buf_q=int(r.buf_q),
                    history=[(int(r.history[i].q), float(r.history[i].score))
                             for i in range(r.history_len)])


def predict_q_from_score(tgt: float) -> int:
    return int(_lib.lib().oavif_tq_predict_q_from_score(float(tgt)))


def interpolate_quantizer(lo: int, hi: int, history, target: float) -> int:
    n = len(history)
    arr = (_lib.TQPass * max(n, 1))()
    for i, (q, s) in enumerate(history):
        arr[i].q, arr[i].score = int(q), float(s)
    return int(_lib.lib().oavif_tq_interpolate_quantizer(lo, hi, arr, n, float(target)))


def find_target_quality(probe: Callable[[int], float], score_tgt: float = 80.0,
                        tolerance: float = 2.0, max_pass: int = 6) -> TQResult:
    L = _lib.lib()
    err: list = []

    def _cb(_user, q, out):
        try:
            out[0] = float(probe(int(q)))
            return 0
        except Exception as e:  # surfaces like the Zig `try` at tq.zig:150
            err.append(e)
            return -100

    cb = _lib.PROBE_FN(_cb)
    res = _lib.TQResult()
    opts = _options(score_tgt, tolerance, max_pass)
    rc = L.oavif_tq_find_target_quality(ctypes.byref(opts), cb, None, ctypes.byref(res))
    if err:
        raise err[0]
    if rc != 0:
        raise Ssimu2Error(rc, "oavif_tq_find_target_quality failed")
    return _result(res)


@dataclass
class SpecStats:
    waves: int = 0           # batches of concurrent probes = the latency of the search in passes
    probes_issued: int = 0   # quantizers encoded + scored in total
    cache_hits: int = 0      # passes answered by an earlier wave


def find_target_quality_speculative(batch_probe: Callable[[List[int]], List[float]],
                                    score_tgt: float = 80.0, tolerance: float = 2.0,
                                    max_pass: int = 6, max_fanout: int = 4, first_wave_fanout: int = 0):
    """`findTargetQuality` with several probes in flight (include/oavif_tq.h, "speculative probe
    fan-out").  batch_probe(qs) -> scores probes the quantizers of one wave, qs[0] being the one
    the search waits for.  `first_wave_fanout`: probes of the first wave (0 = max_fanout; 1 = the
    model's guess alone, so a one-pass search costs what the sequential search costs).
    -> (TQResult, SpecStats); the TQResult equals the sequential one."""
    L = _lib.lib()
    err: list = []

    def _cb(_user, qs, n, out):
        try:
            want = [int(qs[i]) for i in range(n)]
            got = list(batch_probe(want))
            if len(got) != n:
                raise ValueError(f"batch_probe returned {len(got)} scores for {n} quantizers")
            for i in range(n):
                out[i] = float(got[i])
            return 0
        except Exception as e:
            err.append(e)
            return -100

    cb = _lib.BATCH_PROBE_FN(_cb)
    res = _lib.TQResult()
    stats = _lib.TQSpecStats()
    opts = _options(score_tgt, tolerance, max_pass)
    so = _lib.TQSpecOptions(int(max_fanout), int(first_wave_fanout))
    rc = L.oavif_tq_find_target_quality_speculative(ctypes.byref(opts), ctypes.byref(so), cb, None,
                                                    ctypes.byref(res), ctypes.byref(stats))
    if err:
        raise err[0]
    if rc != 0:
        raise Ssimu2Error(rc, "oavif_tq_find_target_quality_speculative failed")
    return _result(res), SpecStats(int(stats.waves), int(stats.probes_issued), int(stats.cache_hits))


def search_speculative_hip(scorers, ref_rgb: np.ndarray,
                           codec: Callable[[int], Tuple[np.ndarray, int]], score_tgt: float = 80.0,
                           tolerance: float = 2.0, max_pass: int = 6, max_fanout: int | None = None,
                           first_wave_fanout: int = 1):
    """One search with its probes fanned over `scorers` (one context = one HIP stream each, same
    device) and as many host threads: every probe of a wave runs codec(q) -- the CPU encode +
    decode -- and scores its frame on its own context, so the GPU work of one probe overlaps the
    CPU work of the others (BASELINE configs[2]).  The first wave is the model's guess alone by
    default (`first_wave_fanout=1`): searches that end on their first pass -- most, on typical
    content -- then cost exactly a sequential search, and speculation starts with the second wave.
    -> (TQResult, SpecStats, {q: avif size})."""
    from concurrent.futures import ThreadPoolExecutor
    scorers = list(scorers)
    if not scorers:
        raise ValueError("need at least one scorer context")
    fan = min(len(scorers), _lib.TQ_MAX_FANOUT) if max_fanout is None else int(max_fanout)
    if fan > len(scorers):
        raise ValueError("max_fanout exceeds the number of scorer contexts")
    ref = np.ascontiguousarray(ref_rgb, dtype=np.uint8)
    have_ref = [False] * fan   # a context gets the reference when a wave first uses it: a search that
    sizes: dict = {}           # ends on its first (one-probe) wave uploads it once, like the sequential one

    def one(args):
        slot, q = args
        if not have_ref[slot]:          # a slot is used by one thread at a time
            scorers[slot].set_reference(ref)
            have_ref[slot] = True
        dec, size = codec(q)
        dec = np.ascontiguousarray(dec, dtype=np.uint8)
        if dec.shape != ref.shape:
            raise ValueError(f"codec returned {dec.shape}, expected {ref.shape}")
        sizes[q] = int(size)
        return scorers[slot].score_against_reference(dec)

    with ThreadPoolExecutor(max_workers=fan) as pool:
        def batch(qs):
            return list(pool.map(one, list(enumerate(qs))))

        res, stats = find_target_quality_speculative(batch, score_tgt, tolerance, max_pass, fan,
                                                     min(int(first_wave_fanout), fan))
    res.last_avif_size = sizes.get(res.buf_q, 0)
    return res, stats, sizes


def search_hip(scorer: Ssimu2, ref_rgb: np.ndarray,
               codec: Callable[[int], Tuple[np.ndarray, int]], score_tgt: float = 80.0,
               tolerance: float = 2.0, max_pass: int = 6) -> TQResult:
    """codec(q) -> (decoded (h, w, 3) uint8, avif size in bytes): the CPU encode+decode."""
    L = _lib.lib()
    ref = np.ascontiguousarray(ref_rgb, dtype=np.uint8)
    h, w, _ = ref.shape
    nbytes = w * h * 3
    err: list = []

    def _cb(_user, q, out_rgb, out_size):
        try:
            dec, size = codec(int(q))
            dec = np.ascontiguousarray(dec, dtype=np.uint8)
            if dec.shape != ref.shape:
                raise ValueError(f"codec returned {dec.shape}, expected {ref.shape}")
            ctypes.memmove(out_rgb, dec.ctypes.data, nbytes)
            out_size[0] = int(size)
            return 0
        except Exception as e:
            err.append(e)
            return -100

    cb = _lib.CODEC_FN(_cb)
    res = _lib.TQResult()
    last = ctypes.c_size_t()
    opts = _options(score_tgt, tolerance, max_pass)
    rc = L.oavif_tq_search_hip(ctypes.byref(opts), scorer._ctx,
                               ref.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), w, h, cb, None,
                               ctypes.byref(res), ctypes.byref(last))
    if err:
        raise err[0]
    if rc != 0:
        raise Ssimu2Error(rc, L.ssimu2_last_error(scorer._ctx).decode())
    out = _result(res)
    out.last_avif_size = int(last.value)
    return out


def search_hip_frames(scorer: Ssimu2, ref_rgb: np.ndarray, codec_frame, score_tgt: float = 80.0,
                      tolerance: float = 2.0, max_pass: int = 6) -> TQResult:
    """The search with the decoded-frame hand-off: codec_frame(q) -> (frame, avif size)
    where `frame` is what decodeAvifCommon leaves behind (oavif_amd.avif_bridge.DecodedFrame: libavif's own
    8-bit RGB or RGBA rows, `rows` / `row_bytes` / `channels`, closed here after the score).  The rows go to
    the device as they are (`ssimu2_score_against_reference_strided`); the alpha-dropping copy loop of
    io.decodeAvifToRgb (io.zig:654-663) never runs on the host.  Same control flow, q and scores as
    search_hip (the device unpacks to the same tight RGB8)."""
    ref = np.ascontiguousarray(ref_rgb, dtype=np.uint8)
    scorer.set_reference(ref)
    last = {"size": 0}

    def probe(q: int) -> float:
        frame, size = codec_frame(int(q))
        try:
            if (frame.height, frame.width) != ref.shape[:2]:
                raise ValueError(f"codec returned {frame.width}x{frame.height}, expected {ref.shape[1]}x{ref.shape[0]}")
            score = scorer.score_decoded_against_reference(frame.rows.reshape(-1), frame.row_bytes, frame.channels)
        finally:
            frame.close()
        last["size"] = int(size)
        return score

    out = find_target_quality(probe, score_tgt, tolerance, max_pass)
    out.last_avif_size = last["size"]
    return out


def prescale(src: np.ndarray, out_depth: int) -> np.ndarray:
    """The source rescaled to the encoder's depth as io.encodeAvifToBuffer does on every pass
    (io.zig:566-617), to be computed once per search (include/oavif_tq.h).  src uint8 or uint16
    (full 16-bit range); out_depth 8 or 10."""
    L = _lib.lib()
    a = np.ascontiguousarray(src)
    u8p, u16p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint16)
    if a.dtype == np.uint8 and out_depth == 10:
        out = np.empty(a.shape, np.uint16)
        L.oavif_prescale_8_to_10(a.ctypes.data_as(u8p), a.size, out.ctypes.data_as(u16p))
    elif a.dtype == np.uint16 and out_depth == 10:
        out = np.empty(a.shape, np.uint16)
        L.oavif_prescale_16_to_10(a.ctypes.data_as(u16p), a.size, out.ctypes.data_as(u16p))
    elif a.dtype == np.uint16 and out_depth == 8:
        out = np.empty(a.shape, np.uint8)
        L.oavif_prescale_16_to_8(a.ctypes.data_as(u16p), a.size, out.ctypes.data_as(u8p))
    elif a.dtype == np.uint8 and out_depth == 8:
        out = a  # io.zig:609-612: the source is handed over as it is
    else:
        raise ValueError("src must be uint8 or uint16, out_depth 8 or 10")
    return out
