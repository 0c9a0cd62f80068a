"""Host-side mirror of the scorer interface the reference binds at tq.zig:37.

    fssimu2.computeSsimu2(allocator, ref, dist, w, h, 3, null) -> f64      (tq.zig:37)

`Ssimu2.compute_ssimu2(ref, dist)` is that call on the MI355X; errors surface as
`Ssimu2Error` carrying the C ABI's negative code (the Zig error-union counterpart).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


class Ssimu2Error(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ssimu2 error {code}: {msg}")
        self.code = code


def _u8p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _check_rgb8(a, name: str) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"{name} must be (h, w, 3) uint8, got {a.shape}")
    return a


def query_device(device: int = 0, instrumented: bool = False) -> dict:
    """ssimu2_query_device: what the library reads off HIP device `device` without creating a context
    (`usable` False = ssimu2_ctx_create would refuse it: not gfx950, or less than 160 KB of LDS per CU)."""
    L = _lib.instr_lib() if instrumented else _lib.lib()
    d = _lib.DeviceInfo()
    rc = L.ssimu2_query_device(int(device), ctypes.byref(d))
    if rc != 0:
        raise Ssimu2Error(rc, L.ssimu2_last_error(None).decode() or "ssimu2_query_device failed")
    return d.as_dict()


class _CtxHolder:
    """The native context and who still needs it: the Ssimu2 object that created it (`owner`) and every page-locked buffer
    handed out by host_alloc whose numpy array is still referenced.  ssimu2_ctx_destroy runs when the owner has closed AND
    the last such array is gone -- closing a scorer never unmaps memory a live array points into."""

    def __init__(self, L, ctx):
        self.L, self.ctx, self.owner, self.buffers = L, ctx, True, 0

    def release_owner(self):
        self.owner = False
        self._maybe_destroy()

    def buffer_gone(self):
        self.buffers -= 1
        self._maybe_destroy()

    def _maybe_destroy(self):
        if not self.owner and self.buffers <= 0 and self.ctx is not None and self.ctx.value:
            self.L.ssimu2_ctx_destroy(self.ctx)
            self.ctx = None


class _PinnedBuffer:
    """One allocation of ssimu2_host_alloc as the base object of the numpy array handed to the caller: every view of that
    array keeps this object alive, and the memory is returned (ssimu2_host_free) when the last of them is gone or when the
    caller says host_free.  Freed exactly once."""

    def __init__(self, holder: _CtxHolder, ptr: int, nbytes: int):
        self.holder, self.ptr, self.nbytes, self.freed = holder, ptr, nbytes, False
        holder.buffers += 1
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def free(self) -> int:
        if self.freed:
            return 0
        self.freed = True
        rc = 0
        if self.holder.ctx is not None and self.holder.ctx.value:
            rc = self.holder.L.ssimu2_host_free(self.holder.ctx, ctypes.c_void_p(self.ptr))
        self.holder.buffer_gone()
        return rc

    def __del__(self):
        try:
            self.free()
        except Exception:   # interpreter shutdown
            pass


class Ssimu2:
    """One scorer context = one HIP stream + device scratch (not re-entrant)."""

    def __init__(self, device: int = 0, stream: int | None = None, instrumented: bool = False,
                 blur: int | None = None):
        """`instrumented=True` binds liboavif_hip_instr.so (the hooks of
        include/ssimu2_hip_internal.h: stage timing, plane download, experiment knobs); the
        default is the product library, which has none of them.  `blur`: _lib.BLUR_FIR /
        _lib.BLUR_RECURSIVE (ssimu2_ctx_set_blur); None = FIR."""
        self.instrumented = bool(instrumented)
        self._L = _lib.instr_lib() if instrumented else _lib.lib()
        self._ctx = ctypes.c_void_p()
        self._holder = None
        rc = self._L.ssimu2_ctx_create(int(device), ctypes.c_void_p(stream or 0),
                                       ctypes.byref(self._ctx))
        if rc != 0:
            msg = self._L.ssimu2_last_error(None).decode()
            self._ctx = ctypes.c_void_p()
            raise Ssimu2Error(rc, msg or "ssimu2_ctx_create failed")
        self._holder = _CtxHolder(self._L, ctypes.c_void_p(self._ctx.value))
        self._pinned = {}   # data address -> weak reference to the _PinnedBuffer behind a host_alloc array
        self.device = device
        if blur is not None and int(blur) != _lib.BLUR_FIR:
            self.set_blur(blur)

    def close(self) -> None:
        """The context is unusable from here on.  Page-locked buffers of host_alloc whose arrays are still referenced stay
        mapped -- and the native context alive underneath them -- until the last such array is dropped."""
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._ctx = ctypes.c_void_p()
            self._pinned = {}
            self._holder.release_owner()

    def device_info(self) -> dict:
        """ssimu2_ctx_device_info: the record the context was created with (arch, LDS per CU, PCI bus id, ...)."""
        d = _lib.DeviceInfo()
        rc = self._L.ssimu2_ctx_device_info(self._ctx, ctypes.byref(d))
        if rc != 0:
            self._raise(rc)
        return d.as_dict()

    def host_alloc(self, shape) -> np.ndarray:
        """ssimu2_host_alloc: a uint8 array of `shape` in page-locked host memory (uploads from it skip the HIP runtime's
        staging copy).  The memory belongs to the array: it is returned when the array and all its views are gone, or at
        host_free(array) -- after which the caller must not touch the array again.  Closing the scorer first is safe (the
        memory stays mapped while an array points into it).  ssimu2_host_free synchronises the whole device: allocate
        once per context and reuse, never per score."""
        shape = tuple(int(x) for x in (shape if hasattr(shape, "__len__") else (shape,)))
        n = int(np.prod(shape))
        ptr = ctypes.c_void_p()
        rc = self._L.ssimu2_host_alloc(self._ctx, n, ctypes.byref(ptr))
        if rc != 0:
            self._raise(rc)
        buf = _PinnedBuffer(self._holder, ptr.value, n)
        a = np.asarray(buf).reshape(shape)      # a.base chain ends at `buf`
        import weakref
        self._pinned[a.ctypes.data] = weakref.ref(buf)
        return a

    def host_free(self, a: np.ndarray) -> None:
        """Return a host_alloc buffer now (a device-wide synchronisation, see host_alloc).  `a` and its views must not be
        used afterwards."""
        ref = getattr(self, "_pinned", {}).pop(a.ctypes.data, None)
        buf = ref() if ref is not None else None
        if buf is None or buf.freed:
            raise ValueError("not a (live) buffer of this context's host_alloc")
        rc = buf.free()
        if rc != 0:
            self._raise(rc)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _raise(self, rc: int):
        raise Ssimu2Error(rc, self._L.ssimu2_last_error(self._ctx).decode())

    # -- host-buffer entry points ---------------------------------------------------------
    def compute_ssimu2(self, ref, dist, channels: int = 3) -> float:
        ref = _check_rgb8(ref, "ref")
        dist = _check_rgb8(dist, "dist")
        if ref.shape != dist.shape:
            raise ValueError("ref and dist must have the same shape")
        h, w, _ = ref.shape
        out = ctypes.c_double()
        rc = self._L.ssimu2_score_rgb8(self._ctx, _u8p(ref), _u8p(dist), w, h, channels,
                                       ctypes.byref(out))
        if rc != 0:
            self._raise(rc)
        return out.value

    def set_reference(self, ref) -> None:
        ref = _check_rgb8(ref, "ref")
        h, w, _ = ref.shape
        rc = self._L.ssimu2_set_reference(self._ctx, _u8p(ref), w, h)
        if rc != 0:
            self._raise(rc)
        self._ref_shape = ref.shape

    def score_against_reference(self, dist) -> float:
        dist = _check_rgb8(dist, "dist")
        if getattr(self, "_ref_shape", None) is not None and dist.shape != self._ref_shape:
            raise ValueError("dist shape differs from the reference's")
        out = ctypes.c_double()
        rc = self._L.ssimu2_score_against_reference(self._ctx, _u8p(dist), ctypes.byref(out))
        if rc != 0:
            self._raise(rc)
        return out.value

    def score_decoded_against_reference(self, pixels, row_bytes: int | None = None,
                                        channels: int | None = None) -> float:
        """Score a decoded frame in libavif's avifRGBImage layout without the CPU copy loop of
        io.decodeAvifToRgb (io.zig:654-663): `pixels` is an (h, w, 3|4) uint8 array whose rows
        may be padded (strides[0] >= w * channels, pixels tightly packed within a row), or a flat
        uint8 buffer with explicit `row_bytes` / `channels`.  Alpha and padding are dropped on
        the device."""
        a = np.asarray(pixels)
        if a.dtype != np.uint8:
            raise TypeError("pixels must be uint8")
        if a.ndim == 3:
            if a.strides[2] != 1 or a.strides[1] != a.shape[2]:
                raise ValueError("pixels of a row must be tightly packed")
            if getattr(self, "_ref_shape", None) is not None and a.shape[:2] != self._ref_shape[:2]:
                raise ValueError("frame size differs from the reference's")
            row_bytes = a.strides[0] if row_bytes is None else row_bytes
            channels = a.shape[2] if channels is None else channels
        elif row_bytes is None or channels is None:
            raise ValueError("flat buffers need row_bytes and channels")
        out = ctypes.c_double()
        ptr = ctypes.cast(ctypes.c_void_p(a.ctypes.data), ctypes.POINTER(ctypes.c_uint8))
        rc = self._L.ssimu2_score_against_reference_strided(self._ctx, ptr, int(row_bytes),
                                                            int(channels), ctypes.byref(out))
        if rc != 0:
            self._raise(rc)
        return out.value

    # -- device-resident entry points (pointers are raw device addresses) -------------------
    def score_device(self, d_ref: int, d_dist: int, w: int, h: int) -> float:
        out = ctypes.c_double()
        rc = self._L.ssimu2_score_rgb8_device(self._ctx, ctypes.c_void_p(d_ref),
                                              ctypes.c_void_p(d_dist), w, h, ctypes.byref(out))
        if rc != 0:
            self._raise(rc)
        return out.value

    def enqueue_device(self, d_ref: int, d_dist: int, w: int, h: int) -> None:
        rc = self._L.ssimu2_enqueue_rgb8_device(self._ctx, ctypes.c_void_p(d_ref),
                                                ctypes.c_void_p(d_dist), w, h)
        if rc != 0:
            self._raise(rc)

    def set_reference_device(self, d_ref: int, w: int, h: int) -> None:
        rc = self._L.ssimu2_set_reference_device(self._ctx, ctypes.c_void_p(d_ref), w, h)
        if rc != 0:
            self._raise(rc)
        self._ref_shape = (h, w, 3)

    def enqueue_against_reference_device(self, d_dist: int) -> None:
        rc = self._L.ssimu2_enqueue_against_reference_device(self._ctx, ctypes.c_void_p(d_dist))
        if rc != 0:
            self._raise(rc)

    def wait(self) -> float:
        out = ctypes.c_double()
        rc = self._L.ssimu2_wait(self._ctx, ctypes.byref(out))
        if rc != 0:
            self._raise(rc)
        return out.value

    # -- measurement / parity hooks: instrumented build only ------------------------------------
    def _need_instr(self):
        if not self.instrumented:
            raise RuntimeError("this hook needs Ssimu2(..., instrumented=True) (liboavif_hip_instr.so)")

    def time_device(self, d_ref: int, d_dist: int, w: int, h: int, iters: int):
        """-> (total device ms for `iters` back-to-back scores, score)."""
        self._need_instr()
        ms = ctypes.c_float()
        out = ctypes.c_double()
        rc = self._L.ssimu2_time_device(self._ctx, ctypes.c_void_p(d_ref), ctypes.c_void_p(d_dist),
                                        w, h, iters, ctypes.byref(ms), ctypes.byref(out))
        if rc != 0:
            self._raise(rc)
        return ms.value, out.value

    def time_stage(self, d_ref: int, d_dist: int, w: int, h: int, stage: int, iters: int) -> float:
        """-> average device ms of one execution of `stage` (_lib.STAGE_*) of the score."""
        self._need_instr()
        ms = ctypes.c_float()
        rc = self._L.ssimu2_time_stage(self._ctx, ctypes.c_void_p(d_ref), ctypes.c_void_p(d_dist),
                                       w, h, stage, iters, ctypes.byref(ms))
        if rc != 0:
            self._raise(rc)
        return ms.value

    def measure_read_stream(self, nbytes: int = 2 << 30, iters: int = 10) -> float:
        """-> measured HBM read-stream bandwidth of the device in GB/s (ssimu2_measure_read_stream)."""
        self._need_instr()
        out = ctypes.c_double()
        rc = self._L.ssimu2_measure_read_stream(self._ctx, ctypes.c_size_t(nbytes), iters,
                                                ctypes.byref(out))
        if rc != 0:
            self._raise(rc)
        return out.value

    def set_blur(self, mode: int) -> None:
        """ssimu2_ctx_set_blur: _lib.BLUR_FIR (default, the fused 9-tap kernels) or
        _lib.BLUR_RECURSIVE (the published recursion, operation for operation: 0.4 ms per 4K pass
        against a cached reference where the default takes 0.16)."""
        rc = self._L.ssimu2_ctx_set_blur(self._ctx, int(mode))
        if rc != 0:
            self._raise(rc)

    def placed_streams(self) -> int:
        """Instrumented build: streams on distinct hardware queues its library instance holds for this
        context's device (ssimu2_instr_placed_streams)."""
        self._need_instr()
        n = ctypes.c_int(0)
        rc = self._L.ssimu2_instr_placed_streams(self._ctx, ctypes.byref(n))
        if rc != 0:
            self._raise(rc)
        return int(n.value)

    def use_graph(self, enabled: bool | None = None):
        """Instrumented build, the hipGraph experiment (ssimu2_instr_use_graph): submit every score of this context as one
        graph launch.  None = only read the counters.  -> (graphs built, graph launches) so far."""
        self._need_instr()
        b, n = ctypes.c_ulonglong(), ctypes.c_ulonglong()
        rc = self._L.ssimu2_instr_use_graph(self._ctx, -1 if enabled is None else int(bool(enabled)), ctypes.byref(b), ctypes.byref(n))
        if rc != 0:
            self._raise(rc)
        return int(b.value), int(n.value)

    def rg_stop_after_scale(self, scale: int) -> None:
        """Instrumented build: the recursive mode keeps the 15 raw planes of `scale` (after the
        horizontal pass and after both passes) downloadable (debug_download what = 4 / 5);
        negative = keep nothing.  (The name is round 2's, when the run stopped after that scale.)"""
        self._need_instr()
        rc = self._L.ssimu2_instr_rg_stop_after_scale(self._ctx, int(scale))
        if rc != 0:
            self._raise(rc)

    def debug_download(self, what: int, scale: int, w: int, h: int) -> np.ndarray:
        """-> (3, h_s, w_s) float32 planes; (15, h_s, w_s) for what = 4 / 5 (see ssimu2_debug_download)."""
        self._need_instr()
        sw, sh = w, h
        for _ in range(scale):
            sw, sh = (sw + 1) // 2, (sh + 1) // 2
        out = np.empty((15 if what in (4, 5) else 3, sh, sw), np.float32)
        ow, oh = ctypes.c_uint32(), ctypes.c_uint32()
        rc = self._L.ssimu2_debug_download(self._ctx, what, scale, w, h,
                                           out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                           ctypes.byref(ow), ctypes.byref(oh))
        if rc != 0:
            self._raise(rc)
        assert (ow.value, oh.value) == (sw, sh)
        return out

    def time_march_rotating(self, d_refs, d_dists, w: int, h: int, iters: int) -> float:
        """-> average device ms of the marching kernel over `iters` launches that rotate over the
        device-resident pairs (d_refs[i], d_dists[i]) -- inputs from HBM, not from the Infinity Cache."""
        self._need_instr()
        n = len(d_refs)
        assert n == len(d_dists) and n > 0
        arr = ctypes.c_void_p * n
        ms = ctypes.c_float()
        rc = self._L.ssimu2_time_march_rotating(self._ctx, arr(*d_refs), arr(*d_dists), n, w, h, iters,
                                                ctypes.byref(ms))
        if rc != 0:
            self._raise(rc)
        return ms.value

    def time_blur_stage_rotating(self, d_frames, w: int, h: int, iters: int):
        """-> (average device ms, algorithmic bytes per launch) of k_ref_blur -- the marching body as a
        plain blur stage: XYB planes of one frame in, one blurred plane per channel out -- over
        `iters` launches rotating over the plane sets of the device-resident frames (HBM-fed)."""
        self._need_instr()
        n = len(d_frames)
        arr = ctypes.c_void_p * n
        ms, nbytes = ctypes.c_float(), ctypes.c_double()
        rc = self._L.ssimu2_time_blur_stage_rotating(self._ctx, arr(*d_frames), n, w, h, iters,
                                                     ctypes.byref(ms), ctypes.byref(nbytes))
        if rc != 0:
            self._raise(rc)
        return ms.value, nbytes.value

    KERNEL_NAMES = {(False, True): ("pyramid", "march_refblur", "finalize"), (False, False): ("pyramid", "march", "finalize"),
                    (True, True): ("convert", "h", "v", "finalize"),
                    (True, False): ("ref_convert", "ref_h", "ref_v_emit", "convert", "h", "v", "finalize")}

    def time_kernels(self, w: int, h: int, d_dists, iters: int, d_ref: int | None = None, d_refs=None, recursive: bool = False):
        """ssimu2_time_kernels (instrumented build): every kernel of a score timed where it runs, from its own dispatch packet.
        `d_ref` = reference-cached passes over the distorted frames `d_dists`; `d_refs` = pair scores of (d_refs[i], d_dists[i]).
        -> ({launch name: average device ms}, stream ms per score with the timestamps, stream ms per score without).
        `recursive` names the launches only (the context's blur mode decides what runs)."""
        self._need_instr()
        n = len(d_dists)
        arr = ctypes.c_void_p * n
        ms = (ctypes.c_float * 8)()
        nl = ctypes.c_int()
        wt, wp = ctypes.c_float(), ctypes.c_float()
        rc = self._L.ssimu2_time_kernels(self._ctx, ctypes.c_void_p(d_ref or 0), arr(*d_refs) if d_refs is not None else None,
                                         arr(*d_dists), n, w, h, iters, ms, ctypes.byref(nl), ctypes.byref(wt), ctypes.byref(wp))
        if rc != 0:
            self._raise(rc)
        if d_refs is None:
            self._ref_shape = (h, w, 3)
        names = self.KERNEL_NAMES[(bool(recursive), d_refs is None)]
        if nl.value != len(names):   # e.g. a frame with one scale has no pyramid launch
            names = tuple(f"launch{k}" for k in range(nl.value))
        return dict(zip(names, (float(ms[k]) for k in range(nl.value)))), float(wt.value), float(wp.value)

    def set_segment_rows(self, rows_scale0: int, rows_other_scales: int) -> None:
        self._need_instr()
        rc = self._L.ssimu2_instr_set_segment_rows(self._ctx, rows_scale0, rows_other_scales)
        if rc != 0:
            self._raise(rc)

    def cache_reference_blur(self, enabled: bool) -> None:
        self._need_instr()
        rc = self._L.ssimu2_instr_cache_reference_blur(self._ctx, 1 if enabled else 0)
        if rc != 0:
            self._raise(rc)

    def last_averages(self):
        """-> ((6, 18) float64 plane averages of the last score, number of scales)."""
        avg = np.zeros(_lib.NUM_SCALES * _lib.STATS_PER_SCALE, np.float64)
        ns = ctypes.c_int()
        rc = self._L.ssimu2_last_averages(self._ctx,
                                          avg.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                          ctypes.byref(ns))
        if rc != 0:
            self._raise(rc)
        return avg.reshape(_lib.NUM_SCALES, _lib.STATS_PER_SCALE), ns.value


def score_many(scorers, d_ref: int, d_dists, w: int, h: int):
    """Score several distorted frames against one reference, fanned over `scorers`
    (independent contexts = independent HIP streams on one or more GPUs).  All pointers are
    device addresses valid on each scorer's device.  Returns the scores in input order; each is
    bit-identical to scoring that frame alone (a context never shares state with another).

    This is the mechanism for speculative quantizer probes: the decoded frames
    of several candidate quantizers are scored concurrently, then tq replays its sequential
    decision logic over the cached scores.
    """
    out = [None] * len(d_dists)
    n = len(scorers)
    for base in range(0, len(d_dists), n):
        batch = list(range(base, min(base + n, len(d_dists))))
        for j, i in enumerate(batch):
            scorers[j].enqueue_device(d_ref, d_dists[i], w, h)
        for j, i in enumerate(batch):
            out[i] = scorers[j].wait()
    return out


def version() -> str:
    return _lib.lib().ssimu2_version().decode()
