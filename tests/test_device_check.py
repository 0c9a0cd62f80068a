"""Device capability check at context creation and page-locked frame buffers (include/ssimu2_hip.h:
ssimu2_query_device / ssimu2_ctx_device_info / ssimu2_host_alloc / ssimu2_host_free).

The library holds gfx950 code only and its recursive-mode vertical pass keeps one workgroup per compute unit by
asking for more than half of a 160 KB LDS: ssimu2_ctx_create checks both before anything is allocated and answers
SSIMU2_ERR_NO_DEVICE ("no usable gfx950 device", ssimu2_hip.h) with the reason, instead of failing at the first
launch.  The CPU part covers the argument contract and the no-GPU answer; the GPU part asserts the record the
context saw and that scores from pinned buffers are the scores from pageable ones."""
import ctypes

import numpy as np
import pytest

from oavif_amd import _lib, synth


def test_query_contract_without_touching_a_context(hip_lib):
    L = hip_lib
    d = _lib.DeviceInfo()
    assert ctypes.sizeof(_lib.DeviceInfo) == 264 and d.struct_size == 264      # the header's layout, no padding surprises
    assert L.ssimu2_query_device(0, None) == _lib.ERR_INVALID_ARG
    d.struct_size = 100                                                        # a caller built against another header
    assert L.ssimu2_query_device(0, ctypes.byref(d)) == _lib.ERR_INVALID_ARG
    assert L.ssimu2_ctx_device_info(None, ctypes.byref(d)) == _lib.ERR_INVALID_ARG
    assert L.ssimu2_host_alloc(None, 16, None) == _lib.ERR_INVALID_ARG
    assert L.ssimu2_host_free(None, None) == _lib.ERR_INVALID_ARG


def test_no_gpu_is_reported_as_no_device_with_a_reason(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    d = _lib.DeviceInfo()
    assert hip_lib.ssimu2_query_device(0, ctypes.byref(d)) == _lib.ERR_NO_DEVICE
    assert b"no usable HIP device" in hip_lib.ssimu2_last_error(None)
    ctx = ctypes.c_void_p()
    assert hip_lib.ssimu2_ctx_create(0, None, ctypes.byref(ctx)) == _lib.ERR_NO_DEVICE and not ctx.value
    assert b"no usable HIP device" in hip_lib.ssimu2_last_error(None)


def test_the_pad_of_the_vertical_pass_follows_the_lds_size():
    """ssimu2_recursive.h rg_v_pad_bytes: with the kernel's own 15 KB of tiles the launch asks for just over half
    of the CU's LDS, whatever that size is; the constant 70 KB of round 4 only fitted a 160 KB CU."""
    import re
    src = open(_lib._HERE + "/csrc/ssimu2_recursive.h").read()
    assert "RG_V_PAD_BYTES" not in src and "rg_v_pad_bytes(unsigned lds_bytes_per_cu)" in src
    n = int(re.search(r"RG_N = (\d+)", src).group(1))
    static = 2 * 3 * (2 * n) * 64 * 4
    for lds in (160 * 1024, 256 * 1024):
        pad = lds // 2 + 1024 - static
        assert pad + static > lds // 2 and pad + static + 1024 <= lds
    hip = open(_lib._HERE + "/csrc/ssimu2_hip.hip").read()
    assert "c->rg_v_pad = rg_v_pad_bytes(info.lds_bytes_per_cu)" in hip and "kMinLdsPerCu = 160u * 1024u" in hip


@pytest.mark.gpu
def test_context_records_the_device_it_checked(scorer):
    import oavif_amd
    info = scorer.device_info()
    assert info["arch"].startswith("gfx950"), info
    assert info["lds_bytes_per_cu"] >= 160 * 1024 and info["wavefront_size"] == 64 and info["usable"], info
    assert info["compute_units"] >= 200 and info["hbm_bytes"] > 200e9, info
    import re
    assert re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.[0-7]", info["pci_bus_id"]), info
    assert oavif_amd.query_device(0) == info                       # the same record without a context
    with pytest.raises(oavif_amd.Ssimu2Error) as e:
        oavif_amd.query_device(63)                                 # no such device
    assert e.value.code == _lib.ERR_NO_DEVICE
    with pytest.raises(oavif_amd.Ssimu2Error) as e:
        oavif_amd.Ssimu2(63)
    assert e.value.code == _lib.ERR_NO_DEVICE and "no usable HIP device" in str(e.value)


@pytest.mark.gpu
def test_pinned_buffers_score_like_pageable_ones(hip_lib, oracle):
    import oavif_amd
    w, h = 640, 360
    ref = synth.make_ref(w, h, 11)
    dst = synth.distort(ref, "blockq", 3)
    for blur in (_lib.BLUR_FIR, _lib.BLUR_RECURSIVE):
        with oavif_amd.Ssimu2(0, blur=blur) as s:
            want = s.compute_ssimu2(ref, dst)
            p_ref, p_dst = s.host_alloc(ref.shape), s.host_alloc(dst.shape)
            p_ref[...] = ref
            p_dst[...] = dst
            assert s.compute_ssimu2(p_ref, p_dst) == want
            s.set_reference(p_ref)
            assert s.score_against_reference(p_dst) == want
            rgba = s.host_alloc((h, w, 4))                        # libavif's RGBA rows in a caller-provided buffer
            rgba[..., :3] = dst
            rgba[..., 3] = 255
            assert s.score_decoded_against_reference(rgba) == want
            s.host_free(rgba)
            with pytest.raises(ValueError):
                s.host_free(rgba)                                  # already returned
            assert abs(want - oracle.compute_ssimu2(ref, dst, oracle.BLUR_FIR if blur == _lib.BLUR_FIR else oracle.BLUR_IIR)) <= 1e-3
        # the remaining buffers are released with the context


def test_pinned_arrays_outlive_the_scorer_that_made_them():
    """closing a scorer (close, __exit__, __del__) while a host_alloc array is still referenced must not unmap the
    memory under it.  The array owns its buffer (its base chain ends at the allocation), the native context is destroyed only
    when the scorer has closed AND the last such array is gone, and a buffer is returned exactly once.  CPU: the native calls
    are recorded by a stand-in library."""
    import ctypes
    import gc
    import numpy as np
    from oavif_amd import scorer

    class Lib:
        def __init__(self):
            self.freed, self.destroyed = [], []

        def ssimu2_host_free(self, ctx, p):
            assert not self.destroyed            # never after the context is gone
            self.freed.append(p.value)
            return 0

        def ssimu2_ctx_destroy(self, ctx):
            self.destroyed.append(ctx.value)

    mem = (ctypes.c_uint8 * 96)()
    lib = Lib()
    holder = scorer._CtxHolder(lib, ctypes.c_void_p(4242))
    a = np.asarray(scorer._PinnedBuffer(holder, ctypes.addressof(mem), 48)).reshape(4, 4, 3)
    b = np.asarray(scorer._PinnedBuffer(holder, ctypes.addressof(mem) + 48, 48))
    view = a[1:3]
    holder.release_owner()                       # Ssimu2.close()
    assert lib.destroyed == [] and lib.freed == []          # both arrays alive: nothing unmapped, the context stays
    a[...] = 7
    del a
    gc.collect()
    assert lib.freed == [] and int(view.sum()) == 7 * 24    # a view keeps the buffer
    del view
    gc.collect()
    assert lib.freed == [ctypes.addressof(mem)] and lib.destroyed == []
    base = b.base
    while not isinstance(base, scorer._PinnedBuffer):
        base = base.base
    assert base.free() == 0 and base.free() == 0            # explicit return (host_free), then nothing more to do
    assert lib.freed == [ctypes.addressof(mem), ctypes.addressof(mem) + 48] and lib.destroyed == [4242]
    del b, base
    gc.collect()
    assert len(lib.freed) == 2 and lib.destroyed == [4242]  # returned exactly once, destroyed exactly once
