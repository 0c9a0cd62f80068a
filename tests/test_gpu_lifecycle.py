"""Life cycle of scorer contexts on the device: what a context allocates comes back when it is destroyed (or,
for the recursive modes' planes, when the context returns to SSIMU2_BLUR_FIR, as include/ssimu2_hip.h says), over
many create / grow / switch / destroy cycles -- a batch host creates and drops contexts for hours.  -m gpu only."""
import numpy as np
import pytest

import oavif_amd
from oavif_amd import _lib, synth

pytestmark = pytest.mark.gpu
MB = 1 << 20
SLACK = 128 * MB  # the HIP runtime grows its own arenas in steps of tens of MB now and then (46 MB seen once in 12 cycles);
                  # a context that leaked would cost 60 MB (FIR buffers) or 230 MB (recursive planes) EVERY cycle: 0.7-2.8 GB by the end


def _free():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_contexts_give_their_device_memory_back(hip_lib):
    small = synth.make_ref(640, 360, 1)
    big = synth.make_ref(1920, 1080, 2)
    d_small, d_big = synth.distort(small, "blockq", 2), synth.distort(big, "noise", 2)
    with oavif_amd.Ssimu2(0) as warm:      # the process-wide pieces (code object, stream pool, the runtime's own
        warm.compute_ssimu2(small, d_small)   # pools for every kernel of both modes) exist before the baseline
        warm.compute_ssimu2(big, d_big)
        warm.set_blur(_lib.BLUR_RECURSIVE)
        warm.set_reference(big)
        warm.score_against_reference(d_big)
    base = _free()
    want = None
    for cycle in range(12):
        s = oavif_amd.Ssimu2(0)
        a = s.compute_ssimu2(small, d_small)                       # FIR, small frame
        b = s.compute_ssimu2(big, d_big)                           # the buffers grow to the largest frame seen
        fir_alive = base - _free()
        s.set_blur(_lib.BLUR_RECURSIVE)
        s.set_reference(big)
        c = s.score_against_reference(d_big)
        rec_alive = base - _free()
        s.set_blur(_lib.BLUR_FIR)                                  # frees the recursive modes' planes
        back_to_fir = base - _free()
        with pytest.raises(oavif_amd.scorer.Ssimu2Error) as err:   # "a cached reference is dropped" (ssimu2_hip.h)
            s.score_against_reference(d_big)
        assert err.value.code == -5 if hasattr(err.value, "code") else "no reference" in str(err.value)
        s.set_reference(big)
        assert s.score_against_reference(d_big) == s.compute_ssimu2(big, d_big) == b
        s.close()
        after = base - _free()
        if want is None:
            want = (a, b, c)
        assert (a, b, c) == want                                   # and the scores never move
        # 1080p: the recursive modes hold 21 padded planes x 1.333 scales = ~0.23 GB on top of the FIR buffers
        assert 150 * MB < rec_alive - fir_alive < 330 * MB, (fir_alive / MB, rec_alive / MB)
        assert abs(back_to_fir - fir_alive) <= SLACK, (back_to_fir / MB, fir_alive / MB)
        assert abs(after) <= SLACK, f"cycle {cycle}: {after / MB:.1f} MB not returned"


def test_many_contexts_at_once_and_out_of_order_destruction(hip_lib):
    ref = synth.make_ref(512, 384, 3)
    dist = synth.distort(ref, "blur", 1)
    with oavif_amd.Ssimu2(0) as s0:
        want = s0.compute_ssimu2(ref, dist)
    base = _free()
    ctxs = [oavif_amd.Ssimu2(0, blur=_lib.BLUR_RECURSIVE if i % 2 else None) for i in range(16)]   # OAVIF_TQ_MAX_FANOUT
    for c in ctxs:
        c.set_reference(ref)
    got = [c.score_against_reference(dist) for c in ctxs]
    assert all(g == want for g in got[0::2])                       # FIR contexts: the pair score's bits
    assert len(set(got[1::2])) == 1 and abs(got[1] - want) < 0.5   # recursive contexts agree among themselves
    for i in (5, 0, 15, 7, 8, 1, 2, 14, 3, 13, 4, 12, 6, 11, 9, 10):
        ctxs[i].close()
    assert abs(base - _free()) <= SLACK
