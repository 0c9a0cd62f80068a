"""Stream placement (oavif_amd/csrc/ssimu2_hip.hip "stream placement"): contexts created without a
caller stream get streams on distinct hardware queues, so the scores of two contexts overlap
whichever two a caller creates.  -m gpu only."""
import time

import pytest

from oavif_amd import synth

pytestmark = pytest.mark.gpu


def _ms_per_score(group, pr, pd, w, h, n=240):
    for c in group:
        c.enqueue_device(pr, pd, w, h)
        c.wait()
    t = time.perf_counter()
    for i in range(n):
        group[i % len(group)].enqueue_device(pr, pd, w, h)
    for c in group:
        c.wait()
    return (time.perf_counter() - t) / n * 1e3


def test_contexts_created_back_to_back_get_distinct_queues(hip_lib):
    """What is ASSERTED is collision detection, not a margin: (i) the placed set holds at least two streams
    (three or four with HIP's default of four hardware queues: BENCH_r04 reports 4) -- asked of the instrumented
    library instance, which runs the same placement code; (ii) every pair among three product contexts created
    back to back scores faster on two contexts than 0.95 x the one-context time per score (a pair that shares
    a hardware queue measures 1.0; pairs on distinct queues 0.87-0.89, best of four runs each).  The ratio
    itself is a reported number -- bench.py's `two_context_ratio` -- not a test criterion: it varies by ~0.03
    between boxes of the pool (round 3's 0.9 assert sat 1-3 % from the measured values)."""
    import torch
    from oavif_amd import Ssimu2
    w, h = 3840, 2160
    ref = synth.make_ref(w, h, 0)
    dst = synth.distort(ref, "blockq", 2)
    tr, td = torch.from_numpy(ref).cuda(), torch.from_numpy(dst).cuda()
    pr, pd = tr.data_ptr(), td.data_ptr()
    with Ssimu2(0, instrumented=True) as probe:
        placed = probe.placed_streams()
    print(f"placed streams on distinct hardware queues: {placed}")
    assert placed >= 2, placed
    # three product contexts: the documented contract is "any two of the first three contexts of a process overlap"
    # (INTEGRATION.md section 5); the product instance's own set may be smaller than the instrumented instance's
    # (each instance probes for itself; with HIP's four hardware queues a fourth context may share one)
    n = 3 if placed >= 3 else 2
    ctx = [Ssimu2(0) for _ in range(n)]
    try:
        _ms_per_score(ctx[:2], pr, pd, w, h, 600)   # clocks
        one = min(_ms_per_score([c], pr, pd, w, h) for c in ctx for _ in range(2))
        for i in range(n):
            for j in range(i + 1, n):
                two = min(_ms_per_score([ctx[i], ctx[j]], pr, pd, w, h) for _ in range(4))
                print(f"contexts {i},{j}: {two:.4f} ms per score on two, {one:.4f} on one, ratio {two / one:.3f}")
                assert two <= 0.95 * one, (i, j, two, one)
        # the scores do not depend on the stream
        a, b = ctx[0].score_device(pr, pd, w, h), ctx[n - 1].score_device(pr, pd, w, h)
        assert a == b
    finally:
        for c in ctx:
            c.close()


def test_a_released_stream_is_reused_and_extra_contexts_still_work(hip_lib):
    from oavif_amd import Ssimu2
    ref = synth.make_ref(160, 96, 1)
    dst = synth.distort(ref, "noise", 2)
    with Ssimu2(0) as s:
        want = s.compute_ssimu2(ref, dst)
    many = [Ssimu2(0) for _ in range(7)]   # more contexts than placed streams
    try:
        assert all(c.compute_ssimu2(ref, dst) == want for c in many)
    finally:
        for c in many:
            c.close()
    with Ssimu2(0) as s:
        assert s.compute_ssimu2(ref, dst) == want


def test_per_kernel_timestamps_add_up_to_the_stream_time_and_the_graph_form_keeps_the_bits(hip_lib):
    """Round 6's two instrumented-build hooks, on the device.  ssimu2_time_kernels: every launch of the library's own enqueue
    path with a start / stop event pair -- the launches are the expected ones, in order, every duration positive, and their
    sum is the stream time of a score (no more than 3 % above the plain-launch stream time, not below 85 % of it: the rest
    is what the launches wait between them).  ssimu2_instr_use_graph: the launches of a score as ONE graph launch -- one
    graph built per context and shape, one graph launch per score, the exact bits of the plain form (the measured outcome,
    slower everywhere, is profiles/r06_graph_ab.log: the product library does not have this path)."""
    import torch
    from oavif_amd import Ssimu2, _lib
    w, h = 1920, 1080
    frames = []
    for k in range(6):
        r = synth.make_ref(w, h, 40 + k)
        frames.append((torch.from_numpy(r).cuda(), torch.from_numpy(synth.distort(r, "blockq", 1 + k % 3)).cuda()))
    refs, dsts = [a.data_ptr() for a, _ in frames], [b.data_ptr() for _, b in frames]
    for blur, cached, names in ((_lib.BLUR_FIR, False, ("pyramid", "march", "finalize")),
                                (_lib.BLUR_FIR, True, ("pyramid", "march_refblur", "finalize")),
                                (_lib.BLUR_RECURSIVE, True, ("convert", "h", "v", "finalize")),
                                (_lib.BLUR_RECURSIVE, False, ("ref_convert", "ref_h", "ref_v_emit", "convert", "h", "v", "finalize"))):
        with Ssimu2(0, instrumented=True, blur=blur) as c:
            if cached:
                st, wall_t, wall_p = c.time_kernels(w, h, dsts, 60, d_ref=refs[0], recursive=blur != _lib.BLUR_FIR)
            else:
                st, wall_t, wall_p = c.time_kernels(w, h, dsts, 60, d_refs=refs, recursive=blur != _lib.BLUR_FIR)
            assert tuple(st) == names, st
            assert all(v > 0.001 for v in st.values())
            total = sum(st.values())
            print(f"blur {blur} cached {cached}: kernels {total * 1e3:.1f} us of {wall_p * 1e3:.1f} us of stream time per score")
            assert 0.85 * wall_p <= total <= 1.03 * wall_p, (st, wall_p)
            assert wall_t >= 0.97 * wall_p                       # the timestamps cost something, never nothing

            def scores():
                out = []
                if cached:
                    c.set_reference_device(refs[0], w, h)
                for i in range(6):
                    if cached:
                        c.enqueue_against_reference_device(dsts[i])
                    else:
                        c.enqueue_device(refs[i], dsts[i], w, h)
                    out.append(c.wait())
                return out
            plain = scores()
            b0, n0 = c.use_graph(True)
            graph = scores()
            b1, n1 = c.use_graph(False)
            assert graph == plain                                 # same kernels, same arguments, same order: same bits
            assert b1 - b0 == 1 and n1 - n0 == 6                  # one graph for the shape, one launch per score
            assert scores() == plain and c.use_graph(None) == (b1, n1)
