"""Parity of the HIP scorer (through the C ABI) with the CPU oracle.  -m gpu only.

Tolerances (written here, as the task requires):
  * vs the oracle's primary (FIR) mode: |dscore| <= 2e-4 and the 108 plane averages to
    rtol 2e-5 -- the blur is bit-identical by construction, what remains is cbrt / division
    / reduction-order rounding.  North_star's bar is +-0.01.
  * vs the oracle's published-recursion (IIR, fp32) mode: not asserted to +-0.01; the gap is
    that recursion's own rounding noise (tests/test_oracle.py, DESIGN.md "Oracle").
The reference has no vectors for this path and its scorer source is absent: every
"expected" here is the repo's own oracle -- fssimu2 parity is UNPINNED.
"""
import os

import numpy as np
import pytest

from oavif_amd import synth

pytestmark = pytest.mark.gpu

TOL_SCORE = 1e-4
RTOL_AVG = 2e-5


def _check_pair(scorer, oracle, ref, dist, tol=TOL_SCORE):
    got = scorer.compute_ssimu2(ref, dist)
    avg_g, ns_g = scorer.last_averages()
    exp, avg_o, ns_o = oracle.compute_ssimu2(ref, dist, oracle.BLUR_FIR, return_averages=True)
    assert ns_g == ns_o
    assert np.allclose(avg_g, avg_o, rtol=RTOL_AVG, atol=1e-9), \
        np.abs(avg_g - avg_o).max()
    assert abs(got - exp) <= tol, (got, exp)
    return got, exp


def test_golden_fixtures(scorer, oracle, golden):
    arrays, meta = golden
    ref = arrays["ref"]
    for p in meta["pairs"]:
        got = scorer.compute_ssimu2(ref, arrays[p["name"]])
        avg, ns = scorer.last_averages()
        assert ns == 6
        assert abs(got - p["score_fir"]) <= TOL_SCORE, (p["name"], got, p["score_fir"])
        assert np.allclose(avg.reshape(-1), p["averages_fir"], rtol=RTOL_AVG, atol=1e-9), p["name"]
    o = meta["odd"]
    got = scorer.compute_ssimu2(arrays["odd_ref"], arrays["odd_dist"])
    assert abs(got - o["score_fir"]) <= TOL_SCORE


def test_real_photographs_through_the_published_quality_ladder(hip_lib, scorer, oracle):
    """Photographic content on the device: scikit-learn's two sample photographs through the JPEG ladder
    SSIMULACRA2 was published with (tests/photo_ladder.py; tests/test_oracle.py holds the checker to the published
    scores).  The HIP path in FIR mode equals the checker's FIR score and in the search path's default mode the
    checker's recursion, to 1e-4, on every rung -- and so sits within 7 points of the published table itself."""
    import oavif_amd
    from oavif_amd import _lib
    from tests.photo_ladder import LADDER, jpeg_round_trip, photographs
    photos = photographs()
    if not photos:
        pytest.skip("scikit-learn's sample photographs are not installed")
    with oavif_amd.Ssimu2(0, blur=_lib.BLUR_RECURSIVE) as rec:
        for q, sub, published in LADDER:
            got = []
            for _name, ref in photos:
                dist = jpeg_round_trip(ref, q, sub)
                _check_pair(scorer, oracle, ref, dist)
                r = rec.compute_ssimu2(ref, dist)
                assert abs(r - oracle.compute_ssimu2(ref, dist, oracle.BLUR_IIR)) <= TOL_SCORE
                rec.set_reference(ref)
                assert rec.score_against_reference(dist) == r
                got.append(r)
            assert abs(float(np.mean(got)) - published) <= 7.0, (q, sub, published, got)


def test_identical_is_exactly_100(scorer):
    ref = synth.make_ref(300, 200, 11)
    assert scorer.compute_ssimu2(ref, ref) == 100.0


@pytest.mark.parametrize("w,h", [(8, 8), (9, 15), (16, 16), (31, 33), (32, 32), (33, 31),
                                 (64, 40), (65, 41), (100, 7), (7, 100), (1, 1), (127, 129),
                                 (255, 3), (513, 259)])
def test_ragged_and_tiny_sizes(scorer, oracle, w, h):
    ref = synth.make_ref(w, h, w * 1000 + h)
    dist = synth.distort(ref, "noise", 2, seed=w)
    _check_pair(scorer, oracle, ref, dist)


@pytest.mark.parametrize("kind,strength", [("blockq", 0), ("blockq", 4), ("noise", 0), ("noise", 4),
                                           ("blur", 2), ("band", 1), ("band", 4)])
def test_distortion_ladder_512(scorer, oracle, kind, strength):
    ref = synth.make_ref(512, 512, 21)
    dist = synth.distort(ref, kind, strength, seed=5)
    _check_pair(scorer, oracle, ref, dist)


def test_extreme_frames(scorer, oracle):
    h, w = 70, 90
    black = np.zeros((h, w, 3), np.uint8)
    white = np.full((h, w, 3), 255, np.uint8)
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    for a, b in [(black, white), (white, black), (noise, black), (black, noise),
                 (noise, noise[::-1].copy())]:
        _check_pair(scorer, oracle, a, b, tol=5e-4)   # scores far below 0 here


def test_1080p_pair(scorer, oracle):
    ref = synth.make_ref(1920, 1080, 31)
    dist = synth.distort(ref, "blockq", 1)
    _check_pair(scorer, oracle, ref, dist)


def test_set_reference_path_is_bit_identical(scorer):
    ref = synth.make_ref(640, 360, 41)
    dists = [synth.distort(ref, "noise", s, seed=s) for s in range(4)]
    direct = [scorer.compute_ssimu2(ref, d) for d in dists]
    scorer.set_reference(ref)
    cached = [scorer.score_against_reference(d) for d in dists]
    assert direct == cached
    # and repeatable
    assert cached == [scorer.score_against_reference(d) for d in dists]


def test_reference_cache_device_path(scorer):
    import torch
    ref = synth.make_ref(777, 333, 43)
    dists = [synth.distort(ref, "blockq", s) for s in range(3)]
    direct = [scorer.compute_ssimu2(ref, d) for d in dists]
    t_ref = torch.from_numpy(ref).cuda().contiguous()
    t_d = [torch.from_numpy(d).cuda().contiguous() for d in dists]
    torch.cuda.synchronize()
    scorer.set_reference_device(t_ref.data_ptr(), 777, 333)
    got = []
    for t in t_d:
        scorer.enqueue_against_reference_device(t.data_ptr())
        got.append(scorer.wait())
    assert got == direct
    # a plain pair score in between invalidates the cached reference
    scorer.compute_ssimu2(ref, dists[0])
    from oavif_amd import Ssimu2Error, _lib
    with pytest.raises(Ssimu2Error) as ei:
        scorer.enqueue_against_reference_device(t_d[0].data_ptr())
    assert ei.value.code == _lib.ERR_NO_REFERENCE


def test_device_resident_entry_points(scorer):
    import torch
    ref = synth.make_ref(400, 300, 51)
    dist = synth.distort(ref, "blur", 1)
    host = scorer.compute_ssimu2(ref, dist)
    t_ref = torch.from_numpy(ref).cuda().contiguous()
    t_dist = torch.from_numpy(dist).cuda().contiguous()
    torch.cuda.synchronize()
    assert scorer.score_device(t_ref.data_ptr(), t_dist.data_ptr(), 400, 300) == host
    scorer.enqueue_device(t_ref.data_ptr(), t_dist.data_ptr(), 400, 300)
    assert scorer.wait() == host


def test_instrumented_build_scores_the_same_bits(scorer, iscorer):
    """liboavif_hip_instr.so is the product sources + hooks: same scores, and its timing hooks work."""
    import torch
    ref = synth.make_ref(400, 300, 51)
    dist = synth.distort(ref, "blur", 1)
    host = scorer.compute_ssimu2(ref, dist)
    assert iscorer.compute_ssimu2(ref, dist) == host
    t_ref = torch.from_numpy(ref).cuda().contiguous()
    t_dist = torch.from_numpy(dist).cuda().contiguous()
    torch.cuda.synchronize()
    ms, s = iscorer.time_device(t_ref.data_ptr(), t_dist.data_ptr(), 400, 300, 3)
    assert s == host and ms > 0
    from oavif_amd import _lib
    assert iscorer.time_stage(t_ref.data_ptr(), t_dist.data_ptr(), 400, 300, _lib.STAGE_MARCH, 3) > 0
    ptrs_r, ptrs_d = [t_ref.data_ptr()] * 2, [t_dist.data_ptr(), t_ref.data_ptr()]
    assert iscorer.time_march_rotating(ptrs_r, ptrs_d, 400, 300, 4) > 0
    assert iscorer.compute_ssimu2(ref, dist) == host          # the hooks leave the context usable
    with pytest.raises(RuntimeError):
        scorer.time_device(t_ref.data_ptr(), t_dist.data_ptr(), 400, 300, 3)   # not in the product


def test_two_contexts_are_independent(hip_lib, scorer):
    from oavif_amd import Ssimu2
    ref = synth.make_ref(320, 200, 61)
    d1 = synth.distort(ref, "noise", 1)
    d2 = synth.distort(ref, "blockq", 3)
    with Ssimu2(0) as other:
        a = scorer.compute_ssimu2(ref, d1)
        b = other.compute_ssimu2(ref, d2)
        assert a == other.compute_ssimu2(ref, d1)
        assert b == scorer.compute_ssimu2(ref, d2)


def test_error_codes(scorer):
    from oavif_amd import Ssimu2, Ssimu2Error, _lib
    ref = synth.make_ref(32, 32, 1)
    with pytest.raises(Ssimu2Error) as ei:
        scorer.compute_ssimu2(ref, ref, channels=4)     # the reference always passes 3
    assert ei.value.code == _lib.ERR_UNSUPPORTED
    with Ssimu2(0) as fresh:
        with pytest.raises(Ssimu2Error) as ei:
            fresh.score_against_reference(ref)
        assert ei.value.code == _lib.ERR_NO_REFERENCE
    with pytest.raises(Ssimu2Error) as ei:
        Ssimu2(99)
    assert ei.value.code == _lib.ERR_NO_DEVICE
    # dimensions are validated before any byte is touched: zero sizes and frames over the
    # 2^31/3-pixel limit are refused (the buffer behind the pointer is never read)
    import ctypes
    L = _lib.lib()
    out = ctypes.c_double()
    p8 = ref.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    for w, h in ((0, 32), (32, 0), (26753, 26757), (65536, 65536), (2**32 - 1, 2**32 - 1)):
        assert L.ssimu2_score_rgb8(scorer._ctx, p8, p8, w, h, 3, ctypes.byref(out)) == _lib.ERR_INVALID_ARG
        assert L.ssimu2_set_reference(scorer._ctx, p8, w, h) == _lib.ERR_INVALID_ARG
    assert scorer.compute_ssimu2(ref, ref) == 100.0    # the context is still usable


# ---- independent anchors -----------------------------------------------------
# The oracle's FIR mode mirrors the kernel's operation order (it is the bit-level contract), so
# agreement with it shows the two follow the same contract, not that the contract is right.  The
# anchors below share no rounding sequence with the kernel's blur:
#   * OR_BLUR_EXACT: the same operator with the blur accumulated in fp64.  Tolerance: 1e-2 score
#     points -- north_star's own +-0.01 -- on every fixture; measured <= 6e-3 on the 192x144
#     fixtures and ~1e-4 on 1080p / 4K frames (rounding noise averages out over more pixels).
#   * OR_BLUR_IIR: the published fp32 recursion.  Its gap to the FIR form is that recursion's
#     own rounding noise; it is RECORDED per fixture in
#     tests/golden/pairs_v1_anchors.json and the HIP path must reproduce the recorded gap.
TOL_EXACT = 1e-2


def _golden_cases(golden):
    arrays, meta = golden
    cases = [(p["name"], arrays["ref"], arrays[p["name"]]) for p in meta["pairs"]]
    cases.append(("odd", arrays["odd_ref"], arrays["odd_dist"]))
    return cases


def test_hip_matches_the_fp64_blur_anchor_on_every_fixture(scorer, golden, anchors):
    worst = 0.0
    for name, ref, dist in _golden_cases(golden):
        got = scorer.compute_ssimu2(ref, dist)
        avg, _ = scorer.last_averages()
        a = anchors[name]
        worst = max(worst, abs(got - a["score_exact"]))
        assert abs(got - a["score_exact"]) <= TOL_EXACT, (name, got, a["score_exact"])
        # the 108 averages against the fp64-blur ones, absolute: fp32 blur rounding passes
        # through the maps' max(0, .) and moves an average by up to 5.4e-6 on these 192x144
        # frames (measured, CPU), whatever its size (1e-7 .. 1e-1); the weighted sum of those
        # shifts is the score difference bounded above
        assert np.allclose(avg.reshape(-1), a["averages_exact"], rtol=0, atol=1e-5), name
    assert worst <= 6.5e-3   # what the CPU data says (tests/golden/extend_golden.py output)


@pytest.mark.parametrize("w,h,seed,kind,strength", [(1920, 1080, 31, "blockq", 1), (3840, 2160, 0, "blockq", 2)])
def test_hip_matches_the_fp64_blur_anchor_on_large_frames(scorer, oracle, w, h, seed, kind, strength):
    ref = synth.make_ref(w, h, seed)
    dist = synth.distort(ref, kind, strength)
    got = scorer.compute_ssimu2(ref, dist)
    oracle.set_num_threads(16)
    exact = oracle.compute_ssimu2(ref, dist, oracle.BLUR_EXACT, omp=True)
    assert abs(got - exact) <= 1e-3, (got, exact)     # well inside north_star's +-0.01


def test_published_recursion_gap_is_the_recorded_one(scorer, golden, anchors):
    """HIP (FIR) vs the published fp32 recursion: the difference equals, fixture by fixture, the
    gap recorded when the anchors were generated (fir - iir, up to 0.11 points: the recursion's
    own rounding noise, not an error of either side -- its two evaluation orders differ from
    each other by as much, see score_iir_fma in the anchors file)."""
    for name, ref, dist in _golden_cases(golden):
        a = anchors[name]
        got = scorer.compute_ssimu2(ref, dist)
        assert abs((got - a["score_iir"]) - a["gap_fir_minus_iir"]) <= 2e-4, name
        # and the recorded gaps themselves stay inside the documented envelope
        assert abs(a["gap_fir_minus_iir"]) < 0.15 and abs(a["score_iir"] - a["score_iir_fma"]) < 0.15


# ---- the search: identical probe sequence and final quantizer, CPU scorer vs HIP scorer ------

@pytest.mark.skipif(not synth.have_avif(), reason="Pillow AVIF codec not available")
@pytest.mark.parametrize("seed,tgt", [(0, 80.0), (1, 60.0), (2, 90.0), (3, 75.0)])
def test_search_quantizer_identical_to_cpu(scorer, oracle, seed, tgt):
    from oavif_amd import tq
    from oracle import tq_oracle
    ref = synth.make_ref(384, 256, 100 + seed)
    cache = {}

    def codec(q):
        if q not in cache:
            cache[q] = synth.avif_roundtrip(ref, q, speed=9)
        return cache[q]

    gpu = tq.search_hip(scorer, ref, codec, score_tgt=tgt, tolerance=2.0, max_pass=6)
    cpu = tq_oracle.find_target_quality(
        lambda q: oracle.compute_ssimu2(ref, codec(q)[0], oracle.BLUR_FIR),
        score_tgt=tgt, tolerance=2.0, max_pass=6)
    assert [q for q, _ in gpu.history] == [q for q, _ in cpu.history]
    assert (gpu.q, gpu.num_pass, gpu.buf_q) == (cpu.q, cpu.num_pass, cpu.buf_q)
    assert max(abs(a[1] - b[1]) for a, b in zip(gpu.history, cpu.history)) <= TOL_SCORE
    assert gpu.last_avif_size == cache[gpu.buf_q][1]


def test_4k_properties(scorer):
    """BASELINE size (3840x2160): size-independent properties instead of a slow oracle run:
    identical -> 100, determinism, monotone ladder, set_reference == direct."""
    ref = synth.make_ref(3840, 2160, 71)
    assert scorer.compute_ssimu2(ref, ref) == 100.0
    scores = [scorer.compute_ssimu2(ref, synth.distort(ref, "blockq", s)) for s in (0, 2, 4)]
    assert scores[0] > scores[1] > scores[2]
    d = synth.distort(ref, "blockq", 2)
    assert scorer.compute_ssimu2(ref, d) == scores[1]
    scorer.set_reference(ref)
    assert scorer.score_against_reference(d) == scores[1]


def test_4k_against_oracle(scorer, oracle):
    """One full-size oracle comparison (OpenMP build, ~10 s of CPU)."""
    ref = synth.make_ref(3840, 2160, 72)
    d = synth.distort(ref, "noise", 2, seed=9)
    got = scorer.compute_ssimu2(ref, d)
    exp = oracle.compute_ssimu2(ref, d, oracle.BLUR_FIR, omp=True)
    assert abs(got - exp) <= TOL_SCORE


@pytest.mark.skipif(not synth.have_avif(), reason="Pillow AVIF codec not available")
def test_batch_driver_single_rank(scorer, tmp_path):
    """measure.py counterpart, one rank: real CPU codec, GPU scorer, CSV out."""
    from PIL import Image
    from oavif_amd import batch
    for i in range(3):
        Image.fromarray(synth.make_ref(160 + 16 * i, 128, 200 + i)).save(tmp_path / f"img{i}.png")
    files = batch.list_images(tmp_path)
    out = tmp_path / "out"
    out.mkdir()

    def enc(_i, path):
        return batch.encode_image(scorer, path, out / f"{path.stem}.avif", 80.0, 2.0, 6, 9)
    res = batch.run_batch(files, enc, 0, 1)
    assert [r.status for r in res] == ["ok"] * 3
    for r in res:
        assert 1 <= r.passes <= 6 and 0 <= r.q <= 100
        assert (out / f"{r.image[:-4]}.avif").stat().st_size == r.final_bytes
        # the file on disk decodes and really has (about) the reported score
        dec = synth.avif_decode((out / f"{r.image[:-4]}.avif").read_bytes())
        ref = np.asarray(Image.open(tmp_path / r.image).convert("RGB"))
        assert abs(scorer.compute_ssimu2(ref, dec) - r.score) < 1e-9
    batch.write_csv(tmp_path / "o.csv", res)
    assert (tmp_path / "o.csv").read_text().splitlines()[0].startswith("Image,Original Bytes")


def test_probe_fanout_over_contexts_matches_sequential(hip_lib, scorer):
    """BASELINE configs[2]: probes fanned across HIP streams on one GPU (independent contexts)."""
    import torch
    import oavif_amd
    ref = synth.make_ref(960, 540, 81)
    dists = [synth.distort(ref, k, s, seed=s) for k, s in
             [("blockq", 0), ("blockq", 2), ("noise", 1), ("noise", 3), ("blur", 1), ("band", 2), ("blur", 3)]]
    seq = [scorer.compute_ssimu2(ref, d) for d in dists]
    t_ref = torch.from_numpy(ref).cuda().contiguous()
    t_d = [torch.from_numpy(d).cuda().contiguous() for d in dists]
    torch.cuda.synchronize()
    ctxs = [oavif_amd.Ssimu2(0) for _ in range(3)]
    try:
        fan = oavif_amd.score_many(ctxs, t_ref.data_ptr(), [t.data_ptr() for t in t_d], 960, 540)
    finally:
        for c in ctxs:
            c.close()
    assert fan == seq
    # replaying the search over cached probe scores gives the sequential search's answer
    from oavif_amd import tq
    table = {q: s for q, s in zip([10, 30, 50, 65, 80, 90, 100], sorted(seq))}
    r1 = tq.find_target_quality(lambda q: table[min(table, key=lambda k: abs(k - q))], score_tgt=80.0)
    r2 = tq.find_target_quality(lambda q: table[min(table, key=lambda k: abs(k - q))], score_tgt=80.0)
    assert (r1.q, r1.history) == (r2.q, r2.history)


def _large_anchor_cases():
    import json as _json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "large_anchors.json")
    return _json.load(open(path))["cases"]


@pytest.mark.parametrize("case", _large_anchor_cases(), ids=lambda c: f"{c['w']}x{c['h']}-{c['kind']}{c['strength']}")
def test_large_frame_anchors_all_blur_modes(hip_lib, case):
    """Full-size parity for BASELINE configs[1..3] (1080p, 4K, 8K) against COMMITTED checker scores
    (tests/golden/large_anchors.json, written by tests/golden/make_large_anchors.py; the frames are
    regenerated from their seeds and checked against the recorded checksum).  Default mode: 1e-4
    against the FIR checker, north_star's own +-0.01 against the fp64-blur evaluation (no rounding
    sequence shared with the kernels).  Recursive modes: 1e-4 against the published fp32 recursion
    and its fused order.  fssimu2 parity stays UNPINNED (these are this repo's checker's numbers)."""
    from oavif_amd import Ssimu2, _lib
    w, h = case["w"], case["h"]
    ref = synth.make_ref(w, h, case["seed"])
    crc = int(np.bitwise_xor.reduce(ref.reshape(-1).astype(np.uint32) * np.arange(1, ref.size + 1, dtype=np.uint32) & 0xFFFFFFFF))
    assert crc == case["ref_crc"], "synth.make_ref no longer regenerates the fixture's frame"
    dist = synth.distort(ref, case["kind"], case["strength"], seed=case["seed"])
    with Ssimu2(0) as s:
        fir = s.compute_ssimu2(ref, dist)
        assert abs(fir - case["score_fir"]) <= TOL_SCORE, (fir, case["score_fir"])
        assert abs(fir - case["score_exact"]) <= 0.01, (fir, case["score_exact"])
        s.set_reference(ref)
        assert s.score_against_reference(dist) == fir
        s.set_blur(_lib.BLUR_RECURSIVE)
        rec = s.compute_ssimu2(ref, dist)
        assert abs(rec - case["score_iir"]) <= TOL_SCORE, (rec, case["score_iir"])
        s.set_reference(ref)
        assert s.score_against_reference(dist) == rec          # the cached-reference pass: same bits
        s.set_blur(_lib.BLUR_RECURSIVE_FMA)
        assert abs(s.compute_ssimu2(ref, dist) - case["score_iir_fma"]) <= TOL_SCORE


def test_8k_against_oracle(scorer, oracle):
    """BASELINE configs[2] size (7680x4320) against a LIVE run of the checker (OpenMP build: ~1-4 s
    per 8K score in FIR mode), a pair that is not in the committed anchors."""
    w, h = 7680, 4320
    ref = synth.make_ref(w, h, 93)
    dist = synth.distort(ref, "noise", 2, seed=3)
    from oavif_amd import hostinfo
    got = scorer.compute_ssimu2(ref, dist)
    avg_g, ns_g = scorer.last_averages()
    oracle.set_num_threads(hostinfo.usable_cores())   # the cgroup's cores, not the host's 256 threads
    exp, avg_o, ns_o = oracle.compute_ssimu2(ref, dist, oracle.BLUR_FIR, omp=True, return_averages=True)
    assert ns_g == ns_o == 6
    assert abs(got - exp) <= TOL_SCORE, (got, exp)
    assert np.allclose(avg_g, avg_o, rtol=RTOL_AVG, atol=1e-9)


def test_8k_properties(scorer):
    """BASELINE configs[2] size (7680x4320): size-independent properties."""
    ref = synth.make_ref(7680, 4320, 91)
    assert scorer.compute_ssimu2(ref, ref) == 100.0
    d1 = synth.distort(ref, "blockq", 1)
    d3 = synth.distort(ref, "blockq", 3)
    s1 = scorer.compute_ssimu2(ref, d1)
    s3 = scorer.compute_ssimu2(ref, d3)
    assert 100.0 > s1 > s3
    assert scorer.compute_ssimu2(ref, d1) == s1           # deterministic
    _, ns = scorer.last_averages()
    assert ns == 6


@pytest.mark.parametrize("w,h", [(119, 40), (120, 40), (121, 40), (239, 33), (240, 33), (241, 33),
                                 (128, 17), (136, 9), (360, 8), (361, 24), (1200, 16), (24, 400),
                                 (9, 1000)])
def test_strip_and_segment_boundaries(scorer, oracle, w, h):
    """Widths around multiples of the 120-column strip, heights around the 8-row minimum
    segment and the 16-row ring, very tall / very wide frames."""
    rng = np.random.default_rng(w * 7919 + h)
    ref = synth.make_ref(w, h, w + h)
    dist = np.clip(ref.astype(np.int16) + rng.integers(-9, 10, ref.shape), 0, 255).astype(np.uint8)
    _check_pair(scorer, oracle, ref, dist)


def _content(kind, w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "gradient":
        img = np.stack([xx * 255 // max(w - 1, 1), yy * 255 // max(h - 1, 1),
                        (xx + yy) * 255 // max(w + h - 2, 1)], -1)
    elif kind == "primaries":      # saturated patches: exercises the opsin clamp / B-Y remap
        cols = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255],
                         [255, 0, 255], [0, 0, 0], [255, 255, 255]])
        img = cols[((xx // 16) + (yy // 16)) % 8]
    elif kind == "checker":        # 1-px checkerboard: maximal high-frequency energy
        img = np.repeat((((xx + yy) & 1) * 255)[..., None], 3, -1)
    elif kind == "text":           # thin dark strokes on light ground
        img = np.full((h, w, 3), 235)
        for _ in range(60):
            x, y = int(rng.integers(0, w)), int(rng.integers(0, h))
            img[y:y + 1 + int(rng.integers(0, 2)), x:x + int(rng.integers(3, 30))] = 20
            img[y:y + int(rng.integers(3, 20)), x:x + 1] = 20
    else:                          # white noise
        img = rng.integers(0, 256, (h, w, 3))
    return np.ascontiguousarray(img.astype(np.uint8))


@pytest.mark.parametrize("kind", ["gradient", "primaries", "checker", "text", "noise"])
def test_content_types(scorer, oracle, kind):
    ref = _content(kind, 250, 190, 5)
    for dk, ds in [("blur", 0), ("band", 2), ("noise", 2)]:
        _check_pair(scorer, oracle, ref, synth.distort(ref, dk, ds, seed=3), tol=5e-4)


def test_one_pixel_difference_is_detected(scorer, oracle):
    ref = synth.make_ref(200, 150, 77)
    d = ref.copy()
    d[75, 100] = 255 - d[75, 100]
    got, exp = _check_pair(scorer, oracle, ref, d)
    assert got < 100.0


@pytest.mark.skipif(not synth.have_avif(), reason="Pillow AVIF codec not available")
def test_cli_search_end_to_end(scorer, tmp_path, capsys):
    """`oavif in.png out.avif` surface: stderr lines as main.zig prints them, measure.py's
    "N passes" contract, file written at the chosen quantizer."""
    import re
    from PIL import Image
    from oavif_amd import cli
    src = tmp_path / "in.png"
    ref = synth.make_ref(320, 240, 303)
    Image.fromarray(ref).save(src)
    out = tmp_path / "out.avif"
    assert cli.main(["--score-tgt", "75", "--tolerance", "1.5", str(src), str(out)], scorer=scorer) == 0
    err = capsys.readouterr().err.splitlines()
    assert err[1].startswith("Read 320x240, RGB, 8-bit, ")
    assert err[2] == "Searching [tgt 75±1.5, speed 9, 8-bit]"
    m = re.fullmatch(r"Found q(\d+) \(score (-?\d+\.\d{2}), (\d+) passes\)", err[3])
    assert m, err[3]
    assert re.search(r"(\d+)\s+passes?", err[3]).group(1) == m.group(3)      # measure.py:27
    assert re.fullmatch(r"Compressed to \d+ bytes \(\d+\.\d{3} bpp\)", err[4])
    dec = synth.avif_decode(out.read_bytes())
    assert abs(scorer.compute_ssimu2(ref, dec) - float(m.group(2))) < 0.006
    # PAM input and a 4-channel source go through the same path (alpha dropped for the scorer)
    from oavif_amd import pam
    rgba = np.dstack([ref, np.full(ref.shape[:2], 200, np.uint8)])
    p = tmp_path / "in.pam"
    p.write_bytes(pam.write_pam(rgba))
    assert cli.main([str(p), str(tmp_path / "o2.avif")], scorer=scorer) == 0
    assert "Read 320x240, RGBA, 8-bit" in capsys.readouterr().err


@pytest.mark.skipif(not synth.have_avif(), reason="Pillow AVIF codec not available")
def test_batch_cli_with_worker_threads(hip_lib, tmp_path):
    """`python -m oavif_amd.batch` end to end on one GPU: 4 worker threads (one scorer context
    each) must give the same per-image results as 1."""
    import csv
    from PIL import Image
    from oavif_amd import batch
    d = tmp_path / "imgs"
    d.mkdir()
    for i in range(6):
        Image.fromarray(synth.make_ref(200 + 8 * i, 150, 400 + i)).save(d / f"im{i:02d}.png")
    rows = {}
    for workers in (1, 4):
        out = tmp_path / f"r{workers}.csv"
        rc = batch.main([str(d), str(out), "--workers", str(workers), "--out-dir", str(tmp_path / f"o{workers}")])
        assert rc == 0
        rows[workers] = [(r[0], r[2], r[6], r[7]) for r in list(csv.reader(open(out)))[1:]]
    assert rows[1] == rows[4]
    assert len(rows[1]) == 6 and all(r[3] == "ok" for r in rows[1])


@pytest.mark.parametrize("w,h", [(333, 217), (512, 512), (121, 9)])
def test_intermediate_planes_are_bit_identical(iscorer, oracle, w, h):
    scorer = iscorer  # plane download is a hook of the instrumented build
    """Stage-by-stage parity (bit-exact, as for integer work): the linear-light pyramid of both
    frames and the cached positive-XYB planes of the reference equal the oracle's planes bit for
    bit -- the arithmetic contract (explicit fmaf order, reproducible cube root) holds per pixel,
    so everything upstream of the blur is identical and the blur itself is the same fmaf chain."""
    ref = synth.make_ref(w, h, 11 * w + h)
    dist = synth.distort(ref, "noise", 2, seed=1)
    lut = oracle.srgb_lut()
    lin = {0: [np.ascontiguousarray(lut[f].transpose(2, 0, 1)) for f in (ref, dist)]}
    scorer.compute_ssimu2(ref, dist)
    _, ns = scorer.last_averages()
    for s in range(1, ns):
        lin[s] = [oracle.downsample2(a) for a in lin[s - 1]]
        got_r = scorer.debug_download(0, s, w, h)
        got_d = scorer.debug_download(1, s, w, h)
        assert np.array_equal(got_r.view(np.uint32), lin[s][0].view(np.uint32)), s
        assert np.array_equal(got_d.view(np.uint32), lin[s][1].view(np.uint32)), s
    scorer.set_reference(ref)
    for s in range(ns):
        got = scorer.debug_download(2, s, w, h)
        exp = oracle.linear_to_xyb(lin[s][0])
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), s


@pytest.mark.parametrize("w,h", [(333, 217), (121, 9), (640, 360), (1000, 700), (1921, 1083), (3840, 2160)])
def test_blur_waves_output_is_bit_identical(iscorer, oracle, w, h):
    """The blur itself, bit for bit: blur(ref * ref) of every XYB channel at every scale, as the
    marching body's blur waves write it in emit mode (k_ref_blur: the same horizontal pass with
    the products fused into the pair sums, the same register-window vertical pass, zero padding,
    strips and segments as in k_march), equals the oracle's FIR planes exactly -- across strip
    (120 columns) and segment boundaries and at the frame border.  And it is within 2 ulp-ish of
    the fp64 evaluation of the same operator (the independent anchor, per pixel)."""
    ref = synth.make_ref(w, h, 7 * w + h)
    iscorer.set_reference(ref)
    lin = np.ascontiguousarray(oracle.srgb_lut()[ref].transpose(2, 0, 1))
    s = 0
    while True:
        xyb = oracle.linear_to_xyb(lin)
        got = iscorer.debug_download(3, s, w, h)
        for c in range(3):
            exp = oracle.blur_product(xyb[c], xyb[c], oracle.BLUR_FIR)
            assert np.array_equal(got[c].view(np.uint32), exp.view(np.uint32)), (s, c)
            exact = oracle.blur_product(xyb[c], xyb[c], oracle.BLUR_EXACT)
            assert np.max(np.abs(got[c] - exact) / np.maximum(np.abs(exact), 1e-6)) < 1e-6, (s, c)
        s += 1
        if s >= 6 or lin.shape[1] < 8 or lin.shape[2] < 8:  # the published loop: next scale iff this one is >= 8x8
            break
        lin = oracle.downsample2(lin)

def test_every_rgb8_colour_converts_bit_identically(iscorer, scorer, oracle):
    """All 2^24 8-bit colours, once each (4096 x 4096): the device's sRGB table, opsin mix,
    cube root and positive-XYB offsets give the oracle's bits for EVERY input the scorer can be
    handed at full resolution (the planes of k_ref_xyb, which shares its device functions with
    the marching kernel's converter waves), and the colour cube scored against a noisy copy
    goes through the marching converter itself."""
    v = np.arange(1 << 24, dtype=np.uint32)
    ref = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    del v
    iscorer.set_reference(ref)
    got = iscorer.debug_download(2, 0, 4096, 4096)
    lin = np.ascontiguousarray(oracle.srgb_lut()[ref].transpose(2, 0, 1))
    exp = oracle.linear_to_xyb(lin)
    del lin
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    del got, exp
    dist = synth.distort(ref, "noise", 1, seed=5)
    assert abs(scorer.compute_ssimu2(ref, dist) - oracle.compute_ssimu2(ref, dist, oracle.BLUR_FIR)) <= TOL_SCORE


def test_caller_owned_stream(hip_lib):
    """ssimu2_ctx_create(device, hipStream_t): all work goes on the caller's stream."""
    import torch
    import oavif_amd
    ref = synth.make_ref(300, 200, 9)
    dist = synth.distort(ref, "blockq", 1)
    with oavif_amd.Ssimu2(0) as own:
        expect = own.compute_ssimu2(ref, dist)
    st = torch.cuda.Stream()
    with oavif_amd.Ssimu2(0, stream=st.cuda_stream) as s:
        assert s.compute_ssimu2(ref, dist) == expect
        t_ref = torch.from_numpy(ref).cuda().contiguous()
        t_d = torch.from_numpy(dist).cuda().contiguous()
        torch.cuda.synchronize()
        s.enqueue_device(t_ref.data_ptr(), t_d.data_ptr(), 300, 200)
        st.synchronize()                      # the caller's own synchronisation covers the work
        assert s.wait() == expect


@pytest.mark.parametrize("seg,tail", [(8, 8), (13, 21), (47, 160), (160, 9), (1, 1)])
def test_any_segment_length_gives_the_same_score(hip_lib, scorer, seg, tail):
    """The per-workgroup row ranges (an experiment knob of the instrumented build) only regroup
    the fp64 partial sums; values outside 8..160 are refused."""
    import oavif_amd
    ref = synth.make_ref(517, 391, 19)
    dist = synth.distort(ref, "noise", 2, seed=4)
    expect = scorer.compute_ssimu2(ref, dist)
    with oavif_amd.Ssimu2(0, instrumented=True) as s:
        if seg < 8 or tail < 8:
            with pytest.raises(oavif_amd.Ssimu2Error):
                s.set_segment_rows(seg, tail)
            return
        s.set_segment_rows(seg, tail)
        got = s.compute_ssimu2(ref, dist)
        assert abs(got - expect) < 1e-7
        s.set_segment_rows(0, 0)
        assert s.compute_ssimu2(ref, dist) == expect


def _decoded_like(dist, channels, pad, seed):
    """`dist` laid out like libavif's avifRGBImage: `channels` bytes per pixel (alpha random),
    rows `pad` bytes longer than their pixels, padding filled with noise."""
    h, w, _ = dist.shape
    rng = np.random.default_rng(seed)
    pitch = w * channels + pad
    buf = rng.integers(0, 256, (h, pitch), dtype=np.uint8)
    view = np.lib.stride_tricks.as_strided(buf, (h, w, channels), (pitch, channels, 1))
    view[..., :3] = dist
    return buf, view


@pytest.mark.parametrize("w,h,channels,pad", [
    (640, 360, 4, 0), (640, 360, 4, 64), (641, 359, 4, 0), (642, 100, 4, 2), (643, 77, 4, 3),
    (640, 360, 3, 0), (640, 360, 3, 32), (333, 217, 3, 1), (8, 8, 4, 0), (9, 9, 4, 5),
    (3840, 2160, 4, 0), (1920, 1080, 4, 0)])
def test_decoded_frame_handoff_matches_cpu_copy(scorer, oracle, w, h, channels, pad):
    """ssimu2_score_against_reference_strided on libavif's RGB(A) rows == the reference's CPU
    copy loop (io.zig:654-663, restated by the oracle) followed by the plain score."""
    ref = synth.make_ref(w, h, 71)
    dist = synth.distort(ref, "blockq", 1)
    buf, view = _decoded_like(dist, channels, pad, seed=w + h)
    tight = oracle.copy_rgb_pixels(view)
    assert np.array_equal(tight, dist)
    scorer.set_reference(ref)
    expect = scorer.score_against_reference(tight)
    assert scorer.score_decoded_against_reference(view) == expect
    # flat-buffer form, as a C caller passes rgb.pixels / rgb.rowBytes
    assert scorer.score_decoded_against_reference(buf.reshape(-1), row_bytes=buf.shape[1],
                                                  channels=channels) == expect
    # repeated and interleaved with the tight path: the staging buffer is private
    assert scorer.score_against_reference(tight) == expect
    assert scorer.score_decoded_against_reference(view) == expect


def test_decoded_frame_handoff_errors(scorer):
    from oavif_amd import Ssimu2Error, _lib
    ref = synth.make_ref(64, 48, 3)
    rgba = np.zeros((48, 64, 4), np.uint8)
    with oavif_amd_scorer() as s:
        with pytest.raises(Ssimu2Error) as ei:       # no reference yet
            s.score_decoded_against_reference(rgba)
        assert ei.value.code == _lib.ERR_NO_REFERENCE
        s.set_reference(ref)
        with pytest.raises(Ssimu2Error) as ei:       # grey+alpha is not a decoder output
            s.score_decoded_against_reference(np.zeros(48 * 64 * 2, np.uint8), row_bytes=128, channels=2)
        assert ei.value.code == _lib.ERR_UNSUPPORTED
        with pytest.raises(Ssimu2Error) as ei:       # rows shorter than their pixels
            s.score_decoded_against_reference(rgba.reshape(-1), row_bytes=64 * 4 - 1, channels=4)
        assert ei.value.code == _lib.ERR_INVALID_ARG
        assert s.score_decoded_against_reference(np.concatenate([ref, rgba[..., 3:]], axis=2)) == 100.0


def oavif_amd_scorer():
    import oavif_amd
    return oavif_amd.Ssimu2(0)


FLIP_TOL = 2e-3   # points; rounding-order effect of mirroring, see test_flip_invariance


def _flip(a):
    return np.ascontiguousarray(a[::-1, ::-1])


@pytest.mark.parametrize("w,h", [(1920, 1088), (640, 352), (1000, 700)])
def test_flip_invariance(scorer, oracle, w, h):
    """Mirroring both frames leaves the metric unchanged in exact arithmetic (symmetric taps,
    pointwise maps) when the dimensions are multiples of 32, so that every scale is mirrored too.
    In fp32 it moves by rounding order only -- the 2x2 box sum and the fused pair products
    fma(a-*b-, a+*b+) of the arithmetic contract are not mirror-symmetric in rounding -- which is
    ~1e-4 of a point here (FLIP_TOL); the mirrored pair still matches the oracle to TOL_SCORE.
    Ragged sizes (1000x700) clamp the odd edge at the other end: oracle comparison only."""
    ref = synth.make_ref(w, h, 83)
    dist = synth.distort(ref, "blockq", 2)
    dist[-40:, -40:] = 255 - dist[-40:, -40:]          # something only the far corner holds
    a = scorer.compute_ssimu2(ref, dist)
    b = scorer.compute_ssimu2(_flip(ref), _flip(dist))
    if w % 32 == 0 and h % 32 == 0:
        assert abs(a - b) < FLIP_TOL
    assert abs(b - oracle.compute_ssimu2(_flip(ref), _flip(dist), oracle.BLUR_FIR)) <= TOL_SCORE


def test_maximum_size_far_corner_is_addressed_correctly(scorer):
    """26752 x 26752 (0.716 Gpx, 2.147 GB per frame: 475 KB under the ABI's limit of 2^31/3 px,
    the frame's bytes just fit a signed 32-bit offset and its fp32 planes do not): every index
    the kernels form must survive sizes where 32-bit element offsets overflow.  No oracle run
    at this size; the size-independent property is flip invariance (to FLIP_TOL) with a strong
    distortion that only the far corner tile holds -- a far-end row or plane offset that wrapped
    would score the mirrored pair differently by about the patch's whole effect -- plus
    identical -> 100 and the patch being seen at all."""
    n, t, reps = 26752, 2432, 11                        # 11 x 11 tiles, t % 32 == 0
    base = synth.make_ref(t, t, 97)
    ref = np.tile(base, (reps, reps, 1))
    dist = np.tile(synth.distort(base, "blockq", 1), (reps, reps, 1))
    assert ref.shape == (n, n, 3)
    without = scorer.compute_ssimu2(ref, dist)
    dist[-t:, -t:] = 255 - dist[-t:, -t:]
    s = scorer.compute_ssimu2(ref, dist)
    _, ns = scorer.last_averages()
    assert ns == 6
    assert s + 100 * FLIP_TOL < without < 100.0         # the far corner is read, and matters
    rf, df = _flip(ref), _flip(dist)
    del dist
    assert abs(scorer.compute_ssimu2(rf, df) - s) < FLIP_TOL
    scorer.set_reference(rf)
    assert scorer.score_against_reference(df) == scorer.compute_ssimu2(rf, df)
    del df
    assert scorer.compute_ssimu2(ref, ref) == 100.0


# ---- probes of one search fanned over contexts / HIP streams (BASELINE configs[2]) --

def _pseudo_codec(ref):
    """Deterministic stand-in for encode(q) -> decode: coarser block quantisation for lower q."""
    def codec(q):
        step = 1 + (100 - q) // 3
        dec = (ref.astype(np.int32) // step) * step + step // 2
        return np.clip(dec, 0, 255).astype(np.uint8), 1000 + 10 * q
    return codec


@pytest.mark.parametrize("w,h,tgt,fan", [(1920, 1080, 80.0, 4), (1920, 1080, 65.0, 6),
                                         (7680, 4320, 80.0, 4)])
def test_speculative_search_over_streams_equals_sequential(hip_lib, scorer, w, h, tgt, fan):
    """The probes of one search run concurrently, each on its own scorer context (HIP stream)
    and host thread; the result must be the sequential search's, pass for pass.  7680x4320 is
    BASELINE configs[2]'s frame size (its 10-bit switch changes the CPU encode only)."""
    import oavif_amd
    from oavif_amd import tq
    ref = synth.make_ref(w, h, 77)
    codec = _pseudo_codec(ref)
    seq = tq.search_hip(scorer, ref, codec, score_tgt=tgt)
    ctxs = [oavif_amd.Ssimu2(0) for _ in range(fan)]
    try:
        res, stats, sizes = tq.search_speculative_hip(ctxs, ref, codec, score_tgt=tgt)
    finally:
        for c in ctxs:
            c.close()
    assert (res.q, res.score, res.num_pass, res.buf_q) == (seq.q, seq.score, seq.num_pass, seq.buf_q)
    assert res.history == seq.history
    assert res.last_avif_size == seq.last_avif_size == 1000 + 10 * seq.buf_q
    assert stats.waves + stats.cache_hits == seq.num_pass and stats.waves <= seq.num_pass
    assert stats.probes_issued == len(sizes) <= stats.waves * fan


@pytest.mark.skipif(not synth.have_avif(), reason="Pillow AVIF codec not available")
@pytest.mark.parametrize("seed,tgt", [(0, 80.0), (1, 70.0)])
def test_speculative_search_with_real_avif_codec(hip_lib, scorer, seed, tgt):
    import oavif_amd
    from oavif_amd import tq
    ref = synth.make_ref(640, 480, 300 + seed)
    codec = lambda q: synth.avif_roundtrip(ref, q, speed=9)
    seq = tq.search_hip(scorer, ref, codec, score_tgt=tgt)
    ctxs = [oavif_amd.Ssimu2(0) for _ in range(8)]
    try:
        res, stats, _ = tq.search_speculative_hip(ctxs, ref, codec, score_tgt=tgt)
    finally:
        for c in ctxs:
            c.close()
    assert (res.q, res.score, res.num_pass, res.history) == (seq.q, seq.score, seq.num_pass, seq.history)
    assert stats.waves <= seq.num_pass


@pytest.mark.skipif(not synth.have_avif(), reason="Pillow AVIF codec not available")
def test_cli_probe_fanout_env_gives_the_same_file_and_lines(hip_lib, tmp_path, capsys, monkeypatch):
    """OAVIF_PROBE_FANOUT=N fans the probes of the CLI's search over N contexts; stderr lines
    and the written AVIF are those of the plain run."""
    from PIL import Image
    from oavif_amd import cli
    src = tmp_path / "in.png"
    Image.fromarray(synth.make_ref(400, 304, 909)).save(src)
    outs = []
    for fan in ("1", "6"):
        monkeypatch.setenv("OAVIF_PROBE_FANOUT", fan)
        out = tmp_path / f"out{fan}.avif"
        assert cli.main(["--score-tgt", "78", str(src), str(out)]) == 0
        outs.append((capsys.readouterr().err.splitlines()[1:], out.read_bytes()))
    assert outs[0][0] == outs[1][0]
    assert outs[0][1] == outs[1][1]


def test_read_stream_probe_reports_a_plausible_bandwidth(iscorer):
    scorer = iscorer
    """ssimu2_measure_read_stream: between 1 and 8 TB/s on an MI355X for a 1 GiB buffer, and the
    context scores normally afterwards."""
    gbs = scorer.measure_read_stream(1 << 30, 5)
    assert 1000.0 < gbs < 8000.0
    ref = synth.make_ref(64, 64, 5)
    assert scorer.compute_ssimu2(ref, ref) == 100.0
    from oavif_amd import Ssimu2Error, _lib
    with pytest.raises(Ssimu2Error) as ei:
        scorer.measure_read_stream(1024, 5)
    assert ei.value.code == _lib.ERR_INVALID_ARG


def test_prefetch_then_create_scores_normally(hip_lib):
    import oavif_amd
    assert hip_lib.ssimu2_prefetch(0) == 0
    assert hip_lib.ssimu2_prefetch(0) == 0
    ref = synth.make_ref(96, 64, 2)
    with oavif_amd.Ssimu2(0) as s:                     # waits for the prefetch, then is quick
        assert s.compute_ssimu2(ref, ref) == 100.0
