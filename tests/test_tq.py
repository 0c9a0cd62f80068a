"""Search control logic: oavif_amd/csrc/tq.cpp (through the C ABI) against the line-by-line
Python restatement of /root/reference/src/tq.zig in oracle/tq_oracle.py, plus every quirk
SURVEY.md 8a lists under findTargetQuality.  No codec and no GPU needed: the pass
(tq.zig:21-38) is a scripted score function."""
import math

import pytest
from hypothesis import given, settings, strategies as st

from oracle import tq_oracle


@pytest.fixture(scope="module")
def tq(hip_lib):
    from oavif_amd import tq as tqmod
    return tqmod


def run_both(tq, fn, **kw):
    calls_a, calls_b = [], []

    def fa(q):
        calls_a.append(q)
        return fn(q)

    def fb(q):
        calls_b.append(q)
        return fn(q)

    a = tq.find_target_quality(fa, **kw)
    b = tq_oracle.find_target_quality(fb, **kw)
    assert calls_a == calls_b, (calls_a, calls_b)
    assert (a.q, a.num_pass, a.buf_q) == (b.q, b.num_pass, b.buf_q)
    assert a.score == b.score
    assert a.history == b.history
    return a


def test_predict_q_closed_form(tq):
    # tq.zig:40-43; values computed by hand in SURVEY.md 8a row A4
    for tgt, q in [(80, 65), (60, 37), (90, 86), (95, 100), (100, 100), (30, 16)]:
        assert tq.predict_q_from_score(tgt) == q == tq_oracle.predict_q_from_score(tgt)
    for t in range(300, 1001):
        assert tq.predict_q_from_score(t / 10) == tq_oracle.predict_q_from_score(t / 10)


def test_survey_hand_trace(tq):
    # tgt 80, tol 2: 65 (84.3) -> binary 55 (77.1) -> linear 59 (79.2) -> accept
    table = {65: 84.3, 55: 77.1, 59: 79.2}
    r = run_both(tq, lambda q: table[q])
    assert (r.q, r.num_pass, r.score) == (59, 3, 79.2)
    assert r.history == [(65, 84.3), (55, 77.1), (59, 79.2)]


def test_tolerance_accepts_score_below_target(tq):
    # quirk 4: |score - tgt| < tol returns immediately, also when score < tgt
    r = run_both(tq, lambda q: 78.5)
    assert (r.q, r.num_pass, r.score) == (65, 1, 78.5)


def test_score_equal_target_takes_else_branch(tq):
    # quirk 3: score == tgt -> lo = q, hi = min(100, q + 0); then tolerance hit
    r = run_both(tq, lambda q: 80.0, tolerance=1.0)
    assert (r.q, r.num_pass) == (65, 1)


def test_duplicate_probe_breaks_before_scoring(tq):
    # quirk 2: a proposed q that was already probed ends the loop without a new pass
    def fn(q):
        return 90.0 if q >= 60 else 10.0
    r = run_both(tq, fn, max_pass=12)
    assert r.num_pass < 12
    assert len({q for q, _ in r.history}) == len(r.history)


def test_final_pick_lowest_q_meeting_target(tq):
    # quirk 9 first half: lowest q with score >= tgt
    def fn(q):
        return 70.0 + 0.2 * q        # q=65 -> 83; crosses 80 at q=50
    r = run_both(tq, fn, tolerance=1.0, max_pass=12)
    assert r.score >= 80.0 or abs(r.score - 80.0) < 1.0


def test_final_pick_when_nothing_meets_target(tq):
    # quirk 9 second half: all scores below target -> highest-scoring probe
    r = run_both(tq, lambda q: 20.0 + q * 0.1, score_tgt=90.0, max_pass=6)
    assert r.q == max(r.history, key=lambda p: p[1])[0]


def test_negative_scores_asymmetry(tq):
    # quirk 9: compares max(score, 0) >= highest but stores the raw (negative) score, so
    # with all-negative scores every probe "wins" in turn and the LAST one is reported
    r = run_both(tq, lambda q: -50.0 + 0.1 * q, score_tgt=80.0, max_pass=4)
    assert r.q == r.history[-1][0]
    assert r.score == r.history[-1][1] < 0


def test_hi_zero_wraps_like_releasefast(tq):
    # quirk 6: lo >= hi - 1 in u32 with hi == 0 wraps -> no collapse exit on that pass.
    # Reach hi == 0: pass 0 above target with a huge error bound, then a probe at q = 0
    # that still scores above target.
    seq = iter([99.0, 99.0, 99.0, 99.0, 99.0, 99.0])
    def fn(q):
        return 99.5
    r = run_both(tq, fn, score_tgt=30.0, tolerance=1.0, max_pass=8)
    assert r.q == min(q for q, _ in r.history)


def test_interpolation_uses_lowest_scoring_points(tq):
    # quirk 7: after sorting by score, points [0],[1],[2] feed the interpolants
    hist = [(65, 90.0), (40, 60.0), (50, 70.0), (58, 85.0)]
    a = tq.interpolate_quantizer(0, 100, hist, 80.0)
    b = tq_oracle.interpolate_quantizer(0, 100, hist, 80.0)
    assert a == b
    # parabola through (60,40),(70,50),(85,58): not through the (90,65) point
    s = sorted(hist, key=lambda p: p[1])[:3]
    r = tq_oracle.quadratic_interpolate([p[1] for p in s], [float(p[0]) for p in s], 80.0)
    assert a == int(min(max(tq_oracle.zig_round(r), 0), 100))


def test_interpolate_degenerate_cases(tq):
    for hist, lo, hi in [([], 10, 21), ([(50, 70.0)], 10, 21),
                         ([(50, 70.0), (60, 70.0)], 0, 100),          # equal scores -> binary
                         ([(50, 70.0), (60, 70.0), (70, 70.0)], 0, 100),
                         ([(50, 70.0), (60, 70.001), (70, 70.002)], 0, 100),  # |denom| < 1e-3
                         ([(10, 5.0), (90, 99.0)], 40, 45)]:           # clamp to [lo, hi]
        assert tq.interpolate_quantizer(lo, hi, hist, 80.0) == \
            tq_oracle.interpolate_quantizer(lo, hi, hist, 80.0)


def test_round_half_away_from_zero(tq):
    # quirk 8: linear interpolant landing exactly on .5 rounds up (Zig @round), not to even
    hist = [(50, 70.0), (51, 90.0)]      # target 80 -> 50.5 -> 51
    assert tq.interpolate_quantizer(0, 100, hist, 80.0) == 51
    hist = [(52, 70.0), (53, 90.0)]      # 52.5 -> 53 (half-to-even would give 52)
    assert tq.interpolate_quantizer(0, 100, hist, 80.0) == 53


def test_probe_error_aborts_search(tq):
    class Boom(Exception):
        pass
    def fn(q):
        raise Boom()
    with pytest.raises(Boom):
        tq.find_target_quality(fn)


def test_max_pass_range(tq):
    from oavif_amd import Ssimu2Error
    with pytest.raises(Ssimu2Error):
        tq.find_target_quality(lambda q: 50.0, max_pass=0)
    with pytest.raises(Ssimu2Error):
        tq.find_target_quality(lambda q: 50.0, max_pass=13)
    r = run_both(tq, lambda q: 50.0, max_pass=1)
    assert r.num_pass == 1


# ---- randomized equivalence: realistic monotone rate-quality curves + noise ------------------

@st.composite
def score_curves(draw):
    a = draw(st.floats(5.0, 60.0))
    b = draw(st.floats(0.2, 1.2))
    knee = draw(st.floats(10.0, 90.0))
    noise = draw(st.lists(st.floats(-1.5, 1.5), min_size=101, max_size=101))
    flat = draw(st.booleans())

    def fn(q):
        base = a + b * q - 0.004 * (q - knee) ** 2 * (1 if q > knee else 0)
        if flat:
            base = round(base)       # plateaus: equal scores at different q
        return base + noise[q]
    return fn


@settings(max_examples=int(__import__('os').environ.get('TQ_EXAMPLES', 400)), deadline=None,
          derandomize='TQ_EXAMPLES' not in __import__('os').environ)
@given(fn=score_curves(), tgt=st.floats(30.0, 100.0), tol=st.floats(1.0, 10.0),
       max_pass=st.integers(1, 12))
def test_cpp_matches_restatement_on_random_curves(tq, fn, tgt, tol, max_pass):
    run_both(tq, fn, score_tgt=tgt, tolerance=tol, max_pass=max_pass)


@settings(max_examples=int(__import__('os').environ.get('TQ_EXAMPLES', 200)), deadline=None,
          derandomize='TQ_EXAMPLES' not in __import__('os').environ)
@given(scores=st.lists(st.floats(-40.0, 100.0), min_size=101, max_size=101),
       tgt=st.floats(30.0, 100.0), max_pass=st.integers(1, 12))
def test_cpp_matches_restatement_on_arbitrary_tables(tq, scores, tgt, max_pass):
    run_both(tq, lambda q: scores[q], score_tgt=tgt, tolerance=1.0, max_pass=max_pass)


# ---- speculative probe fan-out (include/oavif_tq.h): same result as the sequential search -----
def run_speculative(tq, fn, fanout, first_wave=0, **kw):
    """The speculative search against the restatement of tq.zig, plus the wave invariants."""
    waves = []

    def batch(qs):
        waves.append(list(qs))
        return [fn(q) for q in qs]

    res, stats = tq.find_target_quality_speculative(batch, max_fanout=fanout, first_wave_fanout=first_wave, **kw)
    if first_wave and waves:
        assert len(waves[0]) <= first_wave
    ref = tq_oracle.find_target_quality(fn, **kw)
    assert (res.q, res.num_pass, res.buf_q) == (ref.q, ref.num_pass, ref.buf_q)
    assert res.score == ref.score and res.history == ref.history
    issued = [q for w in waves for q in w]
    assert len(issued) == len(set(issued)), "a quantizer was probed twice"
    assert all(0 <= q <= 100 for q in issued)
    assert all(1 <= len(w) <= fanout for w in waves)
    # every wave starts with the quantizer the search is waiting for, in the search's order
    demanded = [w[0] for w in waves]
    order = [q for q, _ in ref.history]
    assert demanded == [q for q in order if q in demanded]
    assert stats.waves == len(waves) and stats.probes_issued == len(issued)
    assert stats.waves + stats.cache_hits == ref.num_pass
    if fanout == 1:
        assert demanded == order
    return res, stats


@pytest.mark.parametrize("fanout", [1, 2, 4, 16])
def test_speculative_hand_trace(tq, fanout):
    table = {65: 84.3, 55: 77.1, 59: 79.2}        # SURVEY 8a hand trace
    res, stats = run_speculative(tq, lambda q: table.get(q, 60.0 + q / 5.0), fanout)
    assert (res.q, res.num_pass) == (59, 3)
    if fanout == 16:
        assert stats.waves < 3                       # 55 = 65 - 2*ceil(4.3) is a pass-0 candidate


@settings(max_examples=int(__import__('os').environ.get('TQ_EXAMPLES', 300)), deadline=None,
          derandomize='TQ_EXAMPLES' not in __import__('os').environ)
@given(fn=score_curves(), tgt=st.floats(30.0, 100.0), tol=st.floats(1.0, 10.0),
       max_pass=st.integers(1, 12), fanout=st.integers(1, 16))
def test_speculative_matches_sequential_on_random_curves(tq, fn, tgt, tol, max_pass, fanout):
    run_speculative(tq, fn, fanout, score_tgt=tgt, tolerance=tol, max_pass=max_pass)
    run_speculative(tq, fn, fanout, first_wave=1 + (max_pass % fanout), score_tgt=tgt, tolerance=tol, max_pass=max_pass)


@settings(max_examples=int(__import__('os').environ.get('TQ_EXAMPLES', 200)), deadline=None,
          derandomize='TQ_EXAMPLES' not in __import__('os').environ)
@given(scores=st.lists(st.floats(-40.0, 100.0), min_size=101, max_size=101),
       tgt=st.floats(30.0, 100.0), max_pass=st.integers(1, 12), fanout=st.integers(1, 16))
def test_speculative_matches_sequential_on_arbitrary_tables(tq, scores, tgt, max_pass, fanout):
    run_speculative(tq, lambda q: scores[q], fanout, score_tgt=tgt, tolerance=1.0, max_pass=max_pass)


def test_speculative_errors(tq):
    from oavif_amd import Ssimu2Error

    class Boom(Exception):
        pass

    def bad(qs):
        raise Boom()
    with pytest.raises(Boom):
        tq.find_target_quality_speculative(bad)
    with pytest.raises(ValueError):
        tq.find_target_quality_speculative(lambda qs: [50.0])   # one score for a wave of four
    for fan in (0, 17):
        with pytest.raises(Ssimu2Error):
            tq.find_target_quality_speculative(lambda qs: [50.0] * len(qs), max_fanout=fan)
    with pytest.raises(Ssimu2Error):
        tq.find_target_quality_speculative(lambda qs: [50.0] * len(qs), max_fanout=4, first_wave_fanout=5)


def test_first_wave_alone_makes_a_one_pass_search_cost_one_probe(tq):
    """first_wave_fanout = 1: a search that ends on the model's guess issues exactly one probe
    (what the sequential search does), a longer one still saves waves from the second wave on."""
    on_target = lambda q: 80.3                                    # noqa: E731  (inside the tolerance at once)
    res, stats = run_speculative(tq, on_target, 8, first_wave=1)
    assert (res.num_pass, stats.waves, stats.probes_issued) == (1, 1, 1)
    res, stats = run_speculative(tq, on_target, 8)
    assert stats.probes_issued > 1                                # the default first wave speculates
    table = {65: 84.3, 55: 77.1, 59: 79.2}                        # SURVEY 8a hand trace: three passes
    res, stats = run_speculative(tq, lambda q: table.get(q, 60.0 + q / 5.0), 16, first_wave=1)
    assert res.num_pass == 3 and stats.waves <= 3 and stats.probes_issued > 3


def test_speculation_saves_waves_on_typical_curves(tq):
    """Not a correctness property: on smooth rate-quality curves around the model of tq.zig:41
    a fan-out of 8 answers a good share of the passes from an earlier wave."""
    import random
    passes = waves = 0
    for seed in range(100):
        r = random.Random(seed)
        off, noise = r.uniform(-8, 8), [r.uniform(-0.5, 0.5) for _ in range(101)]
        fn = lambda q: math.log(max(q, 1) / 6.83) / 0.0282 + off + noise[q]
        res, stats = run_speculative(tq, fn, 8)
        passes += res.num_pass
        waves += stats.waves
    assert waves < 0.85 * passes


# ---- pre-scaling hoist (io.zig:566-617 as functions) -------------------------
def test_prescale_matches_the_reference_formulas_on_every_value(tq):
    import numpy as np
    v8 = np.arange(256, dtype=np.uint8)
    assert np.array_equal(tq.prescale(v8, 10), (v8.astype(np.uint64) * 1023 + 127) // 255)   # io.zig:572
    assert tq.prescale(v8, 10).dtype == np.uint16 and tq.prescale(v8, 10)[-1] == 1023
    assert tq.prescale(v8, 8) is not None and np.array_equal(tq.prescale(v8, 8), v8)        # io.zig:609
    v16 = np.arange(65536, dtype=np.uint16)
    assert np.array_equal(tq.prescale(v16, 10), v16 >> 6)                                     # io.zig:587
    assert np.array_equal(tq.prescale(v16, 8), (v16 >> 8).astype(np.uint8))                   # io.zig:602
    img = np.random.default_rng(3).integers(0, 256, (37, 53, 4), dtype=np.uint8)
    out = tq.prescale(img, 10)
    assert out.shape == img.shape and np.array_equal(out, (img.astype(np.uint32) * 1023 + 127) // 255)
    with pytest.raises(ValueError):
        tq.prescale(img.astype(np.float32), 10)
