"""Pure-Python restatement of /root/reference/src/tq.zig:40-210 (search control logic).

TEST INFRASTRUCTURE ONLY: the checker for oavif_amd/csrc/tq.cpp (C ABI include/oavif_tq.h).
Never imported by the product.  Unlike the scorer, this part of the reference is fully in
the tree, so it is restated line by line; it is pinned by the hand-traced example of
SURVEY.md 8a (tgt 80: q 65 -> 55 -> 59) and by the closed-form values of tq.zig:40-43
(tgt 80 -> 65, 60 -> 37, 90 -> 86, >= 95 -> 100); the reference holds no tests of its own.

`probe(q) -> score` stands for computeScoreAtQuality (tq.zig:21-38).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Tuple


def zig_round(v: float) -> float:
    """Zig @round: half away from zero (Python's round() is half-to-even)."""
    return math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5)


def predict_q_from_score(tgt: float) -> int:  # tq.zig:40-43
    q = 6.83 * math.exp(0.0282 * tgt)
    return int(min(100.0, zig_round(q)))


def linear_interpolate(scores, quals, target) -> Optional[float]:  # tq.zig:45-51
    if len(scores) < 2:
        return None
    if scores[1] == scores[0]:
        return None
    t = (target - scores[0]) / (scores[1] - scores[0])
    return quals[0] + (quals[1] - quals[0]) * t


def quadratic_interpolate(scores, quals, target) -> Optional[float]:  # tq.zig:53-71
    if len(scores) < 3:
        return None
    x0, x1, x2 = scores[0], scores[1], scores[2]
    y0, y1, y2 = quals[0], quals[1], quals[2]
    denom = (x0 - x1) * (x0 - x2) * (x1 - x2)
    if abs(denom) < 0.001:
        return None
    a = (x2 * (y1 - y0) + x1 * (y0 - y2) + x0 * (y2 - y1)) / denom
    b = (x2 * x2 * (y0 - y1) + x1 * x1 * (y2 - y0) + x0 * x0 * (y1 - y2)) / denom
    c = (x1 * x2 * (x1 - x2) * y0 + x2 * x0 * (x2 - x0) * y1 + x0 * x1 * (x0 - x1) * y2) / denom
    return a * target * target + b * target + c


def _clamp_round(r: float) -> int:
    # clamp first: @round(+-inf) is +-inf and clamps to 100 / 0 in the reference; a NaN
    # interpolant (0/0, unreachable with finite distinct scores) is mapped to 0 like tq.cpp
    if r != r:
        return 0
    if math.isinf(r):
        return 100 if r > 0 else 0
    return int(min(max(zig_round(r), 0.0), 100.0))


def interpolate_quantizer(lo: int, hi: int, history: List[Tuple[int, float]], target: float) -> int:
    """tq.zig:73-122.  history = [(q, score), ...] in probe order."""
    binary = (lo + hi) // 2
    if not history:
        return binary
    srt = sorted(history, key=lambda p: p[1])  # list.sort is stable, like std.mem.sort
    scores = [p[1] for p in srt]
    quals = [float(p[0]) for p in srt]
    n = len(history)
    if n == 1:
        pred = binary
    elif n == 2:
        r = linear_interpolate(scores, quals, target)
        pred = _clamp_round(r) if r is not None else binary
    else:
        r = quadratic_interpolate(scores, quals, target)
        if r is not None:
            pred = _clamp_round(r)
        else:
            r = linear_interpolate(scores, quals, target)
            pred = _clamp_round(r) if r is not None else binary
    return min(max(pred, lo), hi)


@dataclass
class TQResult:
    q: int = 0
    score: float = 0.0
    num_pass: int = 0
    buf_q: int = -1
    history: List[Tuple[int, float]] = field(default_factory=list)


def find_target_quality(probe: Callable[[int], float], score_tgt: float = 80.0,
                        tolerance: float = 2.0, max_pass: int = 6) -> TQResult:
    """tq.zig:124-210."""
    res = TQResult()
    hist = res.history
    lo, hi = 0, 100
    for p in range(max_pass):
        q = predict_q_from_score(score_tgt) if p == 0 else interpolate_quantizer(lo, hi, hist, score_tgt)
        if any(h[0] == q for h in hist):  # tq.zig:141-148
            break
        score = probe(q)  # tq.zig:150
        res.num_pass += 1
        res.buf_q = q
        hist.append((q, score))
        abs_err = abs(score - score_tgt)
        if p == 0:  # tq.zig:155-165
            err_bound = int(math.ceil(abs_err) * 4.0)
            if score - score_tgt > 0:
                hi = q
                lo = q - err_bound if q > err_bound else 0
            else:
                lo = q
                hi = min(100, q + err_bound)
        if abs_err < tolerance:  # tq.zig:167-168
            res.q, res.score = q, score
            return res
        if p > 0:  # tq.zig:171-176
            if score > score_tgt:
                hi = q
            else:
                lo = q
        if lo >= ((hi - 1) & 0xFFFFFFFF):  # u32 wrap when hi == 0, tq.zig:179
            break
    best_q = None
    best_score = 0.0
    highest_q, highest_score = 0, 0.0
    for (q, s) in hist:  # tq.zig:183-209
        if s >= score_tgt and (best_q is None or q < best_q):
            best_q, best_score = q, s
        if max(s, 0.0) >= highest_score:
            highest_score, highest_q = s, q
    if best_q is not None:
        res.q, res.score = best_q, best_score
    else:
        res.q, res.score = highest_q, highest_score
    return res
