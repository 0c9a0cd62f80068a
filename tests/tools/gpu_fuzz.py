"""Randomised parity campaign on the GPU box: HIP scorer vs the CPU oracle (FIR mode) over
random sizes, contents and distortions.  Prints the worst deviations; exits 1 on a violation
of the test tolerances (|dscore| <= 1e-4 per 100 points of |score|, averages rtol 2e-5)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oavif_amd
from oavif_amd import synth
from oracle import ssimu2_oracle as orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
# third argument "recursive": the scorer in ssimu2_ctx_set_blur's published-recursion mode
# against the oracle's OR_BLUR_IIR (default: the fused FIR kernels against OR_BLUR_FIR)
recursive = len(sys.argv) > 3 and sys.argv[3] == "recursive"
OR_MODE = orc.BLUR_IIR if recursive else orc.BLUR_FIR
rng = np.random.default_rng(seed)
s = oavif_amd.Ssimu2(0, blur=oavif_amd._lib.BLUR_RECURSIVE if recursive else None)
worst_score, worst_avg, bad = 0.0, 0.0, []
t0 = time.time()
for i in range(n_cases):
    mode = rng.integers(0, 5)
    if mode == 0:   # tiny
        w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    elif mode == 1:  # wide / tall
        w, h = (int(rng.integers(300, 1500)), int(rng.integers(1, 30))) if rng.random() < 0.5 else (int(rng.integers(1, 30)), int(rng.integers(300, 1500)))
    else:
        w, h = int(rng.integers(8, 700)), int(rng.integers(8, 500))
    kind = rng.integers(0, 4)
    if kind == 0:
        ref = synth.make_ref(w, h, int(rng.integers(0, 1 << 30)))
    elif kind == 1:
        ref = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    elif kind == 2:
        ref = np.full((h, w, 3), rng.integers(0, 256, 3), np.uint8)
        ref[:: max(1, int(rng.integers(1, 9)))] = rng.integers(0, 256, 3)
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        ref = np.stack([(xx * int(rng.integers(1, 9))) & 255, (yy * int(rng.integers(1, 9))) & 255, ((xx ^ yy) * 3) & 255], -1).astype(np.uint8)
    ref = np.ascontiguousarray(ref)
    dk = rng.integers(0, 4)
    if dk == 0:
        dist = np.clip(ref.astype(np.int16) + rng.integers(-int(rng.integers(1, 40)), 40, ref.shape), 0, 255).astype(np.uint8)
    elif dk == 1:
        dist = synth.distort(ref, ["blockq", "blur", "band", "noise"][int(rng.integers(0, 4))], int(rng.integers(0, 5)), seed=i)
    elif dk == 2:
        dist = ref.copy(); dist[rng.integers(0, h), rng.integers(0, w)] ^= 0xFF
    else:
        dist = np.roll(ref, int(rng.integers(1, 4)), axis=int(rng.integers(0, 2)))
    dist = np.ascontiguousarray(dist)
    got = s.compute_ssimu2(ref, dist)
    avg, ns = s.last_averages()
    exp, eavg, ens = orc.compute_ssimu2(ref, dist, OR_MODE, return_averages=True)
    ds = abs(got - exp)
    da = float(np.max(np.abs(avg - eavg) / np.maximum(np.abs(eavg), 1e-9))) if ns else 0.0
    worst_score, worst_avg = max(worst_score, ds), max(worst_avg, da)
    # the score magnifies relative error when it is far below zero (degenerate pairs reach
    # -1000): allow 1e-4 per 100 points of |score|
    ok = ns == ens and ds <= 1e-4 * max(1.0, abs(exp) / 100.0) and np.allclose(avg, eavg, rtol=2e-5, atol=1e-9)
    # the other entry points on the same pair must give the same bits: cached reference, and the
    # decoded-frame hand-off from a random RGB/RGBA layout with random row padding
    s.set_reference(ref)
    ch, pad = int(rng.integers(3, 5)), int(rng.integers(0, 3)) * int(rng.integers(0, 9))
    pitch = w * ch + pad
    buf = rng.integers(0, 256, (h, pitch), dtype=np.uint8)
    view = np.lib.stride_tricks.as_strided(buf, (h, w, ch), (pitch, ch, 1))
    view[..., :3] = dist
    same = (s.score_against_reference(dist) == got and s.score_decoded_against_reference(view) == got
            and np.array_equal(orc.copy_rgb_pixels(view), dist))
    ok = ok and same
    if (i + 1) % 1000 == 0:  # a silent run of several minutes is taken for a hang on the GPU box
        print(f"  {i + 1} cases, worst |dscore| {worst_score:.3e}, violations {len(bad)}", flush=True)
    if not ok:
        bad.append((i, w, h, int(kind), int(dk), got, exp, ns, ens))
print(f"[{'recursive' if recursive else 'fir'} blur] {n_cases} cases in {time.time()-t0:.1f}s: worst |dscore| = {worst_score:.3e}, worst rel avg dev (atol-free) = {worst_avg:.3e}, violations = {len(bad)}")
for b in bad[:20]:
    print("  BAD", b)
sys.exit(1 if bad else 0)
