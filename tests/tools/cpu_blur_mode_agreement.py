"""CPU only.  How much does the one thing we cannot pin -- the blur arithmetic of fssimu2 0.1.1 --
matter for the search result?  Runs the same target-quality searches with the oracle's FIR blur
(what the HIP kernels implement) and with the published fp32 recursive Gaussian of libjxl's
SSIMULACRA2 (oracle mode BLUR_IIR), real AVIF probes, and counts how often the final quantizer and
the probe sequence agree.  Usage: cpu_blur_mode_agreement.py [N_IMAGES]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oavif_amd import synth
from oracle import ssimu2_oracle as orc
from oracle import tq_oracle

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 30
orc.build()
rng = np.random.default_rng(2026)
sizes = [(384, 256), (320, 320), (500, 281), (257, 199), (640, 360)]
targets = [55.0, 65.0, 75.0, 80.0, 85.0, 92.0]
same_q = same_seq = n = 0
gaps, dq = [], []
same_q_ii, gaps_ii, dq_ii = 0, [], []
t0 = time.time()
for i in range(n_images):
    w, h = sizes[i % len(sizes)]
    ref = synth.make_ref(w, h, 7000 + i)
    if i % 3 == 1:
        ref = np.clip(ref.astype(np.int16) + rng.integers(-12, 13, ref.shape), 0, 255).astype(np.uint8)
    cache, sc = {}, {}

    def codec(q):
        if q not in cache:
            cache[q] = synth.avif_roundtrip(ref, q, speed=9)
        return cache[q]

    def score(q, mode):
        if (q, mode) not in sc:
            sc[(q, mode)] = orc.compute_ssimu2(ref, codec(q)[0], mode)
        return sc[(q, mode)]
    for tgt in targets:
        a = tq_oracle.find_target_quality(lambda q: score(q, orc.BLUR_FIR), score_tgt=tgt)
        b = tq_oracle.find_target_quality(lambda q: score(q, orc.BLUR_IIR), score_tgt=tgt)
        c = tq_oracle.find_target_quality(lambda q: score(q, orc.BLUR_IIR_FMA), score_tgt=tgt)
        n += 1
        same_q += a.q == b.q
        same_seq += [q for q, _ in a.history] == [q for q, _ in b.history]
        dq.append(abs(a.q - b.q))
        same_q_ii += b.q == c.q
        dq_ii.append(abs(b.q - c.q))
    gaps += [abs(score(q, orc.BLUR_FIR) - score(q, orc.BLUR_IIR)) for q in cache]
    gaps_ii += [abs(score(q, orc.BLUR_IIR_FMA) - score(q, orc.BLUR_IIR)) for q in cache]
    if (i + 1) % 10 == 0:
        print(f"  {i + 1} images: same final q {same_q}/{n}, same probe sequence {same_seq}/{n}", flush=True)
gaps = np.array(gaps)
print(f"{n} searches in {time.time() - t0:.0f}s: final quantizer equal in {same_q} ({100.0 * same_q / n:.1f} %), "
      f"whole probe sequence equal in {same_seq}; largest |dq| {max(dq)}; "
      f"|score_FIR - score_IIR| over {len(gaps)} probes: median {np.median(gaps):.4f}, "
      f"95th pct {np.percentile(gaps, 95):.4f}, max {gaps.max():.4f}")
gaps_ii = np.array(gaps_ii)
print(f"two recursive evaluation orders against each other (unfused vs fused multiply-subtract): final "
      f"quantizer equal in {same_q_ii} of {n} ({100.0 * same_q_ii / n:.1f} %), largest |dq| {max(dq_ii)}; "
      f"|dscore| median {np.median(gaps_ii):.4f}, 95th pct {np.percentile(gaps_ii, 95):.4f}, max {gaps_ii.max():.4f}")
