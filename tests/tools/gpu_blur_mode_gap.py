"""GPU box.  The gap between the scorer's two blur modes (ssimu2_ctx_set_blur: the 9-tap impulse
response vs the published recursion itself) as a function of frame size, on real AVIF probes:
the recursion's rounding noise averages out over more pixels, so the modes agree better on large
frames.  Prints per size the median / 95th percentile / max |score_FIR - score_recursive| and how
often a target-quality search ends on the same quantizer in both modes.
Usage: gpu_blur_mode_gap.py [IMAGES_PER_SIZE]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import oavif_amd
from oavif_amd import synth, tq

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sizes = [(384, 256), (640, 360), (1280, 720), (1920, 1080), (3840, 2160)]
targets = [60.0, 70.0, 80.0, 88.0]
fir = oavif_amd.Ssimu2(0)
rec = oavif_amd.Ssimu2(0, blur=oavif_amd._lib.BLUR_RECURSIVE)
rng = np.random.default_rng(7)
for (w, h) in sizes:
    gaps, same, n, dq, t0 = [], 0, 0, [], time.time()
    for i in range(n_img):
        ref = synth.make_ref(w, h, 9100 + 17 * i + w)
        if i % 3 == 1:
            ref = np.clip(ref.astype(np.int16) + rng.integers(-12, 13, ref.shape), 0, 255).astype(np.uint8)
        cache = {}

        def codec(q):
            if q not in cache:
                cache[q] = synth.avif_roundtrip(ref, q, speed=9)
            return cache[q]
        for tgt in targets:
            a = tq.search_hip(fir, ref, codec, score_tgt=tgt)
            b = tq.search_hip(rec, ref, codec, score_tgt=tgt)
            n += 1
            same += a.q == b.q
            dq.append(abs(a.q - b.q))
        for q in sorted(cache):
            gaps.append(abs(fir.compute_ssimu2(ref, cache[q][0]) - rec.compute_ssimu2(ref, cache[q][0])))
    g = np.array(gaps)
    print(f"{w}x{h}: {len(g)} probes, |score_fir - score_recursive| median {np.median(g):.4f}  95th pct "
          f"{np.percentile(g, 95):.4f}  max {g.max():.4f};  same final quantizer in {same} of {n} searches "
          f"(largest |dq| {max(dq)})  [{time.time() - t0:.0f}s]", flush=True)
fir.close()
rec.close()
