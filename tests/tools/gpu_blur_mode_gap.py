"""GPU box.  The gap between the scorer's two blur modes (ssimu2_ctx_set_blur: the 9-tap impulse
response vs the published recursion itself) as a function of frame size, on real AVIF probes:
the recursion's rounding noise averages out over more pixels, so the modes agree better on large
frames.  Prints per size the median / 95th percentile / max |score_FIR - score_recursive| and how
often a target-quality search ends on the same quantizer in both modes.
Usage: gpu_blur_mode_gap.py [IMAGES_PER_SIZE] [--only WxH] [--json PATH] [--bridge]
--bridge: the probes come from libavif's C API under the reference's calls and defaults (oavif_amd.avif_bridge: tune=iq,
one encoder thread, 8-bit here) instead of Pillow's plugin.
--json writes one record per search (seed, target, the quantizer and score each mode ends on, passes) -- the
4K record of round 4 is profiles/r04_4k_search_both_modes.json (24 searches: 6 images x 4 targets)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import oavif_amd
from oavif_amd import synth, tq

import json
argv = sys.argv[1:]
json_path = only = None
bridge = "--bridge" in argv
if bridge:
    argv.remove("--bridge")
    from oavif_amd import avif_bridge as ab, cli
    assert ab.available(), ab.why_unavailable()
    enc_o = cli.AvifEncOptions()
    enc_o.tenbit = False
if "--json" in argv:
    json_path = argv[argv.index("--json") + 1]
    del argv[argv.index("--json"):argv.index("--json") + 2]
if "--only" in argv:
    only = tuple(int(v) for v in argv[argv.index("--only") + 1].split("x"))
    del argv[argv.index("--only"):argv.index("--only") + 2]
n_img = int(argv[0]) if argv else 6
sizes = [(384, 256), (640, 360), (1280, 720), (1920, 1080), (3840, 2160)]
if only:
    sizes = [only]
records = []
targets = [60.0, 70.0, 80.0, 88.0]
fir = oavif_amd.Ssimu2(0)
rec = oavif_amd.Ssimu2(0, blur=oavif_amd._lib.BLUR_RECURSIVE)
fma = oavif_amd.Ssimu2(0, blur=oavif_amd._lib.BLUR_RECURSIVE_FMA)
rng = np.random.default_rng(7)
for (w, h) in sizes:
    gaps, same, n, dq, t0 = [], 0, 0, [], time.time()
    for i in range(n_img):
        ref = synth.make_ref(w, h, 9100 + 17 * i + w)
        if i % 3 == 1:
            ref = np.clip(ref.astype(np.int16) + rng.integers(-12, 13, ref.shape), 0, 255).astype(np.uint8)
        cache = {}
        src_img = ab.EncoderSource(ref, 8, enc_o) if bridge else None

        def codec(q):
            if q not in cache:
                if bridge:
                    data = src_img.encode(enc_o, q)
                    cache[q] = (ab.decode_rgb8(data), len(data))
                else:
                    cache[q] = synth.avif_roundtrip(ref, q, speed=9)
            return cache[q]
        for tgt in targets:
            a = tq.search_hip(fir, ref, codec, score_tgt=tgt)
            b = tq.search_hip(rec, ref, codec, score_tgt=tgt)
            n += 1
            same += a.q == b.q
            dq.append(abs(a.q - b.q))
            if json_path:
                c = tq.search_hip(fma, ref, codec, score_tgt=tgt)
                records.append({"width": w, "height": h, "image": i, "seed": 9100 + 17 * i + w, "extra_noise": i % 3 == 1,
                                "target": tgt, "q_fir": a.q, "q_recursive": b.q, "q_recursive_fma": c.q,
                                "score_fir": round(a.score, 4), "score_recursive": round(b.score, 4),
                                "score_recursive_fma": round(c.score, 4), "passes_fir": a.num_pass,
                                "passes_recursive": b.num_pass, "passes_recursive_fma": c.num_pass,
                                "probes_fir": [[q, round(sc_, 4)] for q, sc_ in a.history],
                                "probes_recursive": [[q, round(sc_, 4)] for q, sc_ in b.history]})
        for q in sorted(cache):
            gaps.append(abs(fir.compute_ssimu2(ref, cache[q][0]) - rec.compute_ssimu2(ref, cache[q][0])))
    g = np.array(gaps)
    print(f"{w}x{h}: {len(g)} probes, |score_fir - score_recursive| median {np.median(g):.4f}  95th pct "
          f"{np.percentile(g, 95):.4f}  max {g.max():.4f};  same final quantizer in {same} of {n} searches "
          f"(largest |dq| {max(dq)})  [{time.time() - t0:.0f}s]", flush=True)
if json_path:
    nsame = sum(r["q_fir"] == r["q_recursive"] for r in records)
    json.dump({"what": "target-quality searches (tq.zig:124-210 restated, real AVIF probes through " +
                       ("libavif's C API with the reference's calls and defaults: YUV444, tune=iq, speed 9, one thread, 8-bit) " if bridge
                        else "Pillow's libavif/aom speed 9) ") +
                       "driven by the HIP scorer in each blur mode; which mode fssimu2 0.1.1 follows is unknown (parity unpinned)",
               "searches": len(records), "same_quantizer_fir_vs_recursive": nsame,
               "same_quantizer_recursive_vs_recursive_fma": sum(r["q_recursive"] == r["q_recursive_fma"] for r in records),
               "largest_abs_dq_fir_vs_recursive": max(abs(r["q_fir"] - r["q_recursive"]) for r in records),
               "records": records}, open(json_path, "w"), indent=1)
fir.close()
rec.close()
fma.close()
