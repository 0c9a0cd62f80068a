"""Quantizer-match campaign on the GPU box: the same target-quality search (tq.zig:124-210) driven
by the HIP scorer and by the CPU oracle over many synthetic images x targets, real AVIF probes
(Pillow libavif/aom, speed 9).  Requires identical probe sequences and final quantizers; reports
the largest score difference seen on any probe.
Usage: gpu_search_campaign.py [N_IMAGES] [fir|recursive] [pillow|bridge]   (recursive: ssimu2_ctx_set_blur's
published-recursion mode against the oracle's OR_BLUR_IIR; bridge: the probes come from libavif's C API with the
reference's calls and defaults -- oavif_amd.avif_bridge: tune=iq, one thread, every third image RGBA with
--quality-alpha 90 -- and the HIP side runs the search of the CLI / batch path, tq.search_hip_frames: libavif's
RGB(A) rows handed to the scorer as they are)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import oavif_amd
from oavif_amd import synth, tq
from oracle import ssimu2_oracle as orc
from oracle import tq_oracle

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 40
recursive = len(sys.argv) > 2 and sys.argv[2] == "recursive"
bridge = len(sys.argv) > 3 and sys.argv[3] == "bridge"
if bridge:
    from oavif_amd import avif_bridge as ab, cli
    assert ab.available(), ab.why_unavailable()
orc.build()
rng = np.random.default_rng(2026)
sizes = [(384, 256), (320, 320), (500, 281), (257, 199), (640, 360)]
targets = [55.0, 65.0, 75.0, 80.0, 85.0, 92.0]
bad, worst, total_passes, t0 = [], 0.0, 0, time.time()
# real photographs where the image has them (scikit-learn's two sample images and three crops / flips of each):
# they take the place of the first synthetic images
sys.path.insert(0, os.path.join(ROOT, "tests"))
photos = []
try:
    from photo_ladder import photographs
    for _name, ph in photographs():
        photos += [ph, np.ascontiguousarray(ph[:, ::-1]), np.ascontiguousarray(ph[40:340, 100:600]), np.ascontiguousarray(ph[::-1, 200:])]
except Exception:  # noqa: BLE001
    photos = []
print(f"{len(photos)} of the {n_images} images are real photographs (scikit-learn's samples and crops of them)", flush=True)
with oavif_amd.Ssimu2(0, blur=oavif_amd._lib.BLUR_RECURSIVE if recursive else None) as s:
    for i in range(n_images):
        w, h = sizes[i % len(sizes)]
        if i < len(photos):
            ref = photos[i]
            h, w = ref.shape[:2]
        else:
            ref = synth.make_ref(w, h, 7000 + i)
        if i % 3 == 1 and i >= len(photos):   # busier content
            ref = np.clip(ref.astype(np.int16) + rng.integers(-12, 13, ref.shape), 0, 255).astype(np.uint8)
        cache = {}

        def codec(q):
            if q not in cache:
                cache[q] = synth.avif_roundtrip(ref, q, speed=9)
            return cache[q]
        if bridge:
            o = cli.AvifEncOptions()
            o.tenbit = False
            src = ref
            if i % 3 == 2:   # an alpha plane: the decoded frame is RGBA and its alpha is dropped on the device
                o.quality_alpha = 90
                src = np.dstack([ref, np.tile(np.linspace(40, 255, w, dtype=np.uint8), (h, 1))])
            data = {}

            def codec(q):
                if q not in cache:
                    data[q] = ab.encode(src, 8, o, q)
                    cache[q] = (ab.decode_rgb8(data[q]), len(data[q]))
                return cache[q]

            def codec_frame(q):
                codec(q)
                return ab.decode_common(data[q]), len(data[q])
        for tgt in targets:
            g = (tq.search_hip_frames(s, ref, codec_frame, score_tgt=tgt) if bridge
                 else tq.search_hip(s, ref, codec, score_tgt=tgt))
            c = tq_oracle.find_target_quality(
                lambda q: orc.compute_ssimu2(ref, codec(q)[0], orc.BLUR_IIR if recursive else orc.BLUR_FIR), score_tgt=tgt)
            same = [q for q, _ in g.history] == [q for q, _ in c.history] and g.q == c.q and g.num_pass == c.num_pass
            d = max(abs(a[1] - b[1]) for a, b in zip(g.history, c.history))
            worst = max(worst, d)
            total_passes += g.num_pass
            if not same:
                bad.append((i, w, h, tgt, g.history, c.history))
        if (i + 1) % 10 == 0:
            print(f"  {i + 1} images, {total_passes} passes, worst |dscore| {worst:.3e}, mismatches {len(bad)}", flush=True)
print(f"[{'recursive' if recursive else 'fir'} blur, {'libavif bridge (tune=iq, hand-off of libavif rows)' if bridge else 'Pillow codec'}] {n_images} images x {len(targets)} targets = {n_images * len(targets)} searches, {total_passes} passes in "
      f"{time.time() - t0:.1f}s: probe sequences and final quantizers identical in "
      f"{n_images * len(targets) - len(bad)} of {n_images * len(targets)}; worst |dscore| on a probe = {worst:.3e}")
for b in bad[:10]:
    print("  MISMATCH", b)
sys.exit(1 if bad else 0)
