"""Builds the blur-mode pin kit (tests/golden/pin_kit/): (ref, dist) pairs on which the scorer's three blur
modes -- the 9-tap FIR, the published fp32 recursion, the same recursion with its multiply-subtract fused --
differ by far more than north_star's +-0.01, each with the three scores the CPU checker
(oracle/ssimu2_oracle.c) gives it.  Someone who can run fssimu2 0.1.1 scores the same files and feeds the
numbers to scripts/pin_blur_mode.py, which says which mode (if any) fssimu2 agrees with.

Two kinds of pair:
  committed  small PNG files in this directory (real libavif/aom -> dav1d round trips made here with Pillow's
             bundled codec, and one synthetic distortion): scores in the low 90s, where the recursion's rounding
             noise shows through max(0, .) even on small frames (gaps 0.1-0.5 points);
  generated  full-size frames (1920x1080, 3840x2160) from oavif_amd.synth seeds + deterministic distortions,
             written as PNG by `scripts/pin_blur_mode.py --write-pairs DIR`; the kit records their sha256 so a
             regenerated file is known to be the scored one.  Here the modes are 0.4-2 points apart.

Run from the repo root (about five minutes on 8 cores):  python3 tests/golden/make_pin_kit.py
The scores are the checker's, not fssimu2's: parity stays unpinned until somebody runs the kit."""
import hashlib
import json
import os
import platform
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from oavif_amd import synth  # noqa: E402
from oracle import ssimu2_oracle as orc  # noqa: E402
import pin_blur_mode as kit  # noqa: E402

KIT = os.path.join(HERE, "pin_kit")


def three_scores(ref, dst):
    return {"fir": orc.compute_ssimu2(ref, dst, orc.BLUR_FIR, omp=True),
            "recursive": orc.compute_ssimu2(ref, dst, orc.BLUR_IIR, omp=True),
            "recursive_fma": orc.compute_ssimu2(ref, dst, orc.BLUR_IIR_FMA, omp=True)}


def variant_scores(ref, dst):
    """Round 5: the score of every entry of the checker's stage-variant catalogue
    (oracle/ssimu2_oracle.py PIN_VARIANTS: the three blur modes plus single stages switched to plausible
    alternatives), so that fssimu2's scores of the kit name the STAGE that differs, not only the blur."""
    return {name: orc.pin_variant_score(ref, dst, name, omp=True) for name in orc.PIN_VARIANTS}


def main():
    orc.build()
    orc.set_num_threads(min(16, os.cpu_count() or 1))
    os.makedirs(KIT, exist_ok=True)
    pairs = []
    # ---- committed pairs -------------------------------------------------------------------------
    ref_a = synth.make_ref(384, 384, 7101)
    ref_b = synth.make_ref(640, 192, 7102)
    ref_c = synth.make_ref(203, 101, 7103)   # odd at every level (203 -> 102 -> 51 -> 26 -> 13 -> 7) and too small for six scales
    committed = [("a384_avif92", "ref_a384.png", ref_a, synth.avif_roundtrip(ref_a, 92, speed=9)[0], "libavif/aom q92 -> dav1d (Pillow)"),
                 ("a384_avif80", "ref_a384.png", ref_a, synth.avif_roundtrip(ref_a, 80, speed=9)[0], "libavif/aom q80 -> dav1d (Pillow)"),
                 ("b640_noise1", "ref_b640.png", ref_b, synth.distort(ref_b, "noise", 0, seed=3), "synth.distort(noise, sigma 1)"),
                 ("c203_blockq2", "ref_c203.png", ref_c, synth.distort(ref_c, "blockq", 2, seed=5), "synth.distort(blockq, 2); odd sizes, five scales: the pyramid-stage variants")]
    for name, ref_file, ref, dst, how in committed:
        dist_file = f"dist_{name}.png"
        for fn, px in ((ref_file, ref), (dist_file, dst)):
            path = os.path.join(KIT, fn)
            if not os.path.exists(path) or not np.array_equal(kit.read_png_rgb8(path), px):
                open(path, "wb").write(kit.png_rgb8(px))
        pairs.append({"name": name, "kind": "committed", "ref": ref_file, "dist": dist_file, "width": ref.shape[1],
                      "height": ref.shape[0], "distortion": how, "sha256_ref": kit.sha256_pixels(ref),
                      "sha256_dist": kit.sha256_pixels(dst), "scores": three_scores(ref, dst),
                      "variant_scores": variant_scores(ref, dst)})
        if name.startswith("c203"):
            pairs[-1]["purpose"] = "stages"   # too small for the blur modes to be >= 8 tolerances apart: not a blur-mode pair
        print(name, pairs[-1]["scores"], flush=True)
    # ---- generated pairs (see pin_blur_mode.GENERATED for the recipes) ----------------------------
    for name in kit.GENERATED:
        ref, dst = kit.generate(name)
        g = kit.GENERATED[name]
        pairs.append({"name": name, "kind": "generated", "ref": f"ref_{name}.png", "dist": f"dist_{name}.png",
                      "width": g["w"], "height": g["h"], "distortion": f"synth.make_ref(seed {g['seed']}) + synth.distort({g['kind']}, {g['strength']})",
                      "sha256_ref": kit.sha256_pixels(ref), "sha256_dist": kit.sha256_pixels(dst), "scores": three_scores(ref, dst),
                      "variant_scores": variant_scores(ref, dst)})
        print(name, pairs[-1]["scores"], flush=True)
    doc = {"what": "blur-mode pin kit: scores of the CPU checker (oracle/ssimu2_oracle.c) in its three blur modes; "
                   "fssimu2 parity UNPINNED until these pairs are scored by fssimu2 0.1.1 (scripts/pin_blur_mode.py)",
           "modes": {"fir": "SSIMU2_BLUR_FIR / OR_BLUR_FIR", "recursive": "SSIMU2_BLUR_RECURSIVE / OR_BLUR_IIR (published scalar order)",
                     "recursive_fma": "SSIMU2_BLUR_RECURSIVE_FMA / OR_BLUR_IIR_FMA (multiply-subtract fused)"},
           "variants": {name: {"stage": v[2], "what": v[3], "implemented_by_the_hip_scorer": name in ("fir", "recursive", "recursive_fma")}
                        for name, v in orc.PIN_VARIANTS.items()},
           "variants_note": "variant_scores[name] per pair: the checker with ONE stage switched to a plausible alternative "
                            "(oracle/ssimu2_oracle.c OR_VAR_*); scripts/pin_blur_mode.py ranks them against fssimu2's scores and "
                            "names the stage of the nearest one",
           "libm_dependent_variants": {
               "names": sorted(n for n in orc.PIN_VARIANTS if n.endswith("+srgb_powf") or n.endswith("+cbrt_libm")),
               "recorded_with": "%s %s" % platform.libc_ver(),
               "note": "these variants call the host libm's powf / cbrtf, which are not bit-stable across glibc versions: their "
                       "recorded scores are INDICATIVE (under the recursive modes a last-bit colour difference moves a score by "
                       "more than the tolerance); tests compare them exactly only on the glibc they were recorded with"},
           "tolerance": 0.01, "pairs": pairs}
    json.dump(doc, open(os.path.join(KIT, "pin_kit.json"), "w"), indent=1)
    for p in pairs:
        s = p["scores"]
        gaps = sorted([abs(s["fir"] - s["recursive"]), abs(s["fir"] - s["recursive_fma"]), abs(s["recursive"] - s["recursive_fma"])])
        print(f"{p['name']:16s} fir {s['fir']:.4f}  recursive {s['recursive']:.4f}  recursive_fma {s['recursive_fma']:.4f}   smallest gap {gaps[0]:.3f}")


if __name__ == "__main__":
    main()
