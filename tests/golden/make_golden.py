#!/usr/bin/env python3
"""Generate tests/golden/pairs_v1.npz + pairs_v1.json.

The reference holds no golden vectors for the scorer, and its scorer
(fssimu2 0.1.1) is absent, so these are SELF-ORACLE fixtures: inputs are seeded synthetic
frames, real libavif/aom -> dav1d round trips made with Pillow's bundled codec in the
build container, and expected values are the scores / 108 plane averages of
oracle/ssimu2_oracle.c at the time of generation.  They pin the oracle against silent
drift and give the GPU tests fixed inputs that do not depend on Pillow being present.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oavif_amd import synth  # noqa: E402
from oracle import ssimu2_oracle as orc  # noqa: E402

W, H = 192, 144  # smallest 4:3 size that still evaluates all six scales


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    arrays = {}
    meta = {"w": W, "h": H, "pairs": []}
    ref = synth.make_ref(W, H, seed=7)
    arrays["ref"] = ref
    dists = []
    for q in (20, 49, 65, 86):
        d, size = synth.avif_roundtrip(ref, q, speed=9)
        dists.append((f"avif_q{q}", d, {"avif_bytes": size}))
    dists.append(("blockq2", synth.distort(ref, "blockq", 2), {}))
    dists.append(("noise1", synth.distort(ref, "noise", 1, seed=3), {}))
    dists.append(("blur1", synth.distort(ref, "blur", 1), {}))
    dists.append(("identical", ref.copy(), {}))
    for name, d, extra in dists:
        arrays[name] = d
        fir, avg, ns = orc.compute_ssimu2(ref, d, orc.BLUR_FIR, return_averages=True)
        iir = orc.compute_ssimu2(ref, d, orc.BLUR_IIR)
        meta["pairs"].append({"name": name, "score_fir": fir, "score_iir": iir, "nscales": ns,
                              "averages_fir": avg.reshape(-1).tolist(), **extra})
        print(f"{name:10s} fir={fir:.6f} iir={iir:.6f}")
    # odd-sized crop: exercises edge replication in the downsample and ragged tiles
    oref = ref[:131, :173].copy()
    od = arrays["avif_q49"][:131, :173].copy()
    arrays["odd_ref"], arrays["odd_dist"] = oref, od
    fir, avg, ns = orc.compute_ssimu2(oref, od, orc.BLUR_FIR, return_averages=True)
    meta["odd"] = {"w": 173, "h": 131, "score_fir": fir, "nscales": ns,
                   "score_iir": orc.compute_ssimu2(oref, od, orc.BLUR_IIR),
                   "averages_fir": avg.reshape(-1).tolist()}
    np.savez_compressed(os.path.join(out_dir, "pairs_v1.npz"), **arrays)
    with open(os.path.join(out_dir, "pairs_v1.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
