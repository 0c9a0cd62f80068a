#!/usr/bin/env python3
"""Write tests/golden/pairs_v1_anchors.json from the committed tests/golden/pairs_v1.npz.

Independent anchors for the GPU parity tests (correctness must not rest only on
the oracle mode that mirrors the kernel's operation order).  For every fixture pair:
  score_exact / averages_exact  the oracle with the blur accumulated in fp64 (OR_BLUR_EXACT):
                                the operator both fp32 forms approximate, sharing no rounding
                                sequence with the HIP kernel's blur
  score_iir, score_iir_fma      the published fp32 recursive Gaussian in its two legitimate
                                evaluation orders (OR_BLUR_IIR, OR_BLUR_IIR_FMA)
  gap_fir_minus_iir             score_fir - score_iir: the recorded size of the rounding noise
                                of the published recursion on this fixture (NOT a tolerance
                                the HIP path is held to against fssimu2)
Deterministic: reads the stored frames, needs neither Pillow nor a GPU.
Run from the repo root:  python tests/golden/extend_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ssimu2_oracle as orc  # noqa: E402


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    arrays = np.load(os.path.join(here, "pairs_v1.npz"), allow_pickle=False)
    meta = json.load(open(os.path.join(here, "pairs_v1.json")))
    orc.build()
    out = {"source": "tests/golden/extend_golden.py over pairs_v1.npz; oracle/ssimu2_oracle.c modes "
                     "OR_BLUR_EXACT (fp64 blur), OR_BLUR_IIR, OR_BLUR_IIR_FMA",
           "pairs": []}
    todo = [(p["name"], arrays["ref"], arrays[p["name"]], p["score_fir"]) for p in meta["pairs"]]
    todo.append(("odd", arrays["odd_ref"], arrays["odd_dist"], meta["odd"]["score_fir"]))
    for name, ref, dist, fir in todo:
        ex, avg, ns = orc.compute_ssimu2(ref, dist, orc.BLUR_EXACT, return_averages=True)
        iir = orc.compute_ssimu2(ref, dist, orc.BLUR_IIR)
        iir_fma = orc.compute_ssimu2(ref, dist, orc.BLUR_IIR_FMA)
        out["pairs"].append({"name": name, "score_exact": ex, "averages_exact": avg.reshape(-1).tolist(),
                             "score_iir": iir, "score_iir_fma": iir_fma,
                             "gap_fir_minus_exact": fir - ex, "gap_fir_minus_iir": fir - iir})
        print(f"{name:10s} exact={ex:.6f} fir-exact={fir - ex:+.2e} fir-iir={fir - iir:+.2e} iir-iir_fma={iir - iir_fma:+.2e}")
    with open(os.path.join(here, "pairs_v1_anchors.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
