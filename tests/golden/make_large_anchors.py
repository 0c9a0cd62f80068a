"""Anchor scores of seeded synthetic pairs at the BASELINE frame sizes (1920x1080, 3840x2160,
7680x4320), computed by the CPU checker (oracle/ssimu2_oracle.c, OpenMP build) in three blur
modes: FIR (the kernels' contract), EXACT (the same operator accumulated in fp64: shares no
rounding sequence with the kernels) and IIR (the published fp32 recursion).  Written to
tests/golden/large_anchors.json, so that the GPU box compares against committed numbers instead
of a live CPU run (an 8K fp64-blur score takes a minute).  The frames themselves are not stored:
oavif_amd.synth regenerates them from the seeds.

    python tests/golden/make_large_anchors.py          (~6 minutes on 8 cores)

fssimu2 parity stays UNPINNED: these are this repo's checker's numbers, not the reference's.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oavif_amd import synth  # noqa: E402
from oracle import ssimu2_oracle as orc  # noqa: E402

CASES = [  # (w, h, seed, kind, strength)
    (1920, 1080, 11, "blockq", 2), (1920, 1080, 12, "noise", 2),
    (3840, 2160, 21, "blockq", 1), (3840, 2160, 22, "blur", 2),
    (7680, 4320, 91, "blockq", 1), (7680, 4320, 92, "blockq", 3),
]


def main():
    orc.build()
    orc.set_num_threads(os.cpu_count() or 8)
    out = {"generator": "tests/golden/make_large_anchors.py", "numpy": np.__version__, "cases": []}
    for w, h, seed, kind, strength in CASES:
        ref = synth.make_ref(w, h, seed)
        dist = synth.distort(ref, kind, strength, seed=seed)
        t0 = time.time()
        rec = {"w": w, "h": h, "seed": seed, "kind": kind, "strength": strength,
               "ref_crc": int(np.bitwise_xor.reduce(ref.reshape(-1).astype(np.uint32) * np.arange(1, ref.size + 1, dtype=np.uint32) & 0xFFFFFFFF)),
               "score_fir": orc.compute_ssimu2(ref, dist, orc.BLUR_FIR, omp=True),
               "score_exact": orc.compute_ssimu2(ref, dist, orc.BLUR_EXACT, omp=True),
               "score_iir": orc.compute_ssimu2(ref, dist, orc.BLUR_IIR, omp=True),
               "score_iir_fma": orc.compute_ssimu2(ref, dist, orc.BLUR_IIR_FMA, omp=True)}
        out["cases"].append(rec)
        print(rec, f"{time.time() - t0:.0f} s", flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "large_anchors.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
