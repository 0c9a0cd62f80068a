"""CPU only (a minute per 4K line).  The four evaluations of the blur on 1080p and 4K AVIF probes: fp32 9-tap,
published fp32 recursion, the same with fused multiply-subtract, and the operator in fp64.  The recursion's rounding
noise grows with the line length; see profiles/r02_blur_modes_cpu_4k.log."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oavif_amd import synth
from oracle import ssimu2_oracle as orc
orc.build()
for (w, h) in [(1920, 1080), (3840, 2160)]:
    ref = synth.make_ref(w, h, 9100 + w)
    for q in (40, 70):
        dist = synth.avif_roundtrip(ref, q, speed=9)[0]
        t0 = time.time()
        r = {name: orc.compute_ssimu2(ref, dist, mode) for name, mode in
             (("fir", orc.BLUR_FIR), ("iir", orc.BLUR_IIR), ("iir_fma", orc.BLUR_IIR_FMA), ("exact", orc.BLUR_EXACT))}
        print(f"{w}x{h} q={q}: " + "  ".join(f"{k}={v:.4f}" for k, v in r.items()) +
              f"   fir-exact={r['fir'] - r['exact']:+.4f} iir-exact={r['iir'] - r['exact']:+.4f} "
              f"iir_fma-exact={r['iir_fma'] - r['exact']:+.4f} iir-iir_fma={r['iir'] - r['iir_fma']:+.4f}  [{time.time() - t0:.0f}s]", flush=True)
