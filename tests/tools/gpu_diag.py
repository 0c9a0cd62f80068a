"""GPU-box diagnostic: HIP vs oracle deviations (scores and the 108 averages)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oavif_amd
from oavif_amd import synth
from oracle import ssimu2_oracle as orc

arr = np.load(os.path.join(ROOT, "tests/golden/pairs_v1.npz"))
meta = json.load(open(os.path.join(ROOT, "tests/golden/pairs_v1.json")))
s = oavif_amd.Ssimu2(0)
ref = arr["ref"]
for p in meta["pairs"]:
    got = s.compute_ssimu2(ref, arr[p["name"]])
    avg, ns = s.last_averages()
    exp = np.array(p["averages_fir"]).reshape(6, 18)
    rel = np.abs(avg - exp) / np.maximum(np.abs(exp), 1e-12)
    i = np.unravel_index(np.argmax(rel), rel.shape)
    print(f"{p['name']:10s} dscore={got - p['score_fir']:+.3e} max rel avg err={rel.max():.3e} at {i} (exp {exp[i]:.3e})")
for (w, h) in [(512, 512)]:
    r = synth.make_ref(w, h, 5)
    for kind, st in [("blockq", 0), ("noise", 0), ("band", 0), ("blur", 2), ("noise", 4)]:
        d = synth.distort(r, kind, st)
        got = s.compute_ssimu2(r, d)
        avg, ns = s.last_averages()
        exp, eavg, _ = orc.compute_ssimu2(r, d, orc.BLUR_FIR, omp=True, return_averages=True)
        rel = np.abs(avg - eavg) / np.maximum(np.abs(eavg), 1e-12)
        print(f"{w}x{h} {kind}{st}: hip={got:.6f} dscore={got - exp:+.3e} max rel avg={rel.max():.3e}")
