"""Threading contract stress: N threads, one scorer context each, score different pairs
concurrently (host-pointer and reference-cached paths); every result must equal the
single-threaded result bit for bit."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oavif_amd
from oavif_amd import synth

NT, REPS = 8, 40
pairs = []
for i in range(NT):
    w, h = 300 + 37 * i, 200 + 23 * i
    ref = synth.make_ref(w, h, 700 + i)
    pairs.append((ref, [synth.distort(ref, k, s, seed=i) for k, s in (("blockq", 1), ("noise", 2), ("blur", 1))]))
with oavif_amd.Ssimu2(0) as s0:
    expect = [[s0.compute_ssimu2(ref, d) for d in ds] for ref, ds in pairs]
errors = []

def work(i):
    try:
        ref, ds = pairs[i]
        with oavif_amd.Ssimu2(0) as s:
            for r in range(REPS):
                for j, d in enumerate(ds):
                    got = s.compute_ssimu2(ref, d) if r % 2 == 0 else None
                    if r % 2:
                        s.set_reference(ref)
                        got = s.score_against_reference(d)
                    if got != expect[i][j]:
                        errors.append((i, r, j, got, expect[i][j]))
    except Exception as e:
        errors.append((i, repr(e)))

t0 = time.time()
th = [threading.Thread(target=work, args=(i,)) for i in range(NT)]
[t.start() for t in th]; [t.join() for t in th]
print(f"{NT} threads x {REPS} reps x 3 pairs in {time.time()-t0:.1f}s: mismatches/errors = {len(errors)}")
for e in errors[:10]:
    print("  ", e)
sys.exit(1 if errors else 0)
