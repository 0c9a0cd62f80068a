"""Soak of the scorer's contract on the GPU box: for SECONDS (default 150) six threads, one scorer context each,
score frames of changing sizes in changing blur modes through all entry points (pair, cached reference, strided
RGBA hand-off, and -- round 5 -- the cached-reference pass from a page-locked buffer of ssimu2_host_alloc that is
allocated, filled and freed every time), re-creating their contexts now and then -- what a batch host does for hours.  Every score
must equal, bit for bit, the one a single context computed for the same (size, mode, pair) before the threads
started; device memory must be back where it was when the contexts are gone.
Usage: gpu_soak.py [SECONDS]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oavif_amd
from oavif_amd import _lib, synth

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
NT = 6
SIZES = [(96, 64), (333, 211), (640, 360), (1000, 563), (1920, 1080), (129, 2049), (2560, 1440)]
MODES = [None, _lib.BLUR_RECURSIVE, _lib.BLUR_RECURSIVE_FMA]


def free_mb():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / (1 << 20)


pairs = {}
for k, (w, h) in enumerate(SIZES):
    ref = synth.make_ref(w, h, 4000 + k)
    dists = [synth.distort(ref, kind, s, seed=k) for kind, s in (("blockq", 1), ("noise", 2), ("blur", 2))]
    pairs[(w, h)] = (ref, dists)
expect = {}
with oavif_amd.Ssimu2(0) as s0:
    for mode in MODES:
        s0.set_blur(_lib.BLUR_FIR if mode is None else mode)
        for wh, (ref, dists) in pairs.items():
            for j, d in enumerate(dists):
                expect[(wh, mode, j)] = s0.compute_ssimu2(ref, d)
    s0.set_blur(_lib.BLUR_FIR)
base = free_mb()
errors, counts = [], [0] * NT
stop = time.time() + SECONDS


def work(i):
    rng = np.random.default_rng(100 + i)
    s = None
    try:
        while time.time() < stop:
            if s is None or rng.random() < 0.02:      # a fresh context now and then
                if s is not None:
                    pins = {}                          # its page-locked buffers go with it
                    s.close()
                s = oavif_amd.Ssimu2(0)
                pins = {}
            mode = MODES[int(rng.integers(0, 3))]
            s.set_blur(_lib.BLUR_FIR if mode is None else mode)
            wh = SIZES[int(rng.integers(0, len(SIZES)))]
            ref, dists = pairs[wh]
            entry = int(rng.integers(0, 4))
            if entry:
                s.set_reference(ref)
            for j, d in enumerate(dists):
                if entry == 0:
                    got = s.compute_ssimu2(ref, d)
                elif entry == 1:
                    got = s.score_against_reference(d)
                elif entry == 3:                       # decode-into-pinned, as csrc/oavif_host.c does: one buffer per
                    if pins.get(d.shape) is None:      # context and frame size, reused (ssimu2_host_free synchronises the
                        pins[d.shape] = s.host_alloc(d.shape)   # whole device: never allocate / free per score)
                    pin = pins[d.shape]
                    pin[...] = d
                    got = s.score_against_reference(pin)
                else:                                  # libavif's RGBA rows with padding behind every row
                    h, w, _ = d.shape
                    rows = np.zeros((h, w * 4 + 24), np.uint8)
                    rows[:, : w * 4].reshape(h, w, 4)[..., :3] = d
                    rows[:, : w * 4].reshape(h, w, 4)[..., 3] = 200
                    got = s.score_decoded_against_reference(rows.reshape(-1), w * 4 + 24, 4)
                counts[i] += 1
                if got != expect[(wh, mode, j)]:
                    errors.append((i, wh, mode, entry, j, got, expect[(wh, mode, j)]))
    except Exception as e:  # noqa: BLE001
        errors.append((i, repr(e)))
    finally:
        if s is not None:
            s.close()


t0 = time.time()
th = [threading.Thread(target=work, args=(i,)) for i in range(NT)]
[t.start() for t in th]
[t.join() for t in th]
after = free_mb()
print(f"{NT} threads, {time.time() - t0:.0f} s, {sum(counts)} scores over {len(SIZES)} sizes x 3 blur modes x 4 entry points with "
      f"contexts re-created now and then: mismatches / errors = {len(errors)}; device memory free before / after: "
      f"{base:.0f} / {after:.0f} MB (difference {base - after:+.0f} MB)")
for e in errors[:10]:
    print("  ", e)
sys.exit(1 if errors or abs(base - after) > 256 else 0)
