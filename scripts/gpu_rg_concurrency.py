"""How the recursive mode's reference-cached pass scales with the number of scorer contexts in flight on ONE GPU
(the probes of a speculative search, the worker threads of the batch driver): K contexts, each with its own
reference and stream, enqueue 4K passes round-robin from one host thread (enqueue / wait entry points, frames
resident in HBM); aggregate passes per second against one context alone.  Also at 1080p, where a single pass is
bound by the length of its serial chains and leaves most of the chip idle.
    python3 scripts/gpu_rg_concurrency.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import oavif_amd  # noqa: E402
from oavif_amd import _lib, synth  # noqa: E402


def run(w, h, ks=(1, 2, 3, 4, 6)):
    ref = synth.make_ref(w, h, 0)
    tr = torch.from_numpy(ref).cuda().contiguous()
    dists = [torch.from_numpy(synth.distort(ref, kind, st)).cuda().contiguous()
             for kind, st in (("blockq", 2), ("noise", 2), ("blur", 1), ("blockq", 1))]
    torch.cuda.synchronize()
    base = None
    for mode, label in ((_lib.BLUR_RECURSIVE, "recursive"), (None, "fir")):
        for k in ks:
            ctxs = [oavif_amd.Ssimu2(0, blur=mode) for _ in range(k)]
            for c in ctxs:
                c.set_reference_device(tr.data_ptr(), w, h)
            want = None
            best = 1e9
            for _rep in range(3):
                n = 24 * k
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    c = ctxs[i % k]
                    if i >= k:
                        c.wait()
                    c.enqueue_against_reference_device(dists[(i // k) % len(dists)].data_ptr())
                got = [c.wait() for c in ctxs]
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / n)
                want = want or got
            if k == 1:
                base = best
            print(f"{w}x{h} {label:9s} {k} context{'s' if k > 1 else ' '}: {best * 1e3:.3f} ms per pass aggregate "
                  f"({w * h / 1e6 / best:8.0f} MP/s), {base / best:.2f} x one context", flush=True)
            for c in ctxs:
                c.close()


run(3840, 2160)
run(1920, 1080)
