"""Which pairs of scorer contexts (HIP streams) overlap their scores?  ms per 4K score for every pair
among N contexts created back to back, and for each context alone.
    python3 scripts/gpu_stream_pairs.py [N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import oavif_amd  # noqa: E402
from oavif_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
w, h = 3840, 2160
ref = synth.make_ref(w, h, 0)
dst = synth.distort(ref, "blockq", 2)
tr, td = torch.from_numpy(ref).cuda(), torch.from_numpy(dst).cuda()
pr, pd = tr.data_ptr(), td.data_ptr()
ctx = [oavif_amd.Ssimu2(0) for _ in range(N)]


def ms(group, n=160):
    for c in group:
        c.enqueue_device(pr, pd, w, h)
        c.wait()
    t = time.perf_counter()
    for i in range(n):
        group[i % len(group)].enqueue_device(pr, pd, w, h)
    for c in group:
        c.wait()
    return (time.perf_counter() - t) / n * 1e3


ms(ctx[:2], 600)  # clocks
print("alone:", " ".join(f"{ms([c]):.4f}" for c in ctx))
for rep in range(2):
    for i in range(N):
        print(f"pair {i}-*:", " ".join(f"{ms([ctx[i], ctx[j]]):.4f}" if j > i else "  --  " for j in range(N)))
