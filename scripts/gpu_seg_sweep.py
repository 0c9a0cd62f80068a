"""Segment-length sweep of the marching kernel (instrumented build's experiment knob): HBM-fed
kernel time at 4K for rows-per-workgroup at scale 0 x at the other scales.  The rule the product
uses is march_seg_rows() in ssimu2_hip.hip; this script is how it was chosen."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import oavif_amd  # noqa: E402
from oavif_amd import synth  # noqa: E402

w, h = 3840, 2160
NP = 8
ref = synth.make_ref(w, h, 0)
dst = synth.distort(ref, "blockq", 2)
tr = torch.from_numpy(ref).cuda().contiguous()
td = torch.from_numpy(dst).cuda().contiguous()
pairs = [(tr, td)]
for k in range(1, NP):
    a, b = torch.roll(tr, k * w // NP, 1), torch.roll(td, k * w // NP, 1)
    pairs.append((a.contiguous(), b.contiguous()))
torch.cuda.synchronize()
pr, pd = [a.data_ptr() for a, _ in pairs], [b.data_ptr() for _, b in pairs]
s = oavif_amd.Ssimu2(0, instrumented=True)
for _ in range(200):
    s.enqueue_device(pr[0], pd[0], w, h)
s.wait()
seg0s = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,72,90,108,120,135,154".split(","))]
tails = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,32,48,64,96".split(","))]
for s0 in seg0s:
    row = []
    for t in tails:
        s.set_segment_rows(s0, t)
        sc = s.score_device(pr[0], pd[0], w, h)
        us = min(s.time_march_rotating(pr, pd, w, h, 48) for _ in range(3)) * 1e3
        row.append(f"{t}:{us:.1f}")
    print(f"seg0={s0:3d}  " + "  ".join(row) + f"   score={sc:.12f}", flush=True)
s.close()
