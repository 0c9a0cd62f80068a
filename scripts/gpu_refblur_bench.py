"""The per-pass kernel of a FIR-mode search (k_march_refblur) on the GPU box: ms per reference-cached pass,
rotating over NP distorted frames (HBM-fed), one stream, and the score's bits (A/B builds must not move them).
    python3 scripts/gpu_refblur_bench.py [w h [NP]]        library: OAVIF_AMD_LIB"""
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import oavif_amd  # noqa: E402
from oavif_amd import synth  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ref = synth.make_ref(w, h, 0)
dst = synth.distort(ref, "blockq", 2)
tr = torch.from_numpy(ref).cuda().contiguous()
td = torch.from_numpy(dst).cuda().contiguous()
dists = [td] + [torch.roll(td, k * w // NP, 1).contiguous() for k in range(1, NP)]
torch.cuda.synchronize()
pd = [d.data_ptr() for d in dists]
s = oavif_amd.Ssimu2(0)
s.set_reference_device(tr.data_ptr(), w, h)
scores = []
for p in pd:
    s.enqueue_against_reference_device(p)
    scores.append(s.wait())
for i in range(200):  # clocks
    s.enqueue_against_reference_device(pd[i % NP])
s.wait()
best = 1e9
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(64):
        s.enqueue_against_reference_device(pd[i % NP])
    s.wait()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 64)
bits = " ".join(struct.pack(">d", v).hex() for v in scores[:3])
print(f"refblur_bench {w}x{h}: cached pass {best * 1e3:.4f} ms   score bits {bits}")
s.close()
