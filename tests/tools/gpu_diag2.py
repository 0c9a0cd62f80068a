import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
order = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
if order == "torch_first":
    import torch
    torch.cuda.init()
import oavif_amd
from oavif_amd import synth
s = oavif_amd.Ssimu2(0)
if order != "torch_first":
    import torch
print("order", order, "cuda ok", torch.cuda.is_available())
print([l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "hsa-runtime" in l][::8])
from oracle import ssimu2_oracle as orc
for (w, h) in [(1920, 1080), (2560, 1440), (3840, 2160)]:
    ref = synth.make_ref(w, h, 0)
    dst = synth.distort(ref, "blockq", 2)
    host = s.compute_ssimu2(ref, dst)
    avg, ns = s.last_averages()
    exp, eavg, _ = orc.compute_ssimu2(ref, dst, orc.BLUR_FIR, omp=True, return_averages=True)
    print(w, h, "host-ptr", host, "oracle", exp)
    if abs(host - exp) > 1e-3:
        rel = np.abs(avg - eavg) / np.maximum(np.abs(eavg), 1e-12)
        print(" bad stats (scale,stat):", [(int(a), int(b)) for a, b in zip(*np.nonzero(rel > 1e-3))][:40])
    if torch.cuda.is_available():
        tr = torch.from_numpy(ref).cuda(); td = torch.from_numpy(dst).cuda(); torch.cuda.synchronize()
        print(w, h, "dev-ptr ", s.score_device(tr.data_ptr(), td.data_ptr(), w, h))
        for i in range(3):
            s.enqueue_device(tr.data_ptr(), td.data_ptr(), w, h)
        print(w, h, "dev-ptr x3 enqueue", s.wait())
