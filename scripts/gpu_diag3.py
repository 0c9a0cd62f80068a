import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import oavif_amd
from oavif_amd import synth
s = oavif_amd.Ssimu2(0)
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("amdhip64", "hsa-runtime", "hiprtc", "amd_comgr"))})
print("\n".join(libs))
os.system("readelf -d %s | grep -E 'NEEDED|RUNPATH|RPATH|SONAME'" % os.path.join(ROOT, "oavif_amd/lib/liboavif_hip.so"))
os.system("readelf -d /usr/local/lib/python3.10/dist-packages/torch/lib/libamdhip64.so | grep -E 'SONAME'")
w, h = 640, 480
ref = synth.make_ref(w, h, 0); dst = synth.distort(ref, "blockq", 2)
print("host", s.compute_ssimu2(ref, dst))
tr = torch.from_numpy(ref).cuda(); td = torch.from_numpy(dst).cuda(); torch.cuda.synchronize()
print("roundtrip equal", bool((tr.cpu().numpy() == ref).all()), bool((td.cpu().numpy() == dst).all()))
print("ptrs", hex(tr.data_ptr()), hex(td.data_ptr()), tr.is_contiguous(), tr.dtype, tr.shape)
print("dev", s.score_device(tr.data_ptr(), td.data_ptr(), w, h))
print("dev swapped same", s.score_device(tr.data_ptr(), tr.data_ptr(), w, h))
