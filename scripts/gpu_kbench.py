"""Kernel micro-bench on the GPU box: per-scale fused-kernel time and whole-score time at 4K."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import oavif_amd
from oavif_amd import synth
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
ref = synth.make_ref(w, h, 0); dst = synth.distort(ref, "blockq", 2)
tr = torch.from_numpy(ref).cuda().contiguous(); td = torch.from_numpy(dst).cuda().contiguous()
torch.cuda.synchronize()
s = oavif_amd.Ssimu2(0)
score = s.score_device(tr.data_ptr(), td.data_ptr(), w, h)
_, ns = s.last_averages()
ks = [s.time_stage(tr.data_ptr(), td.data_ptr(), w, h, st, 30) * 1e3 for st in range(3)]
ms, _ = s.time_device(tr.data_ptr(), td.data_ptr(), w, h, 50)
tag = f"seg={os.environ.get('OAVIF_AMD_SEG_ROWS','auto')}"
print(f"{tag}: score={score:.9f} stage_us[pyramid,march,finalize]={[round(k,1) for k in ks]} sum={sum(ks):.1f} whole_score_us={ms/50*1e3:.1f} MP/s={w*h/1e6/(ms/50/1e3):.0f}")
s.set_reference_device(tr.data_ptr(), w, h)
import time
for _ in range(5):
    s.enqueue_against_reference_device(td.data_ptr())
sc2 = s.wait()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100):
    s.enqueue_against_reference_device(td.data_ptr())
sc2 = s.wait(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
print(f"cached-reference score: {sc2:.9f} (same={sc2 == score}) {dt*1e6:.1f} us/score  {w*h/1e6/dt:.0f} MP/s")
