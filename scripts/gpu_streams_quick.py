"""4K pair score throughput for 1..3 scorer contexts (streams), inputs resident, two enqueue
patterns: 'all' = every score enqueued up front (deep queues), 'wait' = a context's previous
score is waited for before its next is enqueued (what a host thread per context does)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oavif_amd
from oavif_amd import synth

W, H = 3840, 2160
ref = synth.make_ref(W, H, 0); dst = synth.distort(ref, "blockq", 2)
tr, td = torch.from_numpy(ref).cuda(), torch.from_numpy(dst).cuda()
torch.cuda.synchronize()
print("lib", os.environ.get("OAVIF_AMD_LIB", "default"))
for n in (1, 2, 3):
    ctxs = [oavif_amd.Ssimu2(0) for _ in range(n)]
    for pat in ("all", "wait"):
        best = 1e9
        for rep in range(4):
            for c in ctxs:
                c.enqueue_device(tr.data_ptr(), td.data_ptr(), W, H); c.wait()
            K = 200
            t = time.perf_counter()
            for i in range(K):
                c = ctxs[i % n]
                if pat == "wait" and i >= n: c.wait()
                c.enqueue_device(tr.data_ptr(), td.data_ptr(), W, H)
            for c in ctxs: c.wait()
            best = min(best, (time.perf_counter() - t) / K * 1e3)
        print(f"streams {n} {pat:4s}: {best:.4f} ms/score  {W*H/1e6/best*1e3:.0f} MP/s")
    for c in ctxs: c.close()
s = oavif_amd.Ssimu2(0)
print("stages ms", [round(s.time_stage(tr.data_ptr(), td.data_ptr(), W, H, st, 50), 5) for st in (0, 1, 2)])
