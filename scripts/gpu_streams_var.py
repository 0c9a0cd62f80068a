"""Run-to-run spread of the two-stream 4K throughput (enqueue-all pattern), 10 repetitions."""
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
for n in (2,):
    ctxs = [oavif_amd.Ssimu2(0) for _ in range(n)]
    for K in (200, 1000):
        res = []
        for rep in range(10):
            for c in ctxs:
                c.enqueue_device(tr.data_ptr(), td.data_ptr(), W, H); c.wait()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(K):
                ctxs[i % n].enqueue_device(tr.data_ptr(), td.data_ptr(), W, H)
            for c in ctxs: c.wait()
            res.append((time.perf_counter() - t) / K * 1e3)
        print(f"K={K}", " ".join(f"{r:.4f}" for r in res))
