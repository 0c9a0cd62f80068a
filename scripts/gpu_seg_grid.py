"""Two-stream 4K throughput for a grid of segment lengths (OAVIF_AMD_SEG_ROWS x _TAIL), one
process, interleaved repetitions so that box and clock drift hit every cell alike."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oavif_amd
from oavif_amd import synth

W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
ref = synth.make_ref(W, H, 0); dst = synth.distort(ref, "blockq", 2)
tr, td = torch.from_numpy(ref).cuda(), torch.from_numpy(dst).cuda()
torch.cuda.synchronize()
segs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "135,150").split(",")]
tails = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "48,64,80").split(",")]
cells = {}
for s0 in segs:
    for t in tails:
        os.environ["OAVIF_AMD_SEG_ROWS"] = str(s0)
        os.environ["OAVIF_AMD_SEG_ROWS_TAIL"] = str(t)
        cells[(s0, t)] = [oavif_amd.Ssimu2(0) for _ in range(2)]
res = {k: [] for k in cells}
one = {k: [] for k in cells}
def run(ctxs, K, n=2):
    t = time.perf_counter()
    for i in range(K):
        ctxs[i % n].enqueue_device(tr.data_ptr(), td.data_ptr(), W, H)
    for c in ctxs[:n]: c.wait()
    return (time.perf_counter() - t) / K * 1e3
run(next(iter(cells.values())), 400)  # clocks
for rep in range(6):
    for k, ctxs in cells.items():
        run(ctxs, 50)
        res[k].append(run(ctxs, 400))
        one[k].append(run(ctxs, 200, 1))
for k, v in sorted(res.items(), key=lambda kv: min(kv[1])):
    print(f"seg0 {k[0]:3d} tail {k[1]:3d}: two streams min {min(v):.4f} median {sorted(v)[len(v)//2]:.4f} ms/score "
          f"({W*H/1e6/min(v)*1e3:.0f} MP/s); one stream {min(one[k]):.4f}")
