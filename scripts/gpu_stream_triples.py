"""4K ms/score for every pair and triple out of 6 contexts (stream overlap vs hardware queues)."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oavif_amd
from oavif_amd import synth
W, H = 3840, 2160
ref = synth.make_ref(W, H, 0); dst = synth.distort(ref, "blockq", 2)
tr, td = torch.from_numpy(ref).cuda(), torch.from_numpy(dst).cuda()
torch.cuda.synchronize()
ctxs = [oavif_amd.Ssimu2(0) for _ in range(6)]
def run(group, K):
    t = time.perf_counter()
    for i in range(K):
        ctxs[group[i % len(group)]].enqueue_device(tr.data_ptr(), td.data_ptr(), W, H)
    for j in group: ctxs[j].wait()
    return (time.perf_counter() - t) / K * 1e3
for j in range(6): run((j,), 20)
run((0, 1), 400)
out = []
for r in (2, 3, 4):
    for g in itertools.combinations(range(6), r):
        run(g, 30)
        out.append((min(run(g, 240) for _ in range(2)), g))
for ms, g in sorted(out)[:12]: print(g, f"{ms:.4f}")
print("...")
for ms, g in sorted(out)[-5:]: print(g, f"{ms:.4f}")
for r in (2, 3, 4):
    v = sorted(ms for ms, g in out if len(g) == r)
    print(f"size {r}: best {v[0]:.4f} median {v[len(v)//2]:.4f} worst {v[-1]:.4f}")
