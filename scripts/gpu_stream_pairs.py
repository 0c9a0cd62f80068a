"""Which pairs of scorer contexts (HIP streams, in creation order) overlap their scores?
Two-stream 4K ms/score for several pairs out of 8 contexts of one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oavif_amd
from oavif_amd import synth
W, H = 3840, 2160
ref = synth.make_ref(W, H, 0); dst = synth.distort(ref, "blockq", 2)
tr, td = torch.from_numpy(ref).cuda(), torch.from_numpy(dst).cuda()
torch.cuda.synchronize()
ctxs = [oavif_amd.Ssimu2(0) for _ in range(8)]
def run(pair, K):
    t = time.perf_counter()
    for i in range(K):
        ctxs[pair[i % 2]].enqueue_device(tr.data_ptr(), td.data_ptr(), W, H)
    for j in pair: ctxs[j].wait()
    return (time.perf_counter() - t) / K * 1e3
run((0, 1), 400)
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
for pair in ((0, 1), (2, 3), (4, 5), (6, 7), (0, 2), (1, 3), (0, 4), (3, 7), (2, 6), (1, 2), (5, 6)):
    run(pair, 50)
    print(pair, f"{min(run(pair, 300) for _ in range(3)):.4f} ms/score")
