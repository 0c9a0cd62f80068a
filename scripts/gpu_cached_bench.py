"""Reference-cached 4K pass (inputs resident): ms per score on one stream, and what
OAVIF_AMD_NO_REF_BLUR=1 (reference blur planes not cached) changes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oavif_amd
from oavif_amd import synth
W, H = 3840, 2160
ref = synth.make_ref(W, H, 0); dst = synth.distort(ref, "blockq", 2)
tr, td = torch.from_numpy(ref).cuda(), torch.from_numpy(dst).cuda()
torch.cuda.synchronize()
with oavif_amd.Ssimu2(0) as s:
    s.set_reference_device(tr.data_ptr(), W, H)
    for _ in range(300):
        s.enqueue_against_reference_device(td.data_ptr())
    sc = s.wait()
    best = 1e9
    for rep in range(5):
        t = time.perf_counter()
        for _ in range(200):
            s.enqueue_against_reference_device(td.data_ptr())
        sc = s.wait()
        best = min(best, (time.perf_counter() - t) / 200 * 1e3)
    print(f"NO_REF_BLUR={os.environ.get('OAVIF_AMD_NO_REF_BLUR', '-')}: {best:.4f} ms per cached pass, score {sc!r}")
