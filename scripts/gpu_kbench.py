"""Kernel micro-bench on the GPU box, the workload the rocprofv3 PMC passes run over
(scripts/gpu_pmc_sets.sh): the marching kernel and whole scores at 4K ROTATING over 8 distinct
pairs (inputs from HBM, not from the Infinity Cache), then reference-cached passes rotating over
the 8 distorted frames.  Uses the instrumented build (stage timing hooks)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import oavif_amd  # noqa: E402
from oavif_amd import synth  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 8   # distinct pairs in rotation
ref = synth.make_ref(w, h, 0)
dst = synth.distort(ref, "blockq", 2)
tr = torch.from_numpy(ref).cuda().contiguous()
td = torch.from_numpy(dst).cuda().contiguous()
pairs = [(tr, td)]
for k in range(1, NP):
    a, b = torch.roll(tr, k * w // NP, 1), torch.roll(td, k * w // NP, 1)
    if k & 1:
        a, b = a.flip(0), b.flip(0)
    pairs.append((a.contiguous(), b.contiguous()))
torch.cuda.synchronize()
pr, pd = [a.data_ptr() for a, _ in pairs], [b.data_ptr() for _, b in pairs]
s = oavif_amd.Ssimu2(0, instrumented=True)
score = s.score_device(pr[0], pd[0], w, h)
for _ in range(100):  # clocks
    s.enqueue_device(pr[0], pd[0], w, h)
s.wait()
k_rot = min(s.time_march_rotating(pr, pd, w, h, 8 * NP) for _ in range(3)) * 1e3
ks = [s.time_stage(pr[0], pd[0], w, h, st, 50) * 1e3 for st in range(3)]
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 64
for i in range(n):
    s.enqueue_device(pr[i % NP], pd[i % NP], w, h)
s.wait()
dt = (time.perf_counter() - t0) / n
print(f"kbench: score={score:.9f} march_us rotating={k_rot:.1f} cache_resident={ks[1]:.1f} "
      f"pyramid_us={ks[0]:.1f} finalize_us={ks[2]:.1f} whole_score_us(rotating)={dt * 1e6:.1f} "
      f"MP/s={w * h / 1e6 / dt:.0f}")
s.set_reference_device(pr[0], w, h)
for i in range(8):
    s.enqueue_against_reference_device(pd[i % NP])
s.wait()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    s.enqueue_against_reference_device(pd[i % NP])
s.wait()
dt = (time.perf_counter() - t0) / n
print(f"cached-reference pass (rotating dist frames): {dt * 1e6:.1f} us/score  {w * h / 1e6 / dt:.0f} MP/s")
# the published-recursion mode (ssimu2_ctx_set_blur): latency-bound by construction
s.set_blur(oavif_amd._lib.BLUR_RECURSIVE)
rscore = s.score_device(pr[0], pd[0], w, h)
for i in range(4):
    s.enqueue_device(pr[i % NP], pd[i % NP], w, h)
s.wait()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 16
for i in range(n):
    s.enqueue_device(pr[i % NP], pd[i % NP], w, h)
s.wait()
dt = (time.perf_counter() - t0) / n
print(f"recursive blur mode: score={rscore:.9f} (default mode {score:.9f})  {dt * 1e6:.1f} us/score  "
      f"{w * h / 1e6 / dt:.0f} MP/s")
s.close()
