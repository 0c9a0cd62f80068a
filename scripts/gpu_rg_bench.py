"""The published-recursion blur modes on the GPU box: ms per pair score, per ssimu2_set_reference and
per reference-cached pass (rotating over NP distorted frames: HBM-fed), one stream.
    python3 scripts/gpu_rg_bench.py [w h [NP [mode]]]      mode: recursive | recursive_fma
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (k_rg_xyb / k_rg_h / k_rg_v)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import oavif_amd  # noqa: E402
from oavif_amd import _lib, synth  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 8
mode = {"recursive": _lib.BLUR_RECURSIVE, "recursive_fma": _lib.BLUR_RECURSIVE_FMA}[
    sys.argv[4] if len(sys.argv) > 4 else "recursive"]
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

s = oavif_amd.Ssimu2(0, instrumented=bool(os.environ.get('OAVIF_RG_INSTR')))
fir = s.score_device(pr[0], pd[0], w, h)
s.set_blur(mode)
score = s.score_device(pr[0], pd[0], w, h)
for i in range(8):  # clocks
    s.enqueue_device(pr[i % NP], pd[i % NP], w, h)
s.wait()


def timed(n, fn, wait):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    if wait:
        s.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


pair_s = min(timed(16, lambda i: s.enqueue_device(pr[i % NP], pd[i % NP], w, h), True) for _ in range(3))
setref_s = min(timed(8, lambda i: s.set_reference_device(pr[i % NP], w, h), False) for _ in range(3))
s.set_reference_device(pr[0], w, h)
pass_s = min(timed(32, lambda i: s.enqueue_against_reference_device(pd[i % NP]), True) for _ in range(3))
s.enqueue_against_reference_device(pd[0])
cached = s.wait()
free, total = torch.cuda.mem_get_info()
print(f"rg_bench {w}x{h} mode={mode}: score={score:.9f} cached_pass_score={cached:.9f} (same bits: {score == cached}) "
      f"fir={fir:.9f}")
print(f"rg_bench: pair {pair_s * 1e3:.3f} ms  set_reference {setref_s * 1e3:.3f} ms  cached pass {pass_s * 1e3:.3f} ms "
      f"({w * h / 1e6 / pass_s:.0f} MP/s)  device memory in use {(total - free) / 2**30:.2f} GiB")
s.close()
