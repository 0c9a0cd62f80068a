"""MP/s of the CPU checker (OpenMP build) at 4K on this host, unpinned or pinned to the job's core set
(oavif_amd.hostinfo): python3 scripts/cpu_oracle_rate.py [pin].  Timing aid for bench.py's cpu_baseline."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oavif_amd import hostinfo, synth  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "pin":
    print("pinned to", hostinfo.format_cpus(hostinfo.pin_rank(0, 1, idle=(len(sys.argv) > 2 and sys.argv[2] == "idle"))))
from oracle import ssimu2_oracle as orc  # noqa: E402

orc.build()
n = orc.set_num_threads(hostinfo.usable_cores())
ref = synth.make_ref(3840, 2160, 0)
dst = synth.distort(ref, "blockq", 2)
orc.compute_ssimu2(ref[:256, :256], dst[:256, :256], orc.BLUR_FIR, omp=True)
t = time.perf_counter()
reps = 10
for _ in range(reps):
    orc.compute_ssimu2(ref, dst, orc.BLUR_FIR, omp=True)
dt = (time.perf_counter() - t) / reps
print(f"{n} threads: {3840 * 2160 / 1e6 / dt:.1f} MP/s")
