"""Why does the OpenMP checker's rate depend on where it is pinned?  (14 MP/s on the
idle-core pick of the driver's run against 61 on cpus 0-15.)  Prints what decides it -- cgroup quota,
cpuset, affinity mask, the GPU's NUMA-local cpus, per-cpu busy fractions over 1 s -- and times the checker
at 4K on several core sets, each in a fresh process (affinity is set before OpenMP creates its threads).
    python3 scripts/cpu_pin_diag.py            the table
    python3 scripts/cpu_pin_diag.py --child SET THREADS     one timing (internal)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oavif_amd import hostinfo  # noqa: E402


def child(cpus, threads):
    if cpus:
        os.sched_setaffinity(0, cpus)
    from oavif_amd import synth
    from oracle import ssimu2_oracle as orc
    orc.build()
    n = orc.set_num_threads(threads)
    ref = synth.make_ref(3840, 2160, 0)
    dst = synth.distort(ref, "blockq", 2)
    orc.compute_ssimu2(ref[:256, :256], dst[:256, :256], orc.BLUR_FIR, omp=True)
    ms = []
    for _ in range(6):
        t = time.perf_counter()
        orc.compute_ssimu2(ref, dst, orc.BLUR_FIR, omp=True)
        ms.append((time.perf_counter() - t) * 1e3)
    print(json.dumps({"threads": n, "ms": [round(m, 1) for m in ms], "MPps_median": round(8.2944 / sorted(ms)[len(ms) // 2] * 1e3, 1),
                      "MPps_best": round(8.2944 / min(ms) * 1e3, 1)}))


def run(label, cpus, threads):
    arg = hostinfo.format_cpus(cpus) if cpus else "-"
    r = subprocess.run([sys.executable, __file__, "--child", arg, str(threads)], capture_output=True, text=True)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
    print(f"{label:28s} cpus={arg:40s} {line}", flush=True)


def busy_table(cpus, seconds):
    a = hostinfo.read_cpu_ticks()
    time.sleep(seconds)
    b = hostinfo.read_cpu_ticks()
    out = {}
    for c in cpus:
        if c in a and c in b and b[c][1] > a[c][1]:
            out[c] = (b[c][0] - a[c][0]) / (b[c][1] - a[c][1])
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(hostinfo.parse_cpulist(sys.argv[2]) if sys.argv[2] != "-" else None, int(sys.argv[3]))
        return
    allowed = hostinfo.allowed_cpus()
    quota = hostinfo.cgroup_cpu_quota()
    print("affinity mask:", hostinfo.format_cpus(allowed), f"({len(allowed)} cpus)")
    print("cgroup cpu quota:", quota)
    for f in ("/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpuset.cpus", "/sys/fs/cgroup/cpu.max",
              "/sys/fs/cgroup/cpu.stat", "/proc/loadavg"):
        try:
            print(f, "=", open(f).read().strip().replace("\n", " | "))
        except Exception as e:
            print(f, "unreadable:", type(e).__name__)
    gl = hostinfo.gpu_local_cpulists()
    print("gpu local cpulists:", [hostinfo.format_cpus(g) for g in gl], "visible:", hostinfo.visible_gpu_indices())
    try:
        print("siblings of cpu0:", open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip(),
              " numa nodes:", sorted(os.listdir("/sys/devices/system/node"))[:6])
    except Exception:
        pass
    busy = busy_table(allowed, 1.0)
    hot = sorted(((v, c) for c, v in busy.items() if v > 0.2), reverse=True)
    print(f"busy > 20 % over 1 s: {len(hot)} of {len(busy)} cpus:", hostinfo.format_cpus([c for _, c in hot]))
    n = hostinfo.usable_cores()
    fixed = hostinfo.node_core_sets(1)[0]
    print("node_core_sets(1)[0]:", hostinfo.format_cpus(fixed), "busy:", [round(busy.get(c, -1), 2) for c in fixed])
    pool = hostinfo.sibling_order(allowed)
    idle01 = hostinfo.pick_idle_cpus(pool, n, sample_s=0.1)
    idle1 = hostinfo.pick_idle_cpus(pool, n, sample_s=1.0)
    r03 = hostinfo.parse_cpulist("5,11-12,14-17,20-21,23,25-26,37,40,54,56")
    run("unpinned", None, n)
    run("fixed slice", fixed, n)
    run("first n of mask", pool[:n], n)
    run("idle pick 0.1 s", idle01, n)
    run("idle pick 1 s", idle1, n)
    if all(c in allowed for c in r03):
        run("r03 driver set", r03, n)
    run("fixed slice (again)", fixed, n)
    run("fixed slice, 8 threads", fixed[:8], 8)
    st = open("/sys/fs/cgroup/cpu.stat").read().replace("\n", " | ") if os.path.exists("/sys/fs/cgroup/cpu.stat") else ""
    print("cpu.stat after:", st)


if __name__ == "__main__":
    main()
