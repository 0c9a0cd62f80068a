"""Host-side placement for one-process-per-GPU jobs ("CPU encode/decode for that image
runs on host cores near that GPU").

The CPU codec is 3-4 orders of magnitude slower than the GPU score, so where a rank's encoder
threads run IS the multi-GPU result of a batch.  Everything here is plain Linux (affinity mask,
cgroup quota, sysfs); nothing touches the GPU, so a rank can pin itself before its first HIP call.
"""
from __future__ import annotations

import glob
import os
from typing import Dict, List, Optional, Sequence


def allowed_cpus() -> List[int]:
    try:
        return sorted(os.sched_getaffinity(0))
    except Exception:
        return list(range(os.cpu_count() or 1))


def cgroup_cpu_quota() -> Optional[float]:
    """CPUs' worth of time the cgroup grants this process (cpu.max / cfs quota), None = no limit."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] == "max":
                    return None
                return int(txt[0]) / int(txt[1])
            q = int(txt[0])
            if q <= 0:
                return None
            return q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except Exception:
            continue
    return None


def usable_cores(cap: int = 32) -> int:
    """Host threads this process may really use: the cgroup CPU quota if there is one, else the
    affinity mask, capped (the GPU box gives a 1-GPU job ~16 cores of a 256-thread host;
    oversubscribing OpenMP 16x makes a CPU baseline 5x slower than it is)."""
    n = len(allowed_cpus())
    q = cgroup_cpu_quota()
    if q is not None:
        n = min(n, max(1, int(q + 0.5)))
    return max(1, min(n, cap))


def parse_cpulist(text: str) -> List[int]:
    """"0-3,8,10-11" -> [0, 1, 2, 3, 8, 10, 11] (the format of sysfs cpulist files)."""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def gpu_local_cpulists(sysfs: str = "/sys") -> List[List[int]]:
    """local_cpulist (the cores of the NUMA node a device hangs off) of every AMD display-class PCI
    function, in PCI address order.  [] when sysfs does not say.  (Which of them is HIP device r is NOT this order
    in general: visible_gpus.)"""
    return [c for _p, c in amd_gpu_functions(sysfs)]


def kfd_gpu_nodes(sysfs: str = "/sys", dev: str = "/dev") -> List[dict]:
    """The GPU nodes of the KFD topology, in node order -- the order the ROCm runtime enumerates devices in -- each with
    its PCI address (from `domain` / `location_id`), its DRM render minor and whether THIS process may open that render
    node (a container that was given one GPU of eight sees all eight in sysfs, but only its own /dev/dri/renderD*: the
    runtime skips the others the same way).  Nothing here initialises HIP, so a rank can call it before it pins itself.
    [] when the topology is not there."""
    out: List[dict] = []
    nodes = glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*"))
    try:
        nodes.sort(key=lambda q: int(os.path.basename(q)))
    except ValueError:
        return []
    for nd in nodes:
        pr: Dict[str, str] = {}
        try:
            for ln in open(os.path.join(nd, "properties")):
                k, _, v = ln.strip().partition(" ")
                pr[k] = v
            if int(pr.get("simd_count", "0") or 0) <= 0:
                continue  # a CPU node
            loc, dom = int(pr.get("location_id", "0")), int(pr.get("domain", "0"))
            minor = int(pr.get("drm_render_minor", "-1"))
        except Exception:
            continue
        pci = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
        path = os.path.join(dev, f"dri/renderD{minor}")
        ok = False
        if minor >= 0 and os.path.exists(path):
            try:
                fd = os.open(path, os.O_RDWR)
                os.close(fd)
                ok = True
            except OSError:
                ok = False
        out.append({"node": int(os.path.basename(nd)), "pci": pci, "render_minor": minor, "accessible": ok})
    return out


def _visible_filter(n: int) -> Optional[List[int]]:
    """Indices (into the devices the runtime can open) that ROCR_VISIBLE_DEVICES, then HIP_ / CUDA_VISIBLE_DEVICES
    leave, in the order they give; None when a variable holds something other than plain indices (UUIDs)."""
    idx = list(range(n))
    for group in (("ROCR_VISIBLE_DEVICES",), ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        for var in group:
            v = os.environ.get(var)
            if v:
                try:
                    pick = [int(t) for t in v.split(",") if t.strip() != ""]
                except ValueError:
                    return None
                idx = [idx[k] for k in pick if 0 <= k < len(idx)]
                break
    return idx


def visible_gpus(sysfs: str = "/sys", dev: str = "/dev") -> Optional[List[str]]:
    """PCI addresses of the GPUs this process will see as HIP device 0, 1, ..., found WITHOUT touching the GPU: the GPU
    nodes of the KFD topology whose render node this process may open, in runtime order, then *_VISIBLE_DEVICES.
    None when the KFD topology does not say (callers then fall back to sysfs PCI order + *_VISIBLE_DEVICES).
    Why it matters (round 5, scripts/gpu_topology_probe.py on the pool's boxes): a job that is handed ONE GPU of an
    eight-GPU host runs with ROCR_VISIBLE_DEVICES=0 -- "the first of the GPUs I can open", which was 0000:f1:00.0,
    the eighth in PCI order and on NUMA node 1 -- while sysfs lists all eight; reading that 0 as "GPU 0 of the node"
    pinned the rank to cpus 0-15 on the other socket, and every tenant of the host to the same sixteen."""
    nodes = kfd_gpu_nodes(sysfs, dev)
    acc = [n["pci"] for n in nodes if n["accessible"]]
    if not acc:
        return None
    keep = _visible_filter(len(acc))
    if keep is None:
        return None
    return [acc[k] for k in keep]


def amd_gpu_functions(sysfs: str = "/sys") -> List[tuple]:
    """(PCI address, local_cpulist) of every AMD display-class / accelerator PCI function, in PCI address order."""
    found = []
    for devdir in sorted(glob.glob(os.path.join(sysfs, "bus/pci/devices/*"))):
        try:
            if open(os.path.join(devdir, "vendor")).read().strip() != "0x1002":
                continue
            if not open(os.path.join(devdir, "class")).read().strip().startswith(("0x03", "0x12")):
                continue
            cpus = parse_cpulist(open(os.path.join(devdir, "local_cpulist")).read())
        except Exception:
            continue
        if cpus:
            found.append((os.path.basename(devdir).lower(), cpus))
    return found


def pci_local_cpulist(pci: str, sysfs: str = "/sys") -> List[int]:
    try:
        return parse_cpulist(open(os.path.join(sysfs, "bus/pci/devices", pci, "local_cpulist")).read())
    except Exception:
        return []


def cpu_numa_nodes(cpus: Sequence[int], sysfs: str = "/sys") -> List[int]:
    """The NUMA nodes the given cpus belong to (sorted; [] when sysfs does not say)."""
    want = set(cpus)
    out = []
    for nd in glob.glob(os.path.join(sysfs, "devices/system/node/node[0-9]*")):
        try:
            if want & set(parse_cpulist(open(os.path.join(nd, "cpulist")).read())):
                out.append(int(os.path.basename(nd)[4:]))
        except Exception:
            continue
    return sorted(out)


def sibling_order(cpus: Sequence[int], sysfs: str = "/sys") -> List[int]:
    """`cpus` ordered so that the hardware threads of one physical core are neighbours (key: the
    lowest thread id of the core, from topology/thread_siblings_list): a contiguous slice then
    holds whole cores instead of sharing each core with the slice 128 ids further on."""
    def key(c: int):
        try:
            sib = parse_cpulist(open(os.path.join(sysfs, f"devices/system/cpu/cpu{c}/topology/thread_siblings_list")).read())
            return (min(sib), c)
        except Exception:
            return (c, c)
    return sorted(cpus, key=key)


def core_groups(cpus: Sequence[int], sysfs: str = "/sys") -> Dict[int, int]:
    """cpu id -> id of its physical core (the lowest hardware-thread id of the core); a cpu
    without sysfs topology is its own core."""
    out: Dict[int, int] = {}
    for c in cpus:
        try:
            out[c] = min(parse_cpulist(open(os.path.join(
                sysfs, f"devices/system/cpu/cpu{c}/topology/thread_siblings_list")).read()))
        except Exception:
            out[c] = c
    return out


def one_thread_per_core_first(cpus: Sequence[int], groups: Optional[Dict[int, int]] = None) -> List[int]:
    """`cpus` reordered: the first hardware thread of every physical core, then the second ones, ...
    A slice that a cgroup quota cuts short then keeps whole cores busy instead of pairs of SMT
    siblings (measured on the GPU box, 16-CPU quota, 16 encoder threads: 46.8 images/s on 16
    cores against 40.1 on 8 cores x 2 threads -- and 22 unpinned, see pin_rank)."""
    groups = groups if groups is not None else core_groups(cpus)
    seen: Dict[int, int] = {}
    ranked = []
    for c in cpus:
        g = groups.get(c, c)
        k = seen.get(g, 0)
        seen[g] = k + 1
        ranked.append((k, c))
    return [c for _, c in sorted(ranked, key=lambda t: t[0])]   # stable: keeps the order inside a round


def visible_gpu_indices() -> Optional[List[int]]:
    """Indices (in the node's enumeration) of the GPUs the runtime will show this process, from
    ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when they are plain numbers;
    None = no restriction known."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                return [int(t) for t in v.split(",") if t.strip() != ""]
            except ValueError:
                return None
    return None


def rank_core_sets(local_world: int, cpus: Optional[Sequence[int]] = None,
                   gpu_cpulists: Optional[Sequence[Sequence[int]]] = None,
                   quota: Optional[float] = None,
                   core_of: Optional[Dict[int, int]] = None) -> List[List[int]]:
    """Disjoint host core sets for the `local_world` ranks of one node, rank r <-> GPU r.

    Ranks whose GPUs share a NUMA node split that node's allowed cores into contiguous slices
    (in the order given: sibling_order keeps the hardware threads of a core together);
    without usable sysfs topology the whole allowed set is split the same way.  With a cgroup
    quota below the allowed core count every slice is cut to its share of the quota, so the ranks
    together never run more threads than the cgroup will schedule.  Deterministic: every rank
    computes the same table."""
    cpus = list(cpus) if cpus is not None else sibling_order(allowed_cpus())
    local_world = max(1, int(local_world))
    groups: Dict[tuple, List[int]] = {}
    if gpu_cpulists is not None and len(gpu_cpulists) >= local_world:
        allowed = set(cpus)
        for r in range(local_world):
            nearset = set(gpu_cpulists[r])
            near = tuple(c for c in cpus if c in nearset and c in allowed)
            if not near:
                groups = {}
                break
            groups.setdefault(near, []).append(r)
    if not groups:
        groups = {tuple(cpus): list(range(local_world))}
    sets: List[List[int]] = [[] for _ in range(local_world)]
    for near, ranks in groups.items():
        m = len(ranks)
        per = max(1, len(near) // m)
        for k, r in enumerate(ranks):
            lo = min(k * per, max(len(near) - 1, 0))
            sets[r] = list(near[lo: lo + per]) or [near[-1]]
    if quota is not None:
        share = max(1, int(quota / local_world + 0.5))
        sets = [one_thread_per_core_first(s, core_of)[:share] if len(s) > share else s for s in sets]
    return sets


def gpu_slices(cpus: Sequence[int], gpu_cpulists: Sequence[Sequence[int]]) -> List[Optional[List[int]]]:
    """For every GPU of the node: its slice of the allowed cores `cpus` (in their order) -- the GPUs whose NUMA-local cores
    meet the allowed set share that intersection in contiguous slices, as if every one of them ran a rank -- or None for a
    GPU none of whose local cores this process may use (a cpuset that covers another socket only).  One GPU outside the
    mask does not change the others' slices (it used to drop the whole table to an even split of the allowed
    cores over all GPUs of the host, which pinned a one-GPU tenant with 16 allowed cores to 2)."""
    near_of = []
    for lst in gpu_cpulists:
        ns = set(lst)
        near_of.append(tuple(c for c in cpus if c in ns))
    groups: Dict[tuple, List[int]] = {}
    for g, near in enumerate(near_of):
        if near:
            groups.setdefault(near, []).append(g)
    out: List[Optional[List[int]]] = [None] * len(near_of)
    for near, gs in groups.items():
        per = max(1, len(near) // len(gs))
        for k, g in enumerate(gs):
            lo = min(k * per, max(len(near) - 1, 0))
            out[g] = list(near[lo: lo + per]) or [near[-1]]
    return out


def node_core_sets(local_world: int, procs_per_gpu: int = 1, sysfs: str = "/sys", dev: str = "/dev") -> List[List[int]]:
    """rank_core_sets for this node as it is: sysfs topology, cgroup quota, and the GPUs this job really has.
    Every GPU of the node owns a slice of the cores of its NUMA node (the slice it would get if every GPU of the node
    ran one rank); local rank r sits on HIP device r // procs_per_gpu and takes that GPU's slice (ranks that share a
    GPU split it), cut to the rank's share of the cgroup quota.  So the ranks of a full-node job are near their GPUs,
    and tenants that were each handed some GPUs of a shared host pin themselves to different cores.
    Which physical GPU is HIP device r comes from the KFD topology (visible_gpus: runtime order, render nodes this
    process may open, then *_VISIBLE_DEVICES); without it, sysfs PCI order + *_VISIBLE_DEVICES as before round 5."""
    cpus = sibling_order(allowed_cpus(), sysfs)
    quota = cgroup_cpu_quota()
    ppg = max(1, procs_per_gpu)
    local_world = max(1, int(local_world))
    allg = amd_gpu_functions(sysfs)
    mine_pci = visible_gpus(sysfs, dev)
    if allg and mine_pci:
        index = {pci: k for k, (pci, _c) in enumerate(allg)}
        need = -(-local_world // ppg)                      # GPUs this job's ranks ask for
        have = min(need, len(mine_pci))                    # ... and really have: with fewer (a gloo rehearsal on a smaller box)
        if have >= 1 and all(p_ in index for p_ in mine_pci[:have]):   # rank r sits on device (r // ppg) % have, as the entry points place it
            node = gpu_slices(cpus, [c for _p, c in allg])
            slices = [node[index[mine_pci[g]]] for g in range(have)]
            if any(sl is None for sl in slices):
                # one of THIS job's GPUs has no local core inside the affinity mask: nothing is near it, so the allowed
                # cores are split among the GPUs the job really has, not among every GPU of the host
                slices = rank_core_sets(have, cpus=cpus)
            on_gpu: List[List[int]] = [[] for _ in range(have)]
            for r in range(local_world):
                on_gpu[(r // ppg) % have].append(r)
            sets: List[List[int]] = [[] for _ in range(local_world)]
            for g in range(have):
                k = len(on_gpu[g])
                parts = rank_core_sets(k, cpus=slices[g]) if k > 1 else [list(slices[g])]
                for r, part in zip(on_gpu[g], parts):
                    sets[r] = part
            if all(sets):
                if quota is not None:
                    share = max(1, int(quota / local_world + 0.5))
                    core_of = core_groups([c for x in sets for c in x], sysfs)
                    sets = [one_thread_per_core_first(x, core_of)[:share] if len(x) > share else x for x in sets]
                return sets
    gpus = [c for _p, c in allg] or None
    if gpus and ppg > 1:
        gpus = [g for g in gpus for _ in range(ppg)]
    vis = visible_gpu_indices()
    ngpu = len(gpus) // ppg if gpus else 0
    if gpus and vis and len(vis) < ngpu and all(0 <= v < ngpu for v in vis):
        node = rank_core_sets(ngpu, cpus=cpus, gpu_cpulists=gpus[::ppg])
        mine = [c for v in vis for c in node[v]]
        if mine:
            return rank_core_sets(local_world, cpus=mine, quota=quota)
    return rank_core_sets(local_world, cpus=cpus, gpu_cpulists=gpus, quota=quota)


def read_cpu_ticks(path: str = "/proc/stat") -> Dict[int, tuple]:
    """cpu id -> (busy ticks, total ticks) from /proc/stat ({} when it cannot be read)."""
    out: Dict[int, tuple] = {}
    try:
        for line in open(path):
            if line.startswith("cpu") and line[3:4].isdigit():
                f = line.split()
                v = [int(x) for x in f[1:9]]
                idle = v[3] + v[4]              # idle + iowait
                out[int(f[0][3:])] = (sum(v) - idle, sum(v))
    except Exception:
        return {}
    return out


def pick_idle_cpus(cpus: Sequence[int], n: int, sample_s: float = 1.0, core_of: Optional[Dict[int, int]] = None,
                   sampler=read_cpu_ticks, sleep=None) -> List[int]:
    """The `n` least busy of `cpus` over a sample of /proc/stat, whole physical cores first (a core
    counts as busy as its busiest hardware thread).  OPT-IN only (pin_rank(idle=True) /
    OAVIF_PIN=idle), for a host one shares with other tenants whose threads sit on the fixed slice.
    It is not the default: round 3 made it one with a 0.1 s sample (10 scheduler ticks) and the
    driver's run then measured the OpenMP checker at 14 MP/s on the picked set against 55-61 on the
    fixed slice -- a scattered set costs locality (GPU box, r04: 43 MP/s on sixteen idle cores spread
    over two NUMA nodes, 51 on sixteen scattered over one, 55 on cores 0-15: profiles/r04_pin_diag.log),
    a short sample reads another tenant's momentarily blocked cores as idle, and a pinned thread cannot
    leave a core that turns out busy.  Falls back to the first n of one_thread_per_core_first(cpus)
    when /proc/stat does not help."""
    import time as _time
    cpus = list(cpus)
    order = one_thread_per_core_first(cpus, core_of)
    if n >= len(cpus):
        return order
    a = sampler()
    (sleep or _time.sleep)(sample_s)
    b = sampler()
    if not a or not b or any(c not in a or c not in b for c in cpus):
        return order[:n]
    core_of = core_of if core_of is not None else core_groups(cpus)
    busy = {}
    for c in cpus:
        dt = b[c][1] - a[c][1]
        busy[c] = (b[c][0] - a[c][0]) / dt if dt > 0 else 0.0
    core_busy: Dict[int, float] = {}
    for c in cpus:
        g = core_of.get(c, c)
        core_busy[g] = max(core_busy.get(g, 0.0), busy[c])
    rank = {c: i for i, c in enumerate(order)}
    nth: Dict[int, int] = {}                     # 0 for the first hardware thread of its core seen, 1 for the second, ...
    seen: Dict[int, int] = {}
    for c in cpus:
        g = core_of.get(c, c)
        nth[c] = seen.get(g, 0)
        seen[g] = nth[c] + 1
    # one thread of every core before any second thread; among those the idlest cores (in steps of 10 %:
    # cores that are about equally idle stay in the given order, i.e. contiguous); ties in the given order
    chosen = sorted(cpus, key=lambda c: (nth[c], round(core_busy[core_of.get(c, c)], 1), rank[c]))[:n]
    return sorted(chosen, key=lambda c: rank[c])


def busy_fractions(cpus: Sequence[int], sample_s: float = 1.0, sampler=read_cpu_ticks, sleep=None) -> Dict[int, float]:
    """cpu id -> fraction of `sample_s` it was not idle ({} when /proc/stat cannot be read)."""
    import time as _time
    a = sampler()
    (sleep or _time.sleep)(sample_s)
    b = sampler()
    out: Dict[int, float] = {}
    for c in cpus:
        if c in a and c in b and b[c][1] > a[c][1]:
            out[c] = (b[c][0] - a[c][0]) / (b[c][1] - a[c][1])
    return out


def gpu_near_pool(procs_per_gpu: int = 1) -> List[int]:
    """The allowed cpus of the NUMA node(s) the visible GPU(s) hang off, hardware threads of a core
    neighbours; the whole allowed set when sysfs, the KFD topology and *_VISIBLE_DEVICES do not say."""
    allowed = sibling_order(allowed_cpus())
    allg = amd_gpu_functions()
    mine_pci = visible_gpus()
    if allg and mine_pci:
        by = dict(allg)
        nearset = set(c for p_ in mine_pci for c in by.get(p_, []))
        near = [c for c in allowed if c in nearset]
        if near:
            return near
    gpus = [c for _p, c in allg]
    vis = visible_gpu_indices()
    if gpus and vis and all(0 <= v < len(gpus) for v in vis):
        nearset = set(c for v in vis for c in gpus[v])
        near = [c for c in allowed if c in nearset]
        if near:
            return near
    if gpus:
        nearset = set(gpus[0])
        near = [c for c in allowed if c in nearset]
        if near:
            return near
    return allowed


def candidate_core_sets(n: int, max_sets: int = 3) -> List[List[int]]:
    """Contiguous slices of `n` whole cores near the GPU, the rank's fixed slice (what pin_rank
    takes) first: what bench.py's cpu_baseline times briefly before it keeps the fastest, so that
    another tenant's threads on one slice do not become this repo's "CPU baseline"."""
    first = node_core_sets(1)[0]
    pool = one_thread_per_core_first(gpu_near_pool())
    sets = [list(first)]
    taken = set(first)
    rest = [c for c in pool if c not in taken]
    while len(sets) < max_sets and len(rest) >= n:
        sets.append(rest[:n])
        rest = rest[n:]
    return sets


class PinResult(list):
    """The core set of a rank (a list of cpu ids) + whether the kernel accepted it."""
    pinned: bool = True
    error: str = ""
    how: str = "fixed slice"


def pin_rank(local_rank: int, local_world: int, procs_per_gpu: int = 1, idle: Optional[bool] = None) -> PinResult:
    """Restrict this process (and every thread it starts later) to its rank's core set.  Call it
    before the first GPU call and before any thread pool exists.  Returns the set as a PinResult:
    `.pinned` is False (and `.error` says why) when sched_setaffinity refused -- the rank then
    runs wherever the launcher put it and the caller's summary must say so.

    Every rank takes its deterministic slice of the cores near its GPU (node_core_sets): contiguous
    whole cores, one hardware thread per core first, cut to the rank's share of the cgroup quota.
    Worth doing for a single rank too when the quota is far below the affinity mask: 16 encoder
    threads floating over the 256 CPUs of the GPU host under a 16-CPU quota ran 22 images/s, pinned
    to 16 cores 45 (a quota is enforced by throttling, which a pinned job never hits).
    `idle=True` (or OAVIF_PIN=idle), single rank only: the idlest cores near the GPU over a 1 s
    sample instead -- opt-in, see pick_idle_cpus for why it is not the default."""
    if idle is None:
        idle = os.environ.get("OAVIF_PIN", "") == "idle"
    quota = cgroup_cpu_quota()
    sets = node_core_sets(local_world, procs_per_gpu)
    mine = PinResult(sets[local_rank % len(sets)])
    if idle and local_world == 1 and quota is not None and quota < len(allowed_cpus()):
        n = max(1, int(quota + 0.5))
        picked = pick_idle_cpus(gpu_near_pool(procs_per_gpu), n, sample_s=1.0)
        if picked:
            mine = PinResult(picked)
            mine.how = "idlest cores near the GPU over 1 s"
    try:
        os.sched_setaffinity(0, list(mine))
    except Exception as e:   # reported, not fatal
        mine.pinned = False
        mine.error = f"{type(e).__name__}: {e}"
    return mine


def format_cpus(cpus: Sequence[int]) -> str:
    """[0, 1, 2, 3, 8] -> "0-3,8"."""
    cpus = sorted(cpus)
    parts, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(parts)
