"""Multi-rank path on CPU: world_size 2 over gloo must produce the same gathered records as a
single process (images dealt largest-first over the ranks, one all_gather of fixed-size records,
ranks pinned to disjoint host core sets: SURVEY.md 8e)."""
import csv
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_dist_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def image_dir(tmp_path_factory, hip_lib):
    from PIL import Image
    d = tmp_path_factory.mktemp("imgs")
    rng = np.random.default_rng(0)
    for name in ["a.png", "b.png", "c.jpg", "d.jpeg", "e.png", "bad_one.png", "f.PNG", "skip.txt", "g.png"]:
        if name.endswith(".txt"):
            (d / name).write_text("not an image")
            continue
        img = Image.fromarray(rng.integers(0, 256, (24, 32, 3), dtype=np.uint8))
        img.save(d / name, format="JPEG" if name.lower().endswith(("jpg", "jpeg")) else "PNG")
    return d


def _run(world, image_dir, out, workers=1):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OAVIF_AMD_NO_TORCH="0",
                   BATCH_WORKERS=str(workers))
        procs.append(subprocess.Popen([sys.executable, WORKER, str(image_dir), str(out)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return json.load(open(out)), outs[0]


def test_shard_rule():
    from oavif_amd import batch
    assert batch.shard(10, 0, 4) == [0, 4, 8]
    assert batch.shard(10, 3, 4) == [3, 7]
    assert sorted(sum((batch.shard(257, r, 8) for r in range(8)), [])) == list(range(257))
    assert batch.shard(3, 5, 8) == []


def test_largest_first_dealing():
    """"sort/largest-first if sizes vary".  Deterministic, a partition, i mod N for
    equal sizes, and byte loads within 10 % of each other on a spread of sizes."""
    from oavif_amd import batch
    assert batch.deal_largest_first([7] * 10, 4) == [[0, 4, 8], [1, 5, 9], [2, 6], [3, 7]]
    rng = np.random.default_rng(3)
    sizes = [int(x) for x in rng.integers(200_000, 9_000_000, 256)]
    deal = batch.deal_largest_first(sizes, 8)
    assert sorted(sum(deal, [])) == list(range(256)) and deal == batch.deal_largest_first(sizes, 8)
    loads = [sum(sizes[i] for i in d) for d in deal]
    assert max(loads) <= 1.10 * min(loads), loads
    assert all(sizes[a] >= sizes[b] for d in deal for a, b in zip(d, d[1:]))   # each rank: largest first
    mod = [sum(sizes[i] for i in range(r, 256, 8)) for r in range(8)]
    assert max(loads) - min(loads) < max(mod) - min(mod)                        # better than i mod N
    assert batch.deal_largest_first([5, 1], 4) == [[0], [1], [], []]


def test_rank_core_sets_are_disjoint_near_their_gpu_and_within_the_quota():
    from oavif_amd import hostinfo
    sets = hostinfo.rank_core_sets(8, cpus=list(range(256)))
    assert [len(s) for s in sets] == [32] * 8 and len(set(sum(sets, []))) == 256
    # two sockets, four GPUs each; cores 0-63 + their SMT siblings 128-191 on socket 0
    s0 = list(range(0, 64)) + list(range(128, 192))
    s1 = list(range(64, 128)) + list(range(192, 256))
    sets = hostinfo.rank_core_sets(8, cpus=list(range(256)), gpu_cpulists=[s0] * 4 + [s1] * 4)
    assert all(set(sets[r]) <= set(s0) for r in range(4)) and all(set(sets[r]) <= set(s1) for r in range(4, 8))
    assert len(set(sum(sets, []))) == sum(len(s) for s in sets) == 256
    # a cgroup that grants 16 CPUs to a job of two ranks: eight threads each, still disjoint --
    # and eight different physical cores each, not four cores and their SMT siblings
    sib = [x for c in range(128) for x in (c, c + 128)]            # sibling_order of such a host
    core_of = {c: c % 128 for c in range(256)}
    sets = hostinfo.rank_core_sets(2, cpus=sib, quota=16.0, core_of=core_of)
    assert sets == [list(range(0, 8)), list(range(64, 72))]
    assert hostinfo.one_thread_per_core_first([0, 128, 1, 129, 2], core_of) == [0, 1, 2, 128, 129]
    sets = hostinfo.rank_core_sets(2, cpus=list(range(256)), quota=16.0)
    assert [len(s) for s in sets] == [8, 8] and not set(sets[0]) & set(sets[1])
    assert hostinfo.rank_core_sets(3, cpus=[4, 5]) == [[4], [5], [5]]      # more ranks than cores: still a set each
    # --procs-per-gpu 2 on that node: sixteen ranks, rank r on GPU r // 2, each GPU's slice split in two
    sets = hostinfo.rank_core_sets(16, cpus=list(range(256)), gpu_cpulists=[g for g in [s0] * 4 + [s1] * 4 for _ in range(2)])
    assert [len(s) for s in sets] == [16] * 16 and len(set(sum(sets, []))) == 256
    assert all(set(sets[r]) <= set(s0) for r in range(8)) and all(set(sets[r]) <= set(s1) for r in range(8, 16))
    assert hostinfo.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert hostinfo.format_cpus([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"
    assert 1 <= hostinfo.usable_cores() <= 32
    # a tenant that sees GPU 5 of eight takes the cores GPU 5 would get on a fully used node
    import os
    old = os.environ.get("ROCR_VISIBLE_DEVICES")
    os.environ["ROCR_VISIBLE_DEVICES"] = "5"
    try:
        assert hostinfo.visible_gpu_indices() == [5]
    finally:
        if old is None:
            del os.environ["ROCR_VISIBLE_DEVICES"]
        else:
            os.environ["ROCR_VISIBLE_DEVICES"] = old


def test_gpu_numa_topology_is_read_from_sysfs(tmp_path):
    """gpu_local_cpulists: AMD display / accelerator PCI functions only, in PCI address order;
    sibling_order / core_groups from cpuN/topology/thread_siblings_list."""
    from oavif_amd import hostinfo
    devs = tmp_path / "bus" / "pci" / "devices"
    spec = [("0000:05:00.0", "0x1002", "0x120000", "0-3,8-11"),      # MI-class accelerator, socket 0
            ("0000:85:00.0", "0x1002", "0x120000", "4-7,12-15"),     # ... socket 1
            ("0000:03:00.0", "0x1a03", "0x030000", "0-3,8-11"),      # the BMC's VGA: not AMD
            ("0000:05:00.1", "0x1002", "0x040300", "0-3,8-11"),      # an AMD audio function: not a GPU
            ("0000:45:00.0", "0x1002", "0x038000", "0-3,8-11")]      # display controller, socket 0
    for addr, vendor, cls, cpus in spec:
        d = devs / addr
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n")
        (d / "class").write_text(cls + "\n")
        (d / "local_cpulist").write_text(cpus + "\n")
    got = hostinfo.gpu_local_cpulists(str(tmp_path))
    assert got == [[0, 1, 2, 3, 8, 9, 10, 11], [0, 1, 2, 3, 8, 9, 10, 11], [4, 5, 6, 7, 12, 13, 14, 15]]
    for c in range(16):
        t = tmp_path / "devices" / "system" / "cpu" / f"cpu{c}" / "topology"
        t.mkdir(parents=True)
        (t / "thread_siblings_list").write_text(f"{c % 8},{c % 8 + 8}\n")
    assert hostinfo.sibling_order(range(16), str(tmp_path)) == [0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6, 14, 7, 15]
    assert hostinfo.core_groups([3, 11, 12], str(tmp_path)) == {3: 3, 11: 3, 12: 4}
    sets = hostinfo.rank_core_sets(3, cpus=hostinfo.sibling_order(range(16), str(tmp_path)), gpu_cpulists=got,
                                   quota=6.0, core_of=hostinfo.core_groups(range(16), str(tmp_path)))
    # GPUs 0 and 1 share socket 0 (cores 0-3 + siblings): two cores each, one thread per core under the quota
    assert sets == [[0, 1], [2, 3], [4, 5]]


def _fake_host(tmp_path, kfd_gpus, accessible):
    """An eight-GPU, two-socket host as this pool's boxes show it in sysfs: 4 GPUs per NUMA node (node 0: cpus 0-63 +
    SMT siblings 128-191, node 1: 64-127 + 192-255).  `kfd_gpus`: PCI addresses of the KFD topology's GPU nodes in node
    order; `accessible`: those whose /dev/dri/renderD* this process can open."""
    pcis = ["0000:0a:00.0", "0000:23:00.0", "0000:5a:00.0", "0000:72:00.0", "0000:8b:00.0", "0000:a4:00.0", "0000:d9:00.0", "0000:f1:00.0"]
    sysfs, dev = tmp_path / "sys", tmp_path / "dev"
    for k, addr in enumerate(pcis):
        d = sysfs / "bus" / "pci" / "devices" / addr
        d.mkdir(parents=True)
        (d / "vendor").write_text("0x1002\n")
        (d / "class").write_text("0x120000\n")
        (d / "local_cpulist").write_text("0-63,128-191\n" if k < 4 else "64-127,192-255\n")
        (d / "numa_node").write_text("0\n" if k < 4 else "1\n")
    for c in range(256):
        t = sysfs / "devices" / "system" / "cpu" / f"cpu{c}" / "topology"
        t.mkdir(parents=True)
        (t / "thread_siblings_list").write_text(f"{c % 128},{c % 128 + 128}\n")
    for n, cl in ((0, "0-63,128-191"), (1, "64-127,192-255")):
        nd = sysfs / "devices" / "system" / "node" / f"node{n}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cl + "\n")
    nodes = sysfs / "class" / "kfd" / "kfd" / "topology" / "nodes"
    for n in (0, 1):                                                   # the two CPU nodes
        (nodes / str(n)).mkdir(parents=True)
        (nodes / str(n) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\ndrm_render_minor 0\n")
    (dev / "dri").mkdir(parents=True)
    for k, addr in enumerate(kfd_gpus):
        dom, bus, rest = addr.split(":")
        devn, fn = rest.split(".")
        loc = (int(bus, 16) << 8) | (int(devn, 16) << 3) | int(fn)
        nd = nodes / str(2 + k)
        nd.mkdir(parents=True)
        (nd / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {loc}\ndomain {int(dom, 16)}\ndrm_render_minor {128 + 8 * k}\n")
        if addr in accessible:
            (dev / "dri" / f"renderD{128 + 8 * k}").write_bytes(b"")
    return str(sysfs), str(dev)


def test_the_rank_pins_near_the_gpu_it_really_has(tmp_path, monkeypatch):
    """Round 5 (scripts/gpu_topology_probe.py on the pool's boxes): a job handed ONE GPU of an eight-GPU host runs with
    ROCR_VISIBLE_DEVICES=0 = "the first GPU I can open" -- 0000:f1:00.0 there, the eighth in PCI order, on NUMA node 1.
    The KFD topology (GPU nodes whose render node this process can open, in runtime order) says which GPU that is; the
    rank then takes THAT GPU's slice of its NUMA node's cores, so two tenants of one host never share cores -- before,
    every tenant read the 0 as "GPU 0 of the node" and pinned itself to cpus 0-15 on the other socket."""
    from oavif_amd import hostinfo
    monkeypatch.setattr(hostinfo, "allowed_cpus", lambda: list(range(256)))
    monkeypatch.setattr(hostinfo, "cgroup_cpu_quota", lambda: 16.0)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    sysfs, dev = _fake_host(tmp_path / "a", ["0000:f1:00.0"], {"0000:f1:00.0"})           # the container's view
    nodes = hostinfo.kfd_gpu_nodes(sysfs, dev)
    assert [(n["pci"], n["render_minor"], n["accessible"]) for n in nodes] == [("0000:f1:00.0", 128, True)]
    assert hostinfo.visible_gpus(sysfs, dev) == ["0000:f1:00.0"]
    mine = hostinfo.node_core_sets(1, sysfs=sysfs, dev=dev)
    assert mine == [list(range(112, 128))]                          # the fourth GPU of node 1: its quarter, one thread per core
    assert hostinfo.cpu_numa_nodes(mine[0], sysfs) == [1]
    sysfs2, dev2 = _fake_host(tmp_path / "b", ["0000:d9:00.0"], {"0000:d9:00.0"})         # the neighbouring tenant
    other = hostinfo.node_core_sets(1, sysfs=sysfs2, dev=dev2)
    assert other == [list(range(96, 112))] and not set(other[0]) & set(mine[0])
    two = hostinfo.node_core_sets(2, procs_per_gpu=2, sysfs=sysfs, dev=dev)               # --procs-per-gpu 2 on that one GPU
    assert two == [list(range(112, 120)), list(range(120, 128))]
    # a whole node, the runtime's order not the PCI order, no *_VISIBLE_DEVICES: rank r is near HIP device r
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setattr(hostinfo, "cgroup_cpu_quota", lambda: None)
    order = ["0000:8b:00.0", "0000:0a:00.0", "0000:f1:00.0", "0000:23:00.0", "0000:a4:00.0", "0000:5a:00.0", "0000:d9:00.0", "0000:72:00.0"]
    sysfs3, dev3 = _fake_host(tmp_path / "c", order, set(order))
    assert hostinfo.visible_gpus(sysfs3, dev3) == order
    sets = hostinfo.node_core_sets(8, sysfs=sysfs3, dev=dev3)
    assert len(set(sum(sets, []))) == 256 and all(len(x) == 32 for x in sets)
    for r, addr in enumerate(order):
        assert hostinfo.cpu_numa_nodes(sets[r], sysfs3) == [0 if addr in ("0000:0a:00.0", "0000:23:00.0", "0000:5a:00.0", "0000:72:00.0") else 1], (r, addr)
    # HIP_VISIBLE_DEVICES picks among the GPUs the process can open, in the order it names them
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    assert hostinfo.visible_gpus(sysfs3, dev3) == ["0000:f1:00.0", "0000:8b:00.0"]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-abcdef")                               # a UUID: not interpreted
    assert hostinfo.visible_gpus(sysfs3, dev3) is None
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    # no KFD topology (or nothing that can be opened): the sysfs / *_VISIBLE_DEVICES rule of before
    sysfs4, dev4 = _fake_host(tmp_path / "d", ["0000:f1:00.0"], set())
    assert hostinfo.visible_gpus(sysfs4, dev4) is None
    assert hostinfo.node_core_sets(8, sysfs=sysfs4, dev=dev4) == hostinfo.rank_core_sets(
        8, cpus=hostinfo.sibling_order(range(256), sysfs4), gpu_cpulists=hostinfo.gpu_local_cpulists(sysfs4))


def test_idle_cores_are_picked_on_a_shared_host():
    """pick_idle_cpus: two samples of per-cpu ticks; a core is as busy as its busiest hardware
    thread; whole idle cores first; unreadable /proc/stat falls back to one thread per core."""
    from oavif_amd import hostinfo
    cpus = [0, 8, 1, 9, 2, 10, 3, 11]                       # four cores, sibling order
    core_of = {c: c % 8 for c in cpus}
    samples = iter([
        {c: (0, 0) for c in cpus},
        {0: (95, 100), 8: (0, 100), 1: (0, 100), 9: (0, 100), 2: (50, 100), 10: (0, 100), 3: (1, 100), 11: (0, 100)},
    ])
    got = hostinfo.pick_idle_cpus(cpus, 2, core_of=core_of, sampler=lambda: next(samples), sleep=lambda s: None)
    assert got == [1, 3]                                     # cores 1 and 3 are idle; core 0 is busy on its first thread
    samples = iter([{c: (0, 0) for c in cpus}, {c: (0, 100) for c in cpus}])
    assert hostinfo.pick_idle_cpus(cpus, 3, core_of=core_of, sampler=lambda: next(samples), sleep=lambda s: None) == [0, 1, 2]
    assert hostinfo.pick_idle_cpus(cpus, 3, core_of=core_of, sampler=lambda: {}, sleep=lambda s: None) == [0, 1, 2]
    assert hostinfo.pick_idle_cpus(cpus, 99, core_of=core_of) == [0, 1, 2, 3, 8, 9, 10, 11]
    ticks = hostinfo.read_cpu_ticks()
    assert all(t[1] >= t[0] >= 0 for t in ticks.values())


def test_pin_rank_takes_the_fixed_slice_and_reports_a_refused_affinity(monkeypatch):
    """the default placement is the deterministic slice (idle picking is opt-in:
    its round-3 default, a 0.1 s sample, cost the driver-run cpu_baseline a factor of four); a failed
    sched_setaffinity is reported in the result, not swallowed."""
    from oavif_amd import hostinfo
    cpus = list(range(32))
    monkeypatch.setattr(hostinfo, "allowed_cpus", lambda: cpus)
    monkeypatch.setattr(hostinfo, "cgroup_cpu_quota", lambda: 4.0)
    monkeypatch.setattr(hostinfo, "gpu_local_cpulists", lambda sysfs="/sys": [])
    monkeypatch.setattr(hostinfo, "sibling_order", lambda c, sysfs="/sys": list(c))
    monkeypatch.delenv("OAVIF_PIN", raising=False)
    calls = []
    monkeypatch.setattr(hostinfo.os, "sched_setaffinity", lambda pid, mask: calls.append(sorted(mask)))
    sampled = []
    monkeypatch.setattr(hostinfo, "pick_idle_cpus", lambda pool, n, sample_s=1.0, **k: sampled.append(sample_s) or list(pool)[8:8 + n])
    got = hostinfo.pin_rank(0, 1)
    assert list(got) == [0, 1, 2, 3] and got.pinned and got.how == "fixed slice" and calls[-1] == [0, 1, 2, 3] and not sampled
    got = hostinfo.pin_rank(0, 1, idle=True)                       # opt-in: a 1 s sample, never 0.1
    assert list(got) == [8, 9, 10, 11] and sampled == [1.0] and got.pinned
    monkeypatch.setenv("OAVIF_PIN", "idle")
    assert list(hostinfo.pin_rank(0, 1)) == [8, 9, 10, 11]
    assert list(hostinfo.pin_rank(1, 2)) == [16, 17]               # ranks of a multi-rank job keep their slices
    monkeypatch.delenv("OAVIF_PIN")

    def refuse(pid, mask):
        raise PermissionError("not permitted")
    monkeypatch.setattr(hostinfo.os, "sched_setaffinity", refuse)
    got = hostinfo.pin_rank(0, 1)
    assert list(got) == [0, 1, 2, 3] and not got.pinned and "PermissionError" in got.error
    # candidates for bench.py's cpu_baseline: the fixed slice first, then the following contiguous slices
    assert hostinfo.candidate_core_sets(4) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11]]


@pytest.fixture(scope="module")
def varied_dir(tmp_path_factory, hip_lib):
    from PIL import Image
    d = tmp_path_factory.mktemp("varied")
    rng = np.random.default_rng(5)
    for k in range(20):
        side = int(rng.integers(16, 160))
        img = Image.fromarray(rng.integers(0, 256, (side, side + 8, 3), dtype=np.uint8))
        img.save(d / f"v{k:02d}.png")
    return d


def test_world2_placement_csv_equal_loads_even_affinity_disjoint(varied_dir, tmp_path):
    """two ranks over files of very different sizes -- the gathered CSV equals
    the one-rank run's, the ranks' byte loads are within 10 %, and the ranks pinned themselves to
    disjoint core sets."""
    single, _ = _run(1, varied_dir, tmp_path / "v1.json")
    double, _ = _run(2, varied_dir, tmp_path / "v2.json")
    assert single == double and len(single) == 20
    def rows(path):   # every column but the wall-clock one
        return [r[:5] + r[6:] for r in csv.reader(open(path))]
    assert rows(str(tmp_path / "v1.json") + ".csv") == rows(str(tmp_path / "v2.json") + ".csv")
    r = [json.load(open(f"{tmp_path / 'v2.json'}.rank{k}")) for k in range(2)]
    assert sorted(r[0]["indices"] + r[1]["indices"]) == list(range(20))
    assert max(r[0]["bytes"], r[1]["bytes"]) <= 1.10 * min(r[0]["bytes"], r[1]["bytes"]), (r[0]["bytes"], r[1]["bytes"])
    if len(os.sched_getaffinity(0)) >= 2:
        assert r[0]["affinity"] and r[1]["affinity"] and not set(r[0]["affinity"]) & set(r[1]["affinity"])


def test_world8_gloo_256_images_of_uneven_sizes(tmp_path):
    """BASELINE configs[3]'s shape rehearsed on CPU: 256 images, EIGHT ranks over gloo,
    file sizes spread over a factor of 300 (five large files among many small ones).  The gathered CSV equals the one-rank run's, the ranks' byte
    loads are within 10 %, the shards differ in length and the gather buffer is sized by the longest one.
    (The encode is the scripted, GPU-free one of tests/_dist_worker.py: the search through the C ABI on a
    score table derived from the file name -- sharding, record packing and the one all_gather are the
    batch driver's own.)  No scaling figure follows from this: nothing multi-GPU has run on hardware."""
    d = tmp_path / "imgs256"
    d.mkdir()
    rng = np.random.default_rng(42)
    for k in range(256):
        n = 600_000 if k % 51 == 7 else int(2_000 * 20.0 ** rng.random())   # five large files among many small ones
        (d / f"img_{k:03d}.png").write_bytes(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    single, _ = _run(1, d, tmp_path / "s1.json")
    eight, summary = _run(8, d, tmp_path / "s8.json")
    assert single == eight and len(single) == 256 and all(r[2] == "ok" for r in single)

    def rows(path):   # every column but the wall-clock one
        return [r[:5] + r[6:] for r in csv.reader(open(path))]
    assert rows(str(tmp_path / "s1.json") + ".csv") == rows(str(tmp_path / "s8.json") + ".csv")
    r = [json.load(open(f"{tmp_path / 's8.json'}.rank{k}")) for k in range(8)]
    assert sorted(sum((x["indices"] for x in r), [])) == list(range(256))
    loads = [x["bytes"] for x in r]
    assert max(loads) <= 1.10 * min(loads), loads
    counts = [len(x["indices"]) for x in r]
    assert len(set(counts)) > 1 and all(x["gather_rows_per_rank"] == max(counts) for x in r)   # uneven shards, one buffer size
    assert "Ranks (GPUs): 8" in summary and "256 ok" in summary


def test_world2_gloo_matches_single_process(image_dir, tmp_path):
    single, _ = _run(1, image_dir, tmp_path / "w1.json")
    double, summary = _run(2, image_dir, tmp_path / "w2.json")
    assert single == double
    # measure.py's selection rule: png/jpg/jpeg (case-insensitive), sorted, others skipped
    names = [r[1] for r in single]
    assert names == sorted(names) and "skip.txt" not in names and len(names) == 8
    bad = [r for r in single if r[1] == "bad_one.png"][0]
    assert bad[2] == "error" and bad[7] is None
    ok = [r for r in single if r[2] == "ok"]
    assert len(ok) == 7 and all(1 <= r[5] <= 6 for r in ok)
    assert "Ranks (GPUs): 2" in summary and "7 ok" in summary and "1 errors" in summary


def test_worker_threads_do_not_change_results(image_dir, tmp_path):
    a, _ = _run(2, image_dir, tmp_path / "t1.json", workers=1)
    b, _ = _run(2, image_dir, tmp_path / "t3.json", workers=3)
    assert a == b


def test_csv_schema_matches_reference_tool(image_dir, tmp_path):
    """Header, column order and number formats of /root/reference/scripts/measure.py:178-206."""
    _run(2, image_dir, tmp_path / "w.json")
    rows = list(csv.reader(open(str(tmp_path / "w.json") + ".csv")))
    assert rows[0] == ["Image", "Original Bytes", "Final Bytes", "Savings Bytes", "Savings %",
                       "Encoding Time (ms)", "Passes", "Status", "Error"]
    assert len(rows) == 9
    for r in rows[1:]:
        if r[7] == "ok":
            assert r[2].isdigit() and r[6].isdigit()
            assert "." in r[4] and len(r[4].split(".")[1]) == 2      # "{:.2f}"
            assert "." in r[5] and len(r[5].split(".")[1]) == 2
        else:
            assert r[7] == "error" and r[2] == "" and r[6] == "" and r[8].startswith("Error processing")


from oavif_amd import batch, synth  # noqa: E402


def test_measure_py_positionals_are_accepted():
    """measure.py:111-123 takes `images_dir oavif_path output_csv`; an existing invocation must
    keep working (oavif_path is accepted and unused), and the two-positional form stays."""
    a = batch.parse_cli(["imgs", "./oavif", "out.csv", "--tolerance", "1.5", "--keep"])
    assert (a.images_dir, a.output_csv, a.tolerance, a.keep) == ("imgs", "out.csv", 1.5, True)
    b = batch.parse_cli(["imgs", "out.csv"])
    assert (b.images_dir, b.output_csv, b.tolerance, b.keep) == ("imgs", "out.csv", 2.0, False)
    with pytest.raises(SystemExit):
        batch.parse_cli(["imgs"])


def test_output_names_never_collide(tmp_path):
    files = [tmp_path / n for n in ("a.png", "a.jpg", "b.png", "c.JPEG")]
    names = batch.output_names(files)
    assert names == ["a_png.avif", "a_jpg.avif", "b.avif", "c.avif"]
    assert len(set(names)) == len(names)
    # names that collide after the first rule (a_png.webp-style stems, a.JPG + a.jpg)
    files = [tmp_path / n for n in ("a.png", "a.jpg", "a_png.jpeg", "b.JPG", "b.jpg", "c.png")]
    names = batch.output_names(files)
    assert len(set(names)) == len(names) == 6 and names[5] == "c.avif" and names[1] == "a_jpg.avif"


def test_batch_and_cli_share_one_encode_path_and_keep_alpha(tmp_path, monkeypatch):
    """The batch encodes what oavif encodes (io.zig:564: the RGBA source, not its RGB view): an
    RGBA PNG keeps its alpha plane, and the bytes equal the CLI mirror's for the same quantizer."""
    import io
    from types import SimpleNamespace

    from PIL import Image

    from oavif_amd import cli, tq
    rgb = synth.make_ref(64, 48, 3)
    alpha = np.tile(np.linspace(0, 255, 64, dtype=np.uint8), (48, 1))
    src = tmp_path / "x.png"
    Image.fromarray(np.dstack([rgb, alpha]), "RGBA").save(src)

    def fake_search(scorer, ref, codec, score_tgt, tolerance, max_pass):
        assert ref.shape == (48, 64, 3)                 # the scorer's reference is RGB8 (main.zig:86)
        dec, _size = codec(61)
        assert dec.shape == (48, 64, 3)                 # decodeAvifToRgb drops alpha (io.zig:654-663)
        return SimpleNamespace(q=61, score=80.5, num_pass=1, buf_q=61)
    monkeypatch.setattr(tq, "search_hip", fake_search)

    def fake_search_frames(scorer, ref, codec_frame, score_tgt, tolerance, max_pass):
        # with the libavif bridge on, the frame stays in libavif's RGBA rows
        assert ref.shape == (48, 64, 3)
        frame, _size = codec_frame(61)
        assert (frame.width, frame.height, frame.channels) == (64, 48, 4)
        assert frame.tight_rgb8().shape == (48, 64, 3)
        frame.close()
        return SimpleNamespace(q=61, score=80.5, num_pass=1, buf_q=61)
    monkeypatch.setattr(tq, "search_hip_frames", fake_search_frames)
    out = tmp_path / "x.avif"
    q, score, passes, nbytes = batch.encode_image(None, src, out)
    assert (q, passes, nbytes) == (61, 1, out.stat().st_size)
    assert Image.open(io.BytesIO(out.read_bytes())).mode == "RGBA"
    assert cli.main(["-q", "61", str(src), str(tmp_path / "cli.avif")]) == 0
    assert (tmp_path / "cli.avif").read_bytes() == out.read_bytes()


def test_exec_mode_runs_an_oavif_binary_per_image_like_measure_py(tmp_path):
    """`--exec`: measure.py:41-107 -- one `oavif [--tolerance T] <in> <out>` process per image, the pass count read
    off stderr with measure.py's own expression, a failing image recorded and the batch continued."""
    import stat
    fake = tmp_path / "oavif"
    fake.write_text("""#!/usr/bin/env python3
import sys
args = sys.argv[1:]
tol = None
if args[:1] == ["--tolerance"]:
    tol, args = args[1], args[2:]
src, out = args
sys.stderr.write("\\x1b[31moavif\\x1b[0m | fake\\n")
if "bad" in src:
    sys.stderr.write("error: DecodeFailed\\n")
    sys.exit(1)
n = 1 if "one" in src else 3
sys.stderr.write(f"Read 8x8, RGB, 8-bit, 10 bytes\\nFound q{40 + n} (score {79.5 + n:.2f}, {n} passes)\\n")   # main.zig:106 prints "passes" for 1 too
open(out, "wb").write(b"x" * (100 * n + (7 if tol else 0)))
""")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    files = []
    for name in ("a_one.png", "b_bad.png", "c.png"):
        p = tmp_path / name
        p.write_bytes(b"0" * 1000)
        files.append(p)
    out = tmp_path / "o"
    out.mkdir()
    assert batch.exec_image(str(fake), files[0], out / "a.avif") == (41, 80.5, 1, 100)        # "1 passes" (main.zig:106; measure.py:27's expression needs the "e")
    assert batch.exec_image(str(fake), files[2], out / "c.avif", tolerance=1.5) == (43, 82.5, 3, 307)
    with pytest.raises(RuntimeError) as e:
        batch.exec_image(str(fake), files[1], out / "b.avif")
    assert "non-zero exit status 1" in str(e.value) and "DecodeFailed" in str(e.value)
    res = batch.run_batch(files, lambda i, p: batch.exec_image(str(fake), p, out / f"{p.stem}.avif"), 0, 1)
    assert [(r.status, r.passes, r.final_bytes) for r in res] == [("ok", 1, 100), ("error", None, None), ("ok", 3, 300)]
    assert (res[0].q, res[2].q) == (41, 43)
    args = batch.parse_cli([str(tmp_path), str(fake), str(tmp_path / "r.csv"), "--exec"])
    assert args.exec_oavif and args.oavif_path == str(fake)
    with pytest.raises(SystemExit):
        batch.parse_cli([str(tmp_path), str(tmp_path / "r.csv"), "--exec"])


def test_a_cpuset_of_one_socket_does_not_shrink_a_tenants_slice(tmp_path, monkeypatch):
    """a cpuset-restricted container that was handed GPUs and cores of ONE socket.  Four of the host's eight GPUs
    have no local core inside the mask; that must not drop the placement to an even split of the allowed cores over all eight
    GPUs (a one-GPU tenant with 16 allowed cores ended up pinned to 2)."""
    from oavif_amd import hostinfo
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    monkeypatch.setattr(hostinfo, "cgroup_cpu_quota", lambda: None)
    sysfs, dev = _fake_host(tmp_path / "a", ["0000:f1:00.0"], {"0000:f1:00.0"})           # the fourth GPU of NUMA node 1
    # the mask covers node 1 only (cpus 64-127 + their SMT siblings): the GPU keeps its quarter of node 1, both threads of 16 cores
    monkeypatch.setattr(hostinfo, "allowed_cpus", lambda: list(range(64, 128)) + list(range(192, 256)))
    mine = hostinfo.node_core_sets(1, sysfs=sysfs, dev=dev)
    assert len(mine) == 1 and len(mine[0]) == 32 and hostinfo.cpu_numa_nodes(mine[0], sysfs) == [1]
    assert set(mine[0]) == set(range(112, 128)) | set(range(240, 256))
    # sixteen allowed cores of node 1, one GPU: the tenant gets its share of THOSE (a quarter: four GPUs hang off node 1), not 2 of 16
    monkeypatch.setattr(hostinfo, "allowed_cpus", lambda: list(range(64, 80)))
    mine = hostinfo.node_core_sets(1, sysfs=sysfs, dev=dev)
    assert mine == [list(range(76, 80))]
    # the mask covers the OTHER socket only: nothing is near the job's GPU, so the allowed cores go to the GPUs the job really
    # has (one), not to an eighth each
    monkeypatch.setattr(hostinfo, "allowed_cpus", lambda: list(range(0, 16)))
    assert hostinfo.node_core_sets(1, sysfs=sysfs, dev=dev) == [list(range(0, 16))]
    assert hostinfo.node_core_sets(2, procs_per_gpu=2, sysfs=sysfs, dev=dev) == [list(range(0, 8)), list(range(8, 16))]
    # gpu_slices itself: GPUs outside the mask are None, the others share their intersection
    sl = hostinfo.gpu_slices(list(range(64, 80)), [list(range(0, 64))] * 4 + [list(range(64, 128))] * 4)
    assert sl[:4] == [None] * 4 and sl[4:] == [list(range(64, 68)), list(range(68, 72)), list(range(72, 76)), list(range(76, 80))]


def test_ranks_that_share_a_gpu_in_a_rehearsal_still_pin_near_it(tmp_path, monkeypatch):
    """Round 6 (profiles/r06_bench_n4_gloo_bare_rehearsal.json showed it): four ranks of a gloo rehearsal on a box with ONE
    GPU -- rank r uses device r % 1 -- were pinned to cpus 0-3 / 4-7 / ... of NUMA node 0 although the GPU hangs off node 1:
    with fewer GPUs than ranks the placement fell back to "GPU 0 of the node".  Rank r now takes its part of the slice of
    the GPU it really uses."""
    from oavif_amd import hostinfo
    monkeypatch.setattr(hostinfo, "allowed_cpus", lambda: list(range(256)))
    monkeypatch.setattr(hostinfo, "cgroup_cpu_quota", lambda: 16.0)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    sysfs, dev = _fake_host(tmp_path / "a", ["0000:8b:00.0"], {"0000:8b:00.0"})           # the first GPU of NUMA node 1
    sets = hostinfo.node_core_sets(4, sysfs=sysfs, dev=dev)
    assert sets == [list(range(64, 68)), list(range(68, 72)), list(range(72, 76)), list(range(76, 80))]   # one thread per core
    assert all(hostinfo.cpu_numa_nodes(x, sysfs) == [1] for x in sets)                    # every rank on the GPU's node
    assert set(sum(sets, [])) <= set(range(64, 80)) | set(range(192, 208))                # inside that GPU's own slice
    # two GPUs, four ranks: ranks 0 and 2 near the first, 1 and 3 near the second (device = rank % 2)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    sysfs2, dev2 = _fake_host(tmp_path / "b", ["0000:8b:00.0", "0000:0a:00.0"], {"0000:8b:00.0", "0000:0a:00.0"})
    sets = hostinfo.node_core_sets(4, sysfs=sysfs2, dev=dev2)
    assert [hostinfo.cpu_numa_nodes(x, sysfs2) for x in sets] == [[1], [0], [1], [0]] and len(set(sum(sets, []))) == 16
