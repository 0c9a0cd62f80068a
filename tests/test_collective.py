"""The `collective` record and the placement refusal (oavif_amd/collective.py), on
CPU: two ranks exchange their descriptions through the rendezvous store exactly as bench.py / the batch driver do --
BEFORE any communicator exists (the device description is injected: there is no GPU here) -- and the rules of a real
multi-GPU run are applied to them: two RCCL ranks on one PCI bus id, or fewer visible devices than local ranks, end
the run with rc 4 on EVERY rank without the communicator ever being created (a stand-in for RCCL's initialisation that
fails on that very placement shows the old order would have ended in RCCL's error instead); then the process group is
opened on the same store and one all_gather over it confirms the records.
What a line must carry: backend, world size, per rank the host / device index / bus id / NUMA node / pinned cores,
and the library versions.  The reference has no counterpart: scripts/measure.py:137-158 is a sequential loop."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_collective_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, out, **env):
    port = _free_port()
    procs = []
    for rank in range(world):
        e = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **{k: str(v) for k, v in env.items()})
        procs.append(subprocess.Popen([sys.executable, WORKER, str(out)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=180)[0] for p in procs]
    return [p.returncode for p in procs], [json.load(open(f"{out}.rank{r}")) for r in range(world)], logs


def collective_prefixes():
    from oavif_amd import collective
    return collective.ENV_PREFIXES


def _init_called(out, world):
    return [os.path.exists(f"{out}.init_called.rank{r}") for r in range(world)]


def test_two_ranks_on_two_devices_are_accepted_and_described(tmp_path, hip_lib):
    rcs, recs, logs = _run(2, tmp_path / "ok", FAKE_BUS="0000:05:00.0,0000:15:00.0", CLAIM_BACKEND="nccl",
                           HSA_TEST_MARKER="from-the-test", NCCL_DEBUG="WARN", UNRELATED_VARIABLE="not recorded")
    assert rcs == [0, 0], logs
    assert recs[0]["collective"] == recs[1]["collective"] or all(      # every rank holds the same gathered record
        recs[0]["collective"][k] == recs[1]["collective"][k] for k in ("backend", "world_size", "ranks", "problems"))
    c = recs[0]["collective"]
    assert c["backend"] == "nccl" and c["world_size"] == 2 and c["distinct_devices"] == 2 and c["problems"] == []
    # exchanged through the store first, confirmed over the process group afterwards -- and the line says so
    assert c["gathered_through"].startswith("the rendezvous store (TCP key-value exchange), before any communicator existed")
    assert "confirmed by one all_gather over the job's process group (RCCL, CPU tensors)" in c["gathered_through"]
    assert _init_called(tmp_path / "ok", 2) == [True, True]
    assert [r["rank"] for r in c["ranks"]] == [0, 1] and [r["device"] for r in c["ranks"]] == [0, 1]
    for r, bus in zip(c["ranks"], ("0000:05:00.0", "0000:15:00.0")):
        assert r["pci_bus_id"] == bus and r["numa_node"] in (0, 1) and r["arch"].startswith("gfx950")
        assert r["host"] == socket.gethostname() and r["pid"] > 0 and r["n_cpus"] >= 1 and r["cpus"]
        assert r["local_rank"] == r["rank"] and r["pinned"] in (True, False) and isinstance(r["cpu_numa_nodes"], list)
        # the runtime-steering variables each rank ran under travel with its record
        assert r["env"]["HSA_TEST_MARKER"] == "from-the-test" and r["env"]["NCCL_DEBUG"] == "WARN"
        assert all(k.startswith(collective_prefixes()) for k in r["env"])
    if len(os.sched_getaffinity(0)) >= 2:                            # the ranks pinned themselves to disjoint cores
        from oavif_amd import hostinfo
        a, b = (set(hostinfo.parse_cpulist(r["cpus"])) for r in c["ranks"])
        assert not a & b
    assert set(c["versions"]) >= {"torch", "hip", "rccl"}


def test_two_rccl_ranks_on_one_bus_id_are_refused_on_every_rank(tmp_path, hip_lib):
    rcs, recs, _ = _run(2, tmp_path / "dup", FAKE_BUS="0000:05:00.0,0000:05:00.0", CLAIM_BACKEND="nccl")
    assert rcs == [4, 4]
    for r in recs:
        assert len(r["problems"]) == 1 and "ranks 0 and 1 both sit on the GPU at 0000:05:00.0" in r["problems"][0]
        assert r["collective"]["distinct_devices"] == 1 and r["collective"]["problems"] == r["problems"]
        assert "confirmed" not in r["collective"]["gathered_through"]
    # the refusal is reached BEFORE the communicator -- the stand-in for RCCL's initialisation (which fails
    # on exactly this placement) was never called ...
    assert _init_called(tmp_path / "dup", 2) == [False, False]
    # ... whereas the old order (communicator first, records over it) ends in RCCL's own error, not in the refusal
    rcs, recs, _ = _run(2, tmp_path / "legacy", FAKE_BUS="0000:05:00.0,0000:05:00.0", CLAIM_BACKEND="nccl", LEGACY_GROUP_CHECK=1)
    assert rcs == [1, 1] and all("Duplicate GPU detected" in r["legacy_error"] for r in recs)


def test_a_failing_process_group_ends_every_rank_with_rc_5_and_the_backends_message(tmp_path, hip_lib):
    """an RCCL initialisation failure on a legal placement is not retried and not swallowed: every rank
    prints the backend's message and leaves with rc 5; the record carries the error."""
    rcs, recs, logs = _run(2, tmp_path / "fail", FAKE_BUS="0000:05:00.0,0000:15:00.0", CLAIM_BACKEND="nccl", FAIL_INIT=1)
    assert rcs == [5, 5]
    for r, log in zip(recs, logs):
        assert r["rc"] == 5 and "ncclSystemError" in r["collective"]["error"] and r["collective"]["problems"] == []
        assert "the RCCL process group could not be opened" in log and "ncclSystemError" in log
    assert _init_called(tmp_path / "fail", 2) == [True, True]


def test_fewer_devices_than_ranks_are_refused_with_and_without_a_rendezvous(tmp_path, hip_lib):
    rcs, recs, _ = _run(2, tmp_path / "few", FAKE_BUS="0000:05:00.0,0000:15:00.0", CLAIM_BACKEND="nccl", FAKE_DEVCOUNT=1)
    assert rcs == [4, 4] and "torch.cuda.device_count() < local world" in recs[0]["problems"][0]
    # before any rendezvous: the preflight every rank runs on its own (same count on every rank of a host)
    rcs, recs, _ = _run(2, tmp_path / "pre", FAKE_BUS="a,b", CLAIM_BACKEND="nccl", PREFLIGHT_DEVCOUNT=1)
    assert rcs == [4, 4] and all("1 visible device(s) for 2 ranks" in r["preflight"] for r in recs)


def test_the_store_exchange_also_works_under_torch_distributed_run(tmp_path, hip_lib):
    """The driver starts N > 1 as `python -m torch.distributed.run ... bench.py --gpus N`: there the rendezvous store is the
    elastic agent's (every worker is a client of it, TORCHELASTIC_USE_AGENT_STORE), not one rank 0 hosts.  The same
    open_group must exchange, judge, open the group and confirm under that launcher too -- and refuse under it."""
    env = dict(os.environ, FAKE_BUS="0000:05:00.0,0000:15:00.0", CLAIM_BACKEND="nccl", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = tmp_path / "tr"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), WORKER, str(out)], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    recs = [json.load(open(f"{out}.rank{r}")) for r in range(2)]
    assert all(r["rc"] == 0 and r["collective"]["problems"] == [] and r["collective"]["distinct_devices"] == 2 for r in recs)
    assert "confirmed by one all_gather" in recs[0]["collective"]["gathered_through"] and _init_called(out, 2) == [True, True]
    out = tmp_path / "tr_dup"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), WORKER, str(out)], capture_output=True, text=True, timeout=300,
                       env=dict(env, FAKE_BUS="0000:05:00.0,0000:05:00.0"))
    assert p.returncode != 0 and ("exitcode  : 4" in p.stderr or "exitcode: 4" in p.stderr), p.stderr[-3000:]
    assert _init_called(out, 2) == [False, False]


def test_rules_on_hand_made_records():
    from oavif_amd import collective
    mk = lambda rank, host, bus: {"rank": rank, "host": host, "pci_bus_id": bus}
    ok = [mk(0, "a", "0000:05:00.0"), mk(1, "a", "0000:15:00.0"), mk(2, "b", "0000:05:00.0"), mk(3, "b", "0000:15:00.0")]
    assert collective.problems(ok, "nccl", 4, 2, 2) == []                       # the same bus id on ANOTHER host is fine
    assert collective.problems(ok[:3], "nccl", 4) != []                         # a rank is missing
    dup = [mk(0, "a", "0000:05:00.0"), mk(1, "a", "0000:05:00.0")]
    assert collective.problems(dup, "gloo", 2) == []      # what --procs-per-gpu 2 asks for, over gloo: not a problem there
    assert "rank 1 reports no PCI bus id" in collective.problems([ok[0], mk(1, "a", None)], "nccl", 2)[0]
    assert "rank set" in collective.problems([ok[0], ok[0]], "gloo", 2)[0]      # a rank twice
    assert collective.problems([{"undecodable": "x"}, ok[1]], "gloo", 2)
    assert collective.problems([mk(0, "a", "0000:05:00.0")], "nccl", 1, 1, 1) == []   # the N = 1 line
    rec = collective.rank_record(0, 0, None, device_info={"pci_bus_id": "0000:d9:00.0", "numa_node": 1, "arch": "gfx950"})
    assert collective._decode(collective._encode(rec)) == rec and len(collective._encode(rec)) == collective.RECORD_BYTES
    long = dict(rec, cpus=",".join(str(2 * k) for k in range(600)))
    assert len(collective._encode(long)) == collective.RECORD_BYTES and collective._decode(collective._encode(long))["rank"] == 0
    d = collective.describe("nccl", 1, [rec], "x")
    assert d["world_size"] == 1 and d["distinct_devices"] == 1 and d["ranks"] == [rec] and d["warnings"] == []
    # a rank pinned to the other socket than its GPU's is reported (a warning on the line, not a refusal)
    far = dict(rec, pinned=True, cpus="0-15", cpu_numa_nodes=[0], numa_node=1, rank=0)
    near = dict(far, cpus="112-127", cpu_numa_nodes=[1])
    assert collective.warnings([near]) == [] and collective.problems([far], "nccl", 1, 1, 1) == []
    w = collective.warnings([far])
    assert len(w) == 1 and "pinned to cpus 0-15 of NUMA node(s) [0]" in w[0] and "hangs off node 1" in w[0]
    assert collective.warnings([dict(far, pinned=None)]) == [] and collective.warnings([dict(far, numa_node=-1)]) == []
    # the KFD-order assumption of the placement code is checked against what the runtime reports
    assert collective.warnings([dict(near, assumed_pci_bus_id=near["pci_bus_id"])]) == []
    w = collective.warnings([dict(near, assumed_pci_bus_id="0000:0a:00.0")])
    assert len(w) == 1 and "took HIP device" in w[0] and "0000:0a:00.0" in w[0]


def test_bench_and_batch_leave_with_rc_4_before_any_rendezvous_when_devices_are_missing():
    """Source-level: both entry points run collective.preflight, then collective.open_group (store exchange -> judge ->
    process group), never an init_process_group of their own for the multi-rank job, and hand a refusal's code on (they
    cannot be started here: no GPU)."""
    for path in ("bench.py", os.path.join("oavif_amd", "batch.py")):
        src = open(os.path.join(ROOT, path)).read()
        src = src[src.index("def main("):]
        assert src.index("collective.preflight(backend, local_world)") < src.index("collective.open_group(")
        assert "dist.init_process_group" not in src
        assert "return 4" in src and "return rc_" in src
        assert src.index("launch.needs_self_launch(") < src.index("import torch")   # the supervisor never touches torch / a GPU
    src = open(os.path.join(ROOT, "oavif_amd", "collective.py")).read()
    og = src[src.index("def open_group("):]
    assert og.index("check_in(") < og.index("return coll, RC_REFUSED") < og.index("dist.init_process_group(")


# ---- on the MI355X box (one GPU): what RCCL and the launcher really do ----------------------------------------

@pytest.mark.gpu
def test_bench_line_carries_the_collective_record_through_rccl(hip_lib):
    """the N = 1 line carries `collective` with world_size 1 gathered through RCCL (a process
    group of one rank on backend nccl, device tensors), the rank's device as the library sees it, and the versions."""
    import oavif_amd
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    c = d["collective"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and c["distinct_devices"] == 1 and c["problems"] == [], c
    # the N = 1 record goes the way an N > 1 job's does: store first, then the RCCL group (device tensors) as confirmation
    assert c["gathered_through"].startswith("the rendezvous store") and "error" not in c
    assert "confirmed by one all_gather over the job's process group (RCCL, device tensors)" in c["gathered_through"]
    assert "RCCL process group of one rank" in c["gathered_through"]
    info = oavif_amd.query_device(0)
    r = c["ranks"][0]
    assert (r["rank"], r["device"], r["pci_bus_id"], r["numa_node"]) == (0, 0, info["pci_bus_id"], info["numa_node"])
    assert r["arch"].startswith("gfx950") and r["n_cpus"] >= 1
    assert c["versions"]["rccl"][0].isdigit() and c["versions"]["hip"]
    m = d["default_search_mode"]
    assert m["blur"].startswith("recursive") and 0.2 < m["ms_per_pass"] < 0.6 and m["MP_per_s"] < d["value"]


@pytest.mark.gpu
def test_two_rccl_ranks_on_a_one_gpu_box_are_refused_by_bench_and_batch(tmp_path, hip_lib):
    """Two ranks over RCCL need two devices: on this pool's one-GPU boxes both entry points leave with rc 4 on every
    rank BEFORE any rendezvous (collective.preflight), so no rank waits for the other and nothing plausible is printed --
    under torch.distributed.run and as the BARE command (`python3 bench.py --gpus 2`: the parent launches the ranks itself
    and hands their code on, oavif_amd/launch.py)."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box with exactly one GPU")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    (tmp_path / "imgs").mkdir()
    from PIL import Image
    from oavif_amd import synth
    Image.fromarray(synth.make_ref(64, 48, 1)).save(tmp_path / "imgs" / "a.png")
    for what in (["bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1"],
                 ["-m", "oavif_amd.batch", str(tmp_path / "imgs"), str(tmp_path / "o.csv"), "--out-dir", str(tmp_path / "out")]):
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(_free_port()), *what], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert p.returncode != 0
        assert p.stderr.count("refusing to run: 1 visible device(s) for 2 ranks") == 2, p.stderr[-3000:]
        assert "exitcode  : 4" in p.stderr or "exitcode: 4" in p.stderr, p.stderr[-3000:]
        assert '"value"' not in p.stdout and not (tmp_path / "o.csv").exists()
    for what in (["bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1"],
                 ["-m", "oavif_amd.batch", "--gpus", "2", str(tmp_path / "imgs"), str(tmp_path / "o.csv"), "--out-dir", str(tmp_path / "out")]):
        p = subprocess.run([sys.executable, *what], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert p.returncode == 4, (p.returncode, p.stderr[-3000:])                       # the refusal's code is the command's code
        assert "launching 2 ranks" in p.stderr and p.stderr.count("refusing to run: 1 visible device(s) for 2 ranks") == 2
        assert '"value"' not in p.stdout and not (tmp_path / "o.csv").exists()


@pytest.mark.gpu
def test_bare_two_rank_rehearsal_over_gloo_prints_one_line_with_both_ranks(tmp_path, hip_lib):
    """`OAVIF_BENCH_BACKEND=gloo python3 bench.py --gpus 2 ...` (bare): the parent launches two ranks that share this box's
    GPU, the records travel through the rendezvous store before the process group exists and are confirmed over it, every
    rank leaves the group before rank 0's extras, and the parent's last stdout line is rank 0's JSON line (rc 0)."""
    env = dict(os.environ, OAVIF_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(p.stdout.splitlines()[-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["scores"]) == 2 and len(d["per_rank_own_ms_per_step"]) == 2
    c = d["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and [r["rank"] for r in c["ranks"]] == [0, 1] and c["problems"] == []
    assert c["gathered_through"].startswith("the rendezvous store") and "confirmed by one all_gather" in c["gathered_through"]
    assert all(isinstance(r["env"], dict) for r in c["ranks"])
    assert d["roofline"]["kernel_ms"] > 0          # rank 0's extras ran after the group was left
