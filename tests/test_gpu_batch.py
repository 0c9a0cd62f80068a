"""BASELINE configs[3] end to end on the GPU box: a batch of 1920x1080 PNGs through the batch
driver (the scripts/measure.py counterpart) with the real scorer, sharded over TWO ranks.

The pool gives one GPU per box, so the two ranks share device 0 through the batch driver's own
`--procs-per-gpu 2` (rank r -> GPU r // 2; the one gather then runs over gloo, since RCCL does
not place two ranks on one device).  What this covers of the 8-GPU path: the launcher, the
largest-first dealing, host-core pinning, one scorer context per worker thread on the rank's
device, the gather of result records, CSV and summary on rank 0.  What it does not: RCCL between
devices and 8 devices -- unmeasured until a SCALE record exists.  RCCL itself is exercised as far as one GPU
allows: a process group of ONE rank on backend "nccl" through which the batch driver sends its record gather
(test_record_gather_runs_through_rccl_single_rank).  -m gpu only.
"""
import csv
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oavif_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rows(path):
    with open(path) as f:
        rows = list(csv.reader(f))
    t = rows[0].index("Encoding Time (ms)")
    return rows[0], [[c for k, c in enumerate(r) if k != t] for r in rows[1:]]


def test_batch_of_1080p_pngs_two_ranks_equals_one_rank(tmp_path):
    from PIL import Image
    img_dir = tmp_path / "images"
    img_dir.mkdir()
    base = synth.make_ref(1920, 1080, 77)
    for k in range(16):   # 16 distinct frames from one synthetic 1080p frame
        a = np.roll(base, k * 120, axis=1)
        if k & 1:
            a = a[::-1]
        if k & 2:
            a = a[..., ::-1]
        Image.fromarray(np.ascontiguousarray(a)).save(img_dir / f"img_{k:02d}.png", compress_level=1)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    one = tmp_path / "one.csv"
    p1 = subprocess.run([sys.executable, "-m", "oavif_amd.batch", str(img_dir), str(one), "--workers", "4",
                         "--out-dir", str(tmp_path / "out1")],
                        env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p1.returncode == 0, p1.stderr[-2000:]
    two = tmp_path / "two.csv"
    env2 = dict(env)   # two ranks on ONE GPU: --procs-per-gpu 2 (rank r -> GPU r // 2, the gather over gloo)
    # measure.py's three positionals: images_dir oavif_path output_csv
    p2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                         "-m", "oavif_amd.batch", str(img_dir), "./oavif", str(two), "--workers", "4",
                         "--procs-per-gpu", "2",
                         "--out-dir", str(tmp_path / "out2")],
                        env=env2, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p2.returncode == 0, p2.stderr[-2000:]
    h1, r1 = _rows(one)
    h2, r2 = _rows(two)
    assert h1 == h2 and len(r1) == 16
    assert r1 == r2                                     # same q / bytes / passes / status per image
    assert all(r[h1.index("Status") - 1] == "ok" for r in r1)
    assert "Ranks (GPUs): 2" in p2.stdout and "Images: 16 ok" in p2.stdout and "ranks per GPU: 2" in p2.stdout
    # both ranks really worked: the stderr lines carry the rank that searched each image
    assert "[rank 0]" in p2.stderr and "[rank 1]" in p2.stderr
    # the "N passes" phrase measure.py parses (measure.py:27) is on every per-image line
    assert p2.stderr.count(" passes)") >= 16
    # round 6: the same job as ONE bare command, as measure.py is one command -- `--gpus 1 --procs-per-gpu 2`: the driver
    # starts its own two ranks (oavif_amd/launch.py), relays rank 0's summary and leaves with their code
    three = tmp_path / "three.csv"
    env3 = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p3 = subprocess.run([sys.executable, "-m", "oavif_amd.batch", "--gpus", "1", "--procs-per-gpu", "2", str(img_dir), str(three),
                         "--workers", "4", "--out-dir", str(tmp_path / "out3")],
                        env=env3, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p3.returncode == 0, p3.stderr[-2000:]
    assert "oavif_amd.batch: launching 2 ranks" in p3.stderr
    h3, r3 = _rows(three)
    assert h3 == h1 and r3 == r1
    assert "Ranks (GPUs): 2" in p3.stdout and "Images: 16 ok" in p3.stdout and "ranks per GPU: 2" in p3.stdout
    assert "the rendezvous store" in p3.stdout or "Collective: backend gloo, world 2" in p3.stdout


def test_record_gather_runs_through_rccl_single_rank(tmp_path):
    """`gather_records`' device-tensor all_gather has to have executed on RCCL at least
    once.  A one-GPU box cannot hold two RCCL ranks, so: backend "nccl", world size 1, the early return of a
    one-rank group lifted (OAVIF_GATHER_ALWAYS=1) -- first the function alone on a hand-made record table with
    a buffer larger than the shard, then the whole batch driver, whose CSV must equal the plain run's.
    Subprocesses: a process group is process-wide state that must not leak into the other tests."""
    code = """
import os, sys, numpy as np
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="%d", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
from oavif_amd import batch
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
rng = np.random.default_rng(0)
local = np.zeros((37, batch.RECORD_FIELDS), np.float64)
local[:, 0] = rng.permutation(37)
local[:, 1:] = rng.uniform(-1, 100, (37, batch.RECORD_FIELDS - 1))
assert dist.get_backend() == "nccl"
got = batch.gather_records(local, 37, device=torch.device("cuda", 0), per_rank=64, always=True)
assert got.shape == (37, batch.RECORD_FIELDS) and np.array_equal(got, local[np.argsort(local[:, 0])])
plain = batch.gather_records(local, 37, device=torch.device("cuda", 0), per_rank=64)     # early return of a one-rank group
assert np.array_equal(plain, got)
dist.barrier()
dist.destroy_process_group()
print("rccl gather ok")
""" % (ROOT, _free_port())
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0 and "rccl gather ok" in p.stdout, p.stderr[-2000:]
    if not synth.have_avif():
        return
    from PIL import Image
    img_dir = tmp_path / "images"
    img_dir.mkdir()
    for k in range(4):
        Image.fromarray(synth.make_ref(320 + 16 * k, 200, 880 + k)).save(img_dir / f"g{k}.png")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    outs = []
    for tag, extra in (("plain", {}), ("rccl", {"OAVIF_GATHER_ALWAYS": "1", "MASTER_PORT": str(_free_port())})):
        csv_path = tmp_path / f"{tag}.csv"
        q = subprocess.run([sys.executable, "-m", "oavif_amd.batch", str(img_dir), str(csv_path), "--workers", "2",
                            "--out-dir", str(tmp_path / tag), "--collective-json", str(tmp_path / f"{tag}.json")],
                           env=dict(env, **extra), cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert q.returncode == 0, q.stderr[-2000:]
        outs.append(_rows(csv_path))
    assert outs[0] == outs[1] and len(outs[0][1]) == 4
    # round 5: what the batch says about its process group and its placement.  The rank pins itself (this pool's cgroup
    # grants 16 of 256 CPUs) to the slice of the GPU it REALLY has -- found through the KFD topology, not by reading
    # ROCR_VISIBLE_DEVICES=0 as "GPU 0 of the node" -- so its cores are on its GPU's NUMA node and nothing is warned about
    import json
    import oavif_amd
    info = oavif_amd.query_device(0)
    plain, rccl = (json.load(open(tmp_path / f"{t}.json")) for t in ("plain", "rccl"))
    assert plain["collective"]["backend"] == "none" and plain["images_ok"] == 4 and plain["images_per_s"] > 0
    c = rccl["collective"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and "RCCL, device tensors" in c["gathered_through"]
    assert c["problems"] == [] and c["warnings"] == [], c
    r = c["ranks"][0]
    assert r["pci_bus_id"] == info["pci_bus_id"] and r["numa_node"] == info["numa_node"]
    assert r["assumed_pci_bus_id"] in (None, info["pci_bus_id"])     # the KFD topology named the GPU the runtime then gave us
    if r["pinned"] and info["numa_node"] >= 0:
        assert r["cpu_numa_nodes"] == [info["numa_node"]], r
    assert "Collective: backend nccl, world 1, 1 distinct device(s)" in q.stdout


def test_exec_mode_with_the_compiled_host_equals_the_in_process_batch(tmp_path):
    """`--exec`: the batch as measure.py runs it (measure.py:151-158: one `oavif` process per image) with this
    repository's compiled C host as the `oavif` binary, against the in-process batch: the same CSV -- bytes,
    passes, status -- for every image (both run the reference's libavif calls and the same search and scorer)."""
    from PIL import Image
    from oavif_amd import build as obuild
    if obuild.host_needs_build():
        obuild.build_host()
    img_dir = tmp_path / "images"
    img_dir.mkdir()
    for k, (w, h) in enumerate(((320, 240), (257, 199), (400, 300), (192, 144))):
        a = synth.make_ref(w, h, 40 + k)
        if k == 2:   # an RGBA source: alpha reaches the encoder, not the scorer
            a = np.dstack([a, np.tile(np.linspace(0, 255, w, dtype=np.uint8), (h, 1))])
        Image.fromarray(a).save(img_dir / f"img_{k}.png")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    runs = {}
    for tag, extra in (("inproc", []), ("exec", [obuild.HOST_PATH, "--exec"])):
        out = tmp_path / f"{tag}.csv"
        cmd = [sys.executable, "-m", "oavif_amd.batch", str(img_dir)] + extra[:1] + [str(out)] + extra[1:] + \
              ["--workers", "2", "--tolerance", "1.5", "--out-dir", str(tmp_path / f"o_{tag}"), "--keep"]
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        runs[tag] = _rows(out)
        assert "Images: 4 ok, 0 no-output, 0 errors" in p.stdout
    assert runs["exec"] == runs["inproc"]
    for k in range(4):
        assert (tmp_path / "o_exec" / f"img_{k}.avif").read_bytes() == (tmp_path / "o_inproc" / f"img_{k}.avif").read_bytes()
