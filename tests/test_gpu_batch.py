"""BASELINE configs[3] end to end on the GPU box: a batch of 1920x1080 PNGs through the batch
driver (the scripts/measure.py counterpart) with the real scorer, sharded over TWO ranks.

The pool gives one GPU per box, so the two ranks share device 0 through the batch driver's own
`--procs-per-gpu 2` (rank r -> GPU r // 2; the one gather then runs over gloo, since RCCL does
not place two ranks on one device).  What this covers of the 8-GPU path: the launcher, the
largest-first dealing, host-core pinning, one scorer context per worker thread on the rank's
device, the gather of result records, CSV and summary on rank 0.  What it does not: RCCL itself
and 8 devices -- unmeasured until a SCALE record exists.  -m gpu only.
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
