"""Multi-rank path on CPU: world_size 2 over gloo must produce the same gathered records as a
single process (images shard i mod N, one all_gather of fixed-size records)."""
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
    out = tmp_path / "x.avif"
    q, score, passes, nbytes = batch.encode_image(None, src, out)
    assert (q, passes, nbytes) == (61, 1, out.stat().st_size)
    assert Image.open(io.BytesIO(out.read_bytes())).mode == "RGBA"
    assert cli.main(["-q", "61", str(src), str(tmp_path / "cli.avif")]) == 0
    assert (tmp_path / "cli.avif").read_bytes() == out.read_bytes()
