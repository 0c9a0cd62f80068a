"""scripts/make_scale_json.py (the record scripts/gpu_scale.sh leaves in profiles/scale.json): per N the bench
line's MP/s and the batch's images/s with their speed-up over N = 1 and the `collective` record; a run that was
refused (rc 4: two ranks on one GPU) or is missing is listed as such, nothing is extrapolated, and north_star's
">= 6x at 8 GPUs" is only answered from an 8-GPU run.  CPU only; the inputs are hand-made files of the shapes
bench.py and the batch driver write."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "scripts", "make_scale_json.py")


def _coll(n):
    return {"backend": "nccl", "world_size": n, "distinct_devices": n, "problems": [],
            "ranks": [{"rank": r, "device": r, "pci_bus_id": f"0000:{5 + 16 * r:02x}:00.0"} for r in range(n)]}


def _write(d, n, mp=None, ips=None, rc=0, rows=None):
    (d / f"bench_n{n}.rc").write_text(f"rc={rc}\n")
    if mp is not None:
        (d / f"bench_n{n}.json").write_text("noise\n" + json.dumps({"value": mp, "ms_per_step": 8.29 * n / mp * 1e3, "scaling": "weak",
                                                                     "scores": [60.1] * n, "collective": _coll(n)}) + "\n")
    elif rc == 4:
        (d / f"bench_n{n}.json").write_text(json.dumps({"value": None, "error": "placement refused", "collective": _coll(n)}) + "\n")
    (d / f"batch_n{n}.rc").write_text(f"rc={0 if ips else rc}\n")
    if ips is not None:
        (d / f"batch_n{n}.json").write_text(json.dumps({"collective": _coll(n), "images": 256, "images_ok": 256, "wall_s": 256 / ips,
                                                        "images_per_s": ips, "workers_per_rank": 16, "ranks_per_gpu": 1}))
        with open(d / f"batch_n{n}.csv", "w") as f:
            f.write("Image,Original Bytes,Final Bytes,Savings Bytes,Savings %,Encoding Time (ms),Passes,Status,Error\n")
            for r in rows or [["a.png", "10", "5", "5", "50.00", str(100.0 + n), "2", "ok", ""]]:
                f.write(",".join(r) + "\n")


def _run(d):
    p = subprocess.run([sys.executable, TOOL, str(d)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return json.loads(p.stdout)


def test_one_gpu_record_claims_nothing_about_eight(tmp_path):
    (tmp_path / "devices.txt").write_text("devices visible: 1\n")
    _write(tmp_path, 1, mp=53000.0, ips=80.0)
    r = _run(tmp_path)
    assert [e["n_gpus"] for e in r["bench"]] == [1] and r["bench"][0]["speedup_over_1"] == 1.0
    assert r["batch"][0]["images_per_s"] == 80.0 and r["batch"][0]["csv_equals_n1"] is True
    assert r["bench"][0]["collective"]["world_size"] == 1
    assert r["north_star_batch_6x_at_8_gpus"] == "not measured: no 8-GPU run in this record"


def test_full_node_record_with_a_refused_run(tmp_path):
    (tmp_path / "devices.txt").write_text("devices visible: 8\n")
    _write(tmp_path, 1, mp=50000.0, ips=80.0)
    _write(tmp_path, 2, mp=99000.0, ips=158.0)
    _write(tmp_path, 4, rc=4)                                              # bench refused; the batch of that N never ran
    (tmp_path / "batch_n4.rc").write_text("rc=4\n")
    _write(tmp_path, 8, mp=380000.0, ips=500.0, rows=[["a.png", "10", "5", "5", "50.00", "1.0", "3", "ok", ""]])   # another pass count
    r = _run(tmp_path)
    by = {e["n_gpus"]: e for e in r["bench"]}
    assert by[2]["speedup_over_1"] == 1.98 and by[8]["speedup_over_1"] == 7.6 and by[8]["collective"]["distinct_devices"] == 8
    assert by[4]["status"] == "refused" and "MP_per_s" not in by[4] and by[4]["collective"]["world_size"] == 4
    bb = {e["n_gpus"]: e for e in r["batch"]}
    assert bb[4]["status"] == "refused" and bb[2]["speedup_over_1"] == 1.975
    assert bb[8]["csv_equals_n1"] is False                               # a differing CSV is reported, not hidden
    assert r["north_star_batch_6x_at_8_gpus"] == {"speedup": 6.25, "met": True}
