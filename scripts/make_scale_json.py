#!/usr/bin/env python3
"""profiles/scale.json from the outputs of scripts/gpu_scale.sh (gpurun_out/TAG/): per N the bench line's MP/s and
the batch's images/s with their speed-up over N = 1, the `collective` record of each run, and the CSV agreement of
every batch with the N = 1 batch.  A run that is missing, failed or was refused (rc 4: two ranks on one GPU) is
listed as such; nothing is extrapolated."""
import csv
import json
import os
import sys


def _json_line(path):
    try:
        lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except Exception:
        return None


def _rc(path):
    try:
        return int(open(path).read().strip().split("=")[1])
    except Exception:
        return None


def _passes(path):
    """mean / max of the CSV's Passes column (how many scorer passes an image of the batch takes)"""
    try:
        rows = list(csv.reader(open(path)))
        k = rows[0].index("Passes")
        v = [int(r[k]) for r in rows[1:] if r[k].isdigit()]
        return {"mean": round(sum(v) / len(v), 2), "max": max(v), "min": min(v)} if v else None
    except Exception:
        return None


def _rows(path):
    try:
        rows = list(csv.reader(open(path)))
        t = rows[0].index("Encoding Time (ms)")
        return [[c for k, c in enumerate(r) if k != t] for r in rows[1:]]
    except Exception:
        return None


def main(out_dir):
    res = {"source": "scripts/gpu_scale.sh", "devices": (open(os.path.join(out_dir, "devices.txt")).read().strip()
                                                         if os.path.exists(os.path.join(out_dir, "devices.txt")) else None),
           "bench": [], "batch": []}
    base_mp = base_ips = None
    base_rows = _rows(os.path.join(out_dir, "batch_n1.csv"))
    for n in (1, 2, 4, 8):
        b = _json_line(os.path.join(out_dir, f"bench_n{n}.json"))
        rc = _rc(os.path.join(out_dir, f"bench_n{n}.rc"))
        if b is None and rc is None:
            continue
        e = {"n_gpus": n, "rc": rc}
        if b and b.get("value"):
            base_mp = b["value"] if n == 1 else base_mp
            e.update({"MP_per_s": b["value"], "ms_per_step": b.get("ms_per_step"), "scaling": b.get("scaling"),
                      "speedup_over_1": round(b["value"] / base_mp, 3) if base_mp else None,
                      "scores": b.get("scores"), "collective": b.get("collective")})
        else:
            e.update({"status": "refused" if rc == 4 else "failed", "collective": (b or {}).get("collective")})
        res["bench"].append(e)
    for n in (1, 2, 4, 8):
        j = None
        try:
            j = json.load(open(os.path.join(out_dir, f"batch_n{n}.json")))
        except Exception:
            pass
        rc = _rc(os.path.join(out_dir, f"batch_n{n}.rc"))
        if j is None and rc is None:
            continue
        e = {"n_gpus": n, "rc": rc}
        if j:
            base_ips = j["images_per_s"] if n == 1 else base_ips
            rows = _rows(os.path.join(out_dir, f"batch_n{n}.csv"))
            e.update({"images": j["images"], "images_ok": j["images_ok"], "wall_s": j["wall_s"], "images_per_s": j["images_per_s"],
                      "speedup_over_1": round(j["images_per_s"] / base_ips, 3) if base_ips and j["images_per_s"] else None,
                      "csv_equals_n1": (rows == base_rows) if rows is not None and base_rows is not None else None,
                      "passes_per_image": _passes(os.path.join(out_dir, f"batch_n{n}.csv")),
                      "workers_per_rank": j.get("workers_per_rank"), "collective": j["collective"]})
        else:
            e["status"] = "refused" if rc == 4 else "failed"
        res["batch"].append(e)
    tgt = [e for e in res["batch"] if e["n_gpus"] == 8 and e.get("speedup_over_1")]
    res["north_star_batch_6x_at_8_gpus"] = ({"speedup": tgt[0]["speedup_over_1"], "met": tgt[0]["speedup_over_1"] >= 6.0}
                                            if tgt else "not measured: no 8-GPU run in this record")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
