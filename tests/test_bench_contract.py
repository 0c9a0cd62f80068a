"""bench.py's one-line JSON contract, checked on the line recorded on the MI355X
(profiles/r01_bench.json) and on bench.py's own constants.  CPU only."""
import ast
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    return json.load(open(os.path.join(ROOT, "profiles", "r01_bench.json")))


def test_required_keys_and_types():
    d = _line()
    for k, t in [("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int),
                 ("warmup", int), ("ms_per_step", (int, float)), ("higher_is_better", bool),
                 ("scaling", str), ("dtype", str), ("data", str), ("config", dict),
                 ("roofline", dict), ("cpu_baseline", dict)]:
        assert k in d and isinstance(d[k], t), k
    assert d["vs_baseline"] is None                # BASELINE.md publishes no number for this metric
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["unit"] == "MP/s" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]


def test_value_is_consistent_with_ms_per_step():
    d = _line()
    mp = d["config"]["width"] * d["config"]["height"] / 1e6
    assert abs(d["value"] - d["n_gpus"] * mp / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3


def test_roofline_and_cpu_baseline_objects():
    d = _line()
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["peak"] == 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    assert r["traffic"] is None or r["traffic"] < r["algorithmic_bytes"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["quantizer_match_vs_cpu"]["identical"] is True
    assert d["cached_reference"]["bit_identical_to_pair_score"] is True


def test_bench_byte_model_matches_survey():
    """85.97 B per scale-0 pixel for the whole score; the pyramid writes are
    24 * (1/4 + 1/16 + ...) of it."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    consts = {n.targets[0].id: ast.literal_eval(n.value) for n in ast.parse(src).body
              if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name)
              and n.targets[0].id.startswith(("ALGO_", "HBM_"))}
    assert consts["ALGO_BYTES_PER_PX_SCORE"] == 85.97
    assert consts["HBM_PEAK_GBS"] == 8000.0
    # pyramid writes at 4K: 24 B per pixel of scales 1..5
    w, h, px = 3840, 2160, 0
    for _ in range(5):
        w, h = (w + 1) // 2, (h + 1) // 2
        px += w * h
    assert abs((85.97 - consts["ALGO_BYTES_PER_PX_MARCH"]) - 24 * px / (3840 * 2160)) < 0.01
