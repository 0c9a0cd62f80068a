"""The blur-mode pin kit (tests/golden/pin_kit/, scripts/pin_blur_mode.py): pairs on which
the three blur modes differ by 9x to 380x north_star's +-0.01, with the checker's score in each mode, and the
script that tells from fssimu2's scores of the same files which mode it follows.  Parity vs fssimu2 stays
UNPINNED here: the recorded scores are this repo's checker's (self-oracle); the kit is what unpins it.

CPU: the committed files are what the kit says (sha256), the checker reproduces the recorded scores, one
generated full-size pair regenerates to its sha256, the classifier's verdicts.  -m gpu: the HIP scorer in its
three modes reproduces every recorded score, committed pairs read through the native PNG ingest."""
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pin_blur_mode", os.path.join(ROOT, "scripts", "pin_blur_mode.py"))
kit = importlib.util.module_from_spec(spec)
spec.loader.exec_module(kit)


@pytest.fixture(scope="module")
def doc():
    return kit.load_kit()


def _pixels(p):
    if p["kind"] == "committed":
        return kit.read_png_rgb8(os.path.join(kit.KIT, p["ref"])), kit.read_png_rgb8(os.path.join(kit.KIT, p["dist"]))
    return kit.generate(p["name"])


def test_kit_modes_are_far_apart_and_files_are_the_scored_ones(doc):
    assert doc["tolerance"] == 0.01 and len(doc["pairs"]) >= 6
    big = 0
    for p in doc["pairs"]:
        s = p["scores"]
        gaps = [abs(s[a] - s[b]) for a, b in (("fir", "recursive"), ("fir", "recursive_fma"), ("recursive", "recursive_fma"))]
        assert min(gaps) > 8 * doc["tolerance"], p["name"]          # every pair tells the three modes apart
        big += min(gaps) >= 0.5
        if p["kind"] == "committed":
            ref, dst = _pixels(p)
            assert ref.shape == (p["height"], p["width"], 3)
            assert kit.sha256_pixels(ref) == p["sha256_ref"] and kit.sha256_pixels(dst) == p["sha256_dist"]
    assert big >= 2      # VERDICT: pairs where the three modes are >= 0.5 point apart (the full-size ones)


def test_checker_reproduces_the_recorded_scores(doc, oracle):
    modes = {"fir": oracle.BLUR_FIR, "recursive": oracle.BLUR_IIR, "recursive_fma": oracle.BLUR_IIR_FMA}
    for p in doc["pairs"]:
        if p["kind"] != "committed" and p["name"] != "g1080_blockq1":   # one full-size pair on the CPU, the rest on the GPU box
            continue
        ref, dst = _pixels(p)
        assert kit.sha256_pixels(ref) == p["sha256_ref"] and kit.sha256_pixels(dst) == p["sha256_dist"], p["name"]
        for m, flag in modes.items():
            got = oracle.compute_ssimu2(ref, dst, flag, omp=True)
            assert abs(got - p["scores"][m]) < 1e-9, (p["name"], m, got)


def test_classifier_verdicts(doc, tmp_path):
    rec = {p["name"]: p["scores"] for p in doc["pairs"]}
    for m in kit.MODES:
        verdict, rows, worst = kit.classify(doc, {n: s[m] + 0.004 for n, s in rec.items()})
        assert verdict.startswith(f"MATCH: {m} ") and worst[m] < 0.0041 and all(r[3] == m for r in rows)
    # a scorer that follows none of them (e.g. halfway between two modes)
    verdict, _, worst = kit.classify(doc, {n: 0.5 * (s["fir"] + s["recursive"]) for n, s in rec.items()})
    assert verdict.startswith("NO MODE MATCHES") and min(worst.values()) > 0.01
    # one small pair alone still decides (gaps are >= 8 tolerances everywhere)
    verdict, _, _ = kit.classify(doc, {"a384_avif80": rec["a384_avif80"]["recursive_fma"] - 0.009})
    assert verdict.startswith("MATCH: recursive_fma")
    with pytest.raises(SystemExit):
        kit.classify(doc, {"nope": 1.0})
    # the command line: results file in, verdict out
    f = tmp_path / "r.txt"
    f.write_text("# fssimu2 0.1.1\n" + "\n".join(f"{n}, {s['recursive']:.6f}" for n, s in rec.items()) + "\n")
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_blur_mode.py"), str(f)], capture_output=True, text=True)
    assert out.returncode == 0 and "MATCH: recursive (" in out.stdout
    lst = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_blur_mode.py"), "--list"], capture_output=True, text=True)
    assert "g4k_noise1" in lst.stdout and "a384_avif92" in lst.stdout


def test_png_round_trip_of_the_kits_writer(tmp_path):
    rng = np.random.default_rng(1)
    px = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    p = tmp_path / "x.png"
    p.write_bytes(kit.png_rgb8(px))
    assert np.array_equal(kit.read_png_rgb8(str(p)), px)


@pytest.mark.gpu
def test_hip_scorer_reproduces_every_recorded_score_in_its_mode(doc, hip_lib):
    """All seven pairs (two of them 3840x2160) in the three modes of a context: within 1e-4 of the checker's
    recorded score -- so a maintainer who finds fssimu2 on, say, `recursive` gets that arithmetic from the
    device by one setter.  Committed pairs are read through the native PNG ingest (oavif_png_decode)."""
    from oavif_amd import Ssimu2, _lib, png
    modes = {"fir": _lib.BLUR_FIR, "recursive": _lib.BLUR_RECURSIVE, "recursive_fma": _lib.BLUR_RECURSIVE_FMA}
    with Ssimu2(0) as s:
        for p in doc["pairs"]:
            if p["kind"] == "committed":
                ref = png.load_png(open(os.path.join(kit.KIT, p["ref"]), "rb").read())[0]
                dst = png.load_png(open(os.path.join(kit.KIT, p["dist"]), "rb").read())[0]
            else:
                ref, dst = kit.generate(p["name"])
            assert kit.sha256_pixels(ref) == p["sha256_ref"] and kit.sha256_pixels(dst) == p["sha256_dist"], p["name"]
            for m, flag in modes.items():
                s.set_blur(flag)
                got = s.compute_ssimu2(ref, dst)
                assert abs(got - p["scores"][m]) < 1e-4, (p["name"], m, got, p["scores"][m])
