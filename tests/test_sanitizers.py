"""CPU-only sanitizer builds of the host-side native code (no sanitizer exists for the GPU on
this pool): the search logic of oavif_amd/csrc/tq.cpp under ASan + UBSan and under TSan (its
speculative search hands probe waves to caller threads), and the CPU checker oracle/ssimu2_oracle.c
under ASan + UBSan.  The harnesses are tests/c/tq_sanitize.cpp and tests/c/oracle_sanitize.c."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
TQ = os.path.join(ROOT, "oavif_amd", "csrc", "tq.cpp")
ORACLE = os.path.join(ROOT, "oracle", "ssimu2_oracle.c")


def _run(cmd, exe, args=(), env=None):
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    p = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]
    return p.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ missing")
@pytest.mark.parametrize("san,iters", [("address,undefined", "600"), ("thread", "200")])
def test_tq_search_code_is_clean_under_sanitizers(tmp_path, san, iters):
    exe = str(tmp_path / "tq_san")
    out = _run(["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={san}", "-fno-sanitize-recover=all",
                "-I", INC, TQ, os.path.join(ROOT, "tests", "c", "tq_sanitize.cpp"), "-o", exe, "-lpthread"],
               exe, [iters], env={"TSAN_OPTIONS": "halt_on_error=1"})
    assert "tq_sanitize ok" in out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc missing")
def test_oracle_is_clean_under_sanitizers(tmp_path):
    exe = str(tmp_path / "or_san")
    out = _run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                "-ffp-contract=off", ORACLE, os.path.join(ROOT, "tests", "c", "oracle_sanitize.c"),
                "-o", exe, "-lm"], exe)
    assert "oracle_sanitize ok" in out
