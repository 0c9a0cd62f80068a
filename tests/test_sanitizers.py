"""CPU-only sanitizer builds of the host-side native code (no sanitizer exists for the GPU on
this pool): the search logic of oavif_amd/csrc/tq.cpp under ASan + UBSan and under TSan (its
speculative search hands probe waves to caller threads), and the CPU checker oracle/ssimu2_oracle.c
under ASan + UBSan, and the native PNG ingest (oavif_amd/csrc/png_ingest.cpp) under ASan + UBSan
with thousands of corrupted files.  The harnesses are tests/c/tq_sanitize.cpp,
tests/c/oracle_sanitize.c and tests/c/png_sanitize.cpp."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
TQ = os.path.join(ROOT, "oavif_amd", "csrc", "tq.cpp")
ORACLE = os.path.join(ROOT, "oracle", "ssimu2_oracle.c")
PNG = os.path.join(ROOT, "oavif_amd", "csrc", "png_ingest.cpp")


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


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ missing")
def test_png_ingest_survives_corrupted_files_under_sanitizers(tmp_path):
    """Untrusted input: 24 valid seed files (every colour type / depth, with and without Adam7) and
    7,200 corruptions of them -- truncations, bit flips, bit flips with the chunk CRC repaired so
    that they reach the zlib / filter / palette code -- must each end in an error code or a clean
    decode, with no ASan / UBSan report."""
    exe = str(tmp_path / "png_san")
    out = _run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                "-I", INC, PNG, os.path.join(ROOT, "tests", "c", "png_sanitize.cpp"), "-o", exe, "-lz"],
               exe, ["300"])
    assert "png_sanitize ok" in out


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ missing")
def test_fast_inflate_equals_zlib_and_survives_corruption_under_sanitizers(tmp_path):
    """oavif_amd/csrc/inflate_fast.h (the DEFLATE decoder of the PNG ingest) against zlib itself: 1,250 valid
    streams of every level / strategy / content decoded through output strips of awkward sizes must give the
    source bytes; 10,000 corrupted or truncated streams must give an error or exactly zlib's bytes -- and zlib
    must not accept anything this decoder refuses.  ASan + UBSan, no recovery."""
    exe = str(tmp_path / "inflate_diff")
    out = _run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                os.path.join(ROOT, "tests", "c", "inflate_diff.cpp"), "-o", exe, "-lz"], exe, ["250"])
    assert "inflate_diff ok" in out
