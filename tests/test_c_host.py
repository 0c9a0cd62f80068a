"""oavif_amd/csrc/oavif_host.c: main.zig's flow (main.zig:37-117) as a compiled C program over the two public
headers and libavif -- the drop-in end to end with no Python in the loop.

CPU part: it builds, its `-q` bypass writes the bytes the Python mirror writes (both make the reference's libavif
calls, io.zig:544-636), argument errors carry the reference's names, and without a GPU a search fails loudly.
GPU part: a whole search (PNG / PAM in, AVIF out) prints the same "Found q.. (score .., N passes)" line and
writes the same file as the Python mirror on the same scorer."""
import os
import re
import subprocess

import numpy as np
import pytest

from oavif_amd import avif_bridge as ab
from oavif_amd import build as obuild
from oavif_amd import cli, pam, synth

pytestmark = pytest.mark.skipif(not ab.available(), reason=f"libavif bridge unavailable: {ab.why_unavailable()}")


@pytest.fixture(scope="module")
def host(hip_lib):
    if obuild.host_needs_build():
        obuild.build_host()
    return obuild.HOST_PATH


def _env(**kw):
    e = dict(os.environ, OAVIF_LIBAVIF=ab._find_library())
    e.update(kw)
    return e


def _run(host, args, **env):
    return subprocess.run([host, *args], capture_output=True, text=True, timeout=300, env=_env(**env))


def _inputs(tmp_path, w=200, h=136, seed=1):
    from PIL import Image
    ref = synth.make_ref(w, h, seed)
    png = tmp_path / "a.png"
    Image.fromarray(ref).save(png)
    rgba = np.dstack([ref, np.tile(np.linspace(0, 255, w, dtype=np.uint8), (h, 1))])
    p = tmp_path / "b.pam"
    p.write_bytes(pam.write_pam(rgba))
    return ref, png, p


def test_bypass_writes_the_python_mirrors_bytes(host, tmp_path, capsys):
    _ref, png, p = _inputs(tmp_path)
    for args, src in ((["-q", "61", "--tenbit", "0"], png),
                      (["-q", "40", "--tenbit", "0", "--quality-alpha", "90", "--tune", "ssim", "-s", "8"], p),
                      (["--quality", "70", "--color-primaries", "1", "--transfer-characteristics", "13",
                        "--matrix-coefficients", "6", "--tenbit", "0"], png)):
        r = _run(host, [*args, str(src), str(tmp_path / "c.avif")])
        assert r.returncode == 0, r.stderr
        assert cli.main([*args, str(src), str(tmp_path / "p.avif")]) == 0
        perr = capsys.readouterr().err.splitlines()
        cerr = r.stderr.splitlines()
        assert cerr[1:4] == perr[1:4]                       # Read / Encoding / Compressed lines (main.zig:78-98)
        assert re.fullmatch(r"Read 200x136, RGBA?, 8-bit, \d+ bytes", cerr[1])
        assert (tmp_path / "c.avif").read_bytes() == (tmp_path / "p.avif").read_bytes()


def test_16_bit_palette_and_icc_sources_go_the_mirrors_way(host, tmp_path, capsys):
    """io.loadPNG's output rules through the C ABI's loader in both hosts: a 16-bit RGB PNG (hbd: RGBA16 to the
    encoder after `>> 8`, or `>> 6` where 10-bit can be written; `>> 8` for the scorer, io.zig:63-95,587,602), a
    palette PNG (RGBA8) and an ICC profile handed on to the AVIF (io.zig:556-560): same stderr lines, same bytes."""
    import io as _io
    from PIL import Image, ImageCms
    from tests.test_png import write_png
    rng = np.random.default_rng(5)
    ref = synth.make_ref(96, 64, 9)
    deep = (ref.astype(np.uint16) << 8) | rng.integers(0, 256, ref.shape, dtype=np.uint16)
    icc = ImageCms.ImageCmsProfile(ImageCms.createProfile("sRGB")).tobytes()
    (tmp_path / "deep.png").write_bytes(write_png(deep, 2, 16, icc=icc))
    idx = (ref[..., :1] >> 4).astype(np.uint8)
    plte = np.stack([np.arange(16) * 17, 255 - np.arange(16) * 17, (np.arange(16) * 40) % 256], axis=1)
    (tmp_path / "pal.png").write_bytes(write_png(idx, 3, 4, plte=plte))
    for name, read_line in (("deep.png", "Read 96x64, RGBA, 16-bit, "), ("pal.png", "Read 96x64, RGBA, 8-bit, ")):
        r = _run(host, ["-q", "70", str(tmp_path / name), str(tmp_path / "c.avif")])
        assert r.returncode == 0, r.stderr
        assert cli.main(["-q", "70", str(tmp_path / name), str(tmp_path / "p.avif")]) == 0
        perr = capsys.readouterr().err.splitlines()
        cerr = r.stderr.splitlines()
        assert cerr[1].startswith(read_line) and cerr[1:4] == perr[1:4], (cerr, perr)
        assert (tmp_path / "c.avif").read_bytes() == (tmp_path / "p.avif").read_bytes()
    r = _run(host, ["-q", "70", str(tmp_path / "deep.png"), str(tmp_path / "d.avif")])
    assert Image.open(_io.BytesIO((tmp_path / "d.avif").read_bytes())).info.get("icc_profile") == icc


def test_default_depth_is_reported_like_the_mirror(host, tmp_path):
    _ref, png, _p = _inputs(tmp_path)
    r = _run(host, ["-q", "50", str(png), str(tmp_path / "c.avif")])     # --tenbit 1 is the default
    assert r.returncode == 0
    want = 10 if ab.supports_depth(10) else 8
    assert f"Encoding [q50, speed 9, {want}-bit]" in r.stderr
    assert ("note: the reference would write 10-bit" in r.stderr) == (want == 8)
    assert ab.probe((tmp_path / "c.avif").read_bytes())["depth"] == want


def test_argument_errors_carry_the_references_names(host, tmp_path):
    _ref, png, _p = _inputs(tmp_path, 32, 32)
    cases = [(["--speed", "11", str(png), "x.avif"], "InvalidOptionValue", "Error: --speed must be between 0 and 10"),
             (["--tenbit", "2", str(png), "x.avif"], "InvalidOptionValue", "Error: --tenbit must be 0 or 1"),
             (["--max-pass"], "MissingOptionValue", "Error: Missing --max-pass value"),
             (["--tune", "psnr", str(png), "x.avif"], "InvalidTuneMode", ""),
             ([str(png)], "MissingInputOrOutput", ""),
             ([str(png), "x.avif", "extra"], "UnexpectedArgument", "Error: Unexpected argument: extra"),
             (["-q", "50", str(tmp_path / "nope.png"), "x.avif"], "FileNotFound", ""),
             (["-q", "50", "in.jpg", "x.avif"], "UnsupportedImageFormat", "")]
    for args, name, line in cases:
        r = _run(host, args)
        assert r.returncode == 1 and f"error: {name}" in r.stderr, (args, r.stderr)
        assert line in r.stderr
    bad = tmp_path / "bad.pam"
    bad.write_bytes(b"P7\nWIDTH 2\nHEIGHT 2\nDEPTH 3\nMAXVAL 65535\nTUPLTYPE RGB\nENDHDR\n" + bytes(24))
    assert "error: UnsupportedPamMaxVal" in _run(host, ["-q", "50", str(bad), "x.avif"]).stderr    # io.zig:369
    # the PAM acceptance rules of io.zig:309-406, error for error as oavif_amd.pam (the Python mirror) raises them
    body = bytes(2 * 2 * 4)
    pams = {"NotAPamFile": b"P6\n2 2\n255\n" + body,
            "HeaderNotFound": b"P7\nWIDTH 2\nHEIGHT 2\nDEPTH 3\nMAXVAL 255\n" + body.replace(b"\n", b"x"),
            "InvalidPamDimensions": b"P7\nWIDTH 2\nDEPTH 3\nMAXVAL 255\nENDHDR\n" + body,
            "UnsupportedPamDepth": b"P7\nWIDTH 1\nHEIGHT 1\nDEPTH 5\nMAXVAL 255\nENDHDR\n" + body,
            "PamTupleMismatch": b"P7\nWIDTH 2\nHEIGHT 2\nDEPTH 3\nMAXVAL 255\nTUPLTYPE rgb_alpha\nENDHDR\n" + body,
            "UnsupportedPamTuple": b"P7\nWIDTH 2\nHEIGHT 2\nDEPTH 1\nMAXVAL 255\nTUPLTYPE BLACKANDWHITE\nENDHDR\n" + body,
            "InsufficientDataInFile": b"P7\nWIDTH 3\nHEIGHT 3\nDEPTH 4\nMAXVAL 255\nENDHDR\n" + body,
            "InvalidCharacter": b"P7\nWIDTH 2x\nHEIGHT 2\nDEPTH 3\nMAXVAL 255\nENDHDR\n" + body}
    for name, blob in pams.items():
        f = tmp_path / f"{name}.pam"
        f.write_bytes(blob)
        with pytest.raises(pam.PamError) as pe:
            pam.load_pam(blob)
        assert pe.value.name == name
        assert f"error: {name}" in _run(host, ["-q", "50", str(f), str(tmp_path / "x.avif")]).stderr, name
    ok = tmp_path / "ok.pam"   # comments, CR LF, '+' and '_' in numbers, a blank-line header end, an unknown tuple type, trailing bytes
    ok.write_bytes(b"P7\r\n# made by hand\r\nWIDTH +2\nHEIGHT 2_\nDEPTH 3\nMAXVAL 255 \nTUPLTYPE WHATEVER\n\n" + bytes(range(12)) + b"tail")
    raster, w_, h_, ch_ = pam.load_pam(ok.read_bytes())
    assert (w_, h_, ch_) == (2, 2, 3)
    assert _run(host, ["-q", "50", "--tenbit", "0", str(ok), str(tmp_path / "ok.avif")]).returncode == 0
    assert ab.probe((tmp_path / "ok.avif").read_bytes())["width"] == 2
    r = _run(host, ["-q", "50", str(png), str(tmp_path / "x.avif")], OAVIF_LIBAVIF="/nonexistent/libavif.so")
    assert r.returncode == 1 and "error: LibavifUnavailable" in r.stderr


def test_help_and_version_lead_and_a_fifo_input_is_read_to_its_end(host, tmp_path, capsys):
    """(i) main.zig:46-61 -- -h / --help / -v / --version are honoured while they are the leading
    arguments (the usage text is parse_args.zig:180-238's, the Python mirror prints the same); behind another
    argument `-h` is an ordinary (unknown) argument.  (ii) a non-seekable input (a FIFO named x.pam) is read until
    end of file instead of by ftell's -1."""
    import threading
    r = _run(host, ["-h"])
    assert r.returncode == 0 and "usage:  oavif [options] <in> <out.avif>" in r.stderr
    cli.print_usage()
    mirror = capsys.readouterr().err
    assert r.stderr.split("\n", 1)[1].strip() == mirror.strip()
    assert "show this help" in _run(host, ["--version", "--help", "whatever"]).stderr   # help wins (main.zig:60-61)
    v = _run(host, ["-v"])
    assert v.returncode == 0 and "scorer oavif_amd ssimu2 gfx950" in v.stderr and "libavif" in v.stderr
    assert _run(host, ["-q", "50", "-h", "x"]).returncode == 1
    _ref, _png, p = _inputs(tmp_path, 40, 24)
    fifo = tmp_path / "pipe.pam"
    os.mkfifo(fifo)
    t = threading.Thread(target=lambda: fifo.open("wb").write(p.read_bytes()))
    t.start()
    r = _run(host, ["-q", "50", "--tenbit", "0", str(fifo), str(tmp_path / "f.avif")])
    t.join()
    assert r.returncode == 0, r.stderr
    assert _run(host, ["-q", "50", "--tenbit", "0", str(p), str(tmp_path / "g.avif")]).returncode == 0
    assert (tmp_path / "f.avif").read_bytes() == (tmp_path / "g.avif").read_bytes()


def test_search_without_a_gpu_fails_loudly(host, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _ref, png, _p = _inputs(tmp_path, 64, 48)
    r = _run(host, [str(png), str(tmp_path / "x.avif")])
    assert r.returncode == 1 and "error: NoDevice" in r.stderr and not (tmp_path / "x.avif").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("blur", ["recursive", "fir"])
def test_search_end_to_end_equals_the_python_mirror(host, tmp_path, capsys, monkeypatch, blur):
    """The C host and the Python mirror drive the same search code (oavif_tq_find_target_quality), the same
    libavif calls and the same scorer: same probes, same q, same score to the printed digits, same file."""
    _ref, png, p = _inputs(tmp_path, 320, 240, seed=303)
    monkeypatch.setenv("OAVIF_SSIMU2_BLUR", blur)
    for args, src in ((["--score-tgt", "75", "--tolerance", "1.5", "--tenbit", "0"], png),
                      (["-t", "88", "--tolerance", "1", "--max-pass", "8", "--tenbit", "0", "--quality-alpha", "80"], p)):
        r = _run(host, [*args, str(src), str(tmp_path / "c.avif")], OAVIF_SSIMU2_BLUR=blur, OAVIF_HOST_TIMES="1")
        assert r.returncode == 0, r.stderr
        assert cli.main([*args, str(src), str(tmp_path / "p.avif")]) == 0
        perr = capsys.readouterr().err.splitlines()
        cerr = [l for l in r.stderr.splitlines() if not l.startswith("  [")]     # without the phase timeline
        assert any(l.startswith("  [") and "first probe encoded and decoded" in l for l in r.stderr.splitlines())
        assert cerr[1:5] == perr[1:5], (cerr, perr)          # Read, Searching, Found, Compressed
        m = re.fullmatch(r"Found q(\d+) \(score (-?\d+\.\d{2}), (\d+) passes\)", cerr[3])
        assert m and re.search(r"(\d+)\s+passes?", cerr[3]).group(1) == m.group(3)      # measure.py:27
        assert any(l.startswith("times: encode ") for l in cerr)
        assert (tmp_path / "c.avif").read_bytes() == (tmp_path / "p.avif").read_bytes()
        # the same search with its probes fanned over 4 scorer contexts and threads (BASELINE configs[2];
        # oavif_tq_find_target_quality_speculative from compiled code): the sequential search's result
        f = _run(host, [*args, str(src), str(tmp_path / "f.avif")], OAVIF_SSIMU2_BLUR=blur, OAVIF_HOST_TIMES="1",
                 OAVIF_PROBE_FANOUT="4")
        assert f.returncode == 0, f.stderr
        ferr = [l for l in f.stderr.splitlines() if not l.startswith("  [")]
        assert ferr[1:5] == cerr[1:5], (ferr, cerr)
        assert any(l.startswith("speculative: ") for l in ferr)
        assert (tmp_path / "f.avif").read_bytes() == (tmp_path / "c.avif").read_bytes()


def test_host_is_clean_under_sanitizers(tmp_path, hip_lib):
    """ASan + UBSan build of the host (CPU; libavif and the product library stay uninstrumented): the bypass on
    PNG / RGBA PAM, every argument error, unreadable and corrupted inputs, and a search that stops at NoDevice --
    no report, no leak on any path (LeakSanitizer on)."""
    import shutil
    import torch
    if shutil.which("gcc") is None:
        pytest.skip("gcc missing")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "oavif_amd", "lib")
    exe = str(tmp_path / "host_san")
    subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=gnu11",
                    "-I", os.path.join(root, "include"), os.path.join(root, "oavif_amd", "csrc", "oavif_host.c"),
                    "-o", exe, "-L", libdir, "-loavif_hip", "-ldl", "-lm", "-lpthread", f"-Wl,-rpath,{libdir}",
                    "-Wl,-rpath-link,/opt/rocm/lib"], check=True, capture_output=True)
    _ref, png, p = _inputs(tmp_path, 96, 64)
    raw = png.read_bytes()
    (tmp_path / "cut.png").write_bytes(raw[: len(raw) // 2])
    flipped = bytearray(raw)
    flipped[len(raw) // 2] ^= 0x5A
    (tmp_path / "flip.png").write_bytes(bytes(flipped))
    praw = p.read_bytes()
    (tmp_path / "cut.pam").write_bytes(praw[: len(praw) - 100])
    (tmp_path / "hdr.pam").write_bytes(b"P7\nWIDTH 99999\nHEIGHT 99999\nDEPTH 4\nMAXVAL 255\nENDHDR\n" + bytes(64))
    out = str(tmp_path / "o.avif")
    cases = [(["-q", "60", "--tenbit", "0", str(png), out], 0), (["-q", "60", str(p), out], 0),
             (["-q", "30", "--quality-alpha", "99", "--tune", "ssimulacra2", "--tenbit", "0", str(p), out], 0),
             (["--speed", "99", str(png), out], 1), (["--max-pass"], 1), ([str(png)], 1),
             (["-q", "5", str(tmp_path / "nope.png"), out], 1), (["-q", "5", str(tmp_path / "cut.png"), out], 1),
             (["-q", "5", str(tmp_path / "flip.png"), out], 1), (["-q", "5", str(tmp_path / "cut.pam"), out], 1),
             (["-q", "5", str(tmp_path / "hdr.pam"), out], 1)]
    if not torch.cuda.is_available():
        cases.append(([str(png), out], 1))          # the search path up to ssimu2_ctx_create
    for args, want in cases:
        r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300,
                           env=_env(ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1",
                                    OAVIF_HOST_ATEXIT="1"))      # leave through exit(): LeakSanitizer reports there
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (args, r.stderr[-3000:])
        assert r.returncode == want, (args, r.stderr[-800:])


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_search_path_of_the_host_under_sanitizers_with_a_stub_scorer(tmp_path, san):
    """The host's whole search path on the CPU: oavif_host.c + the real search code (tq.cpp) + the real PNG ingest
    + the real libavif, with tests/c/stub_scorer.c standing in for the GPU scorer (test infrastructure: a monotone
    stand-in score).  ASan + UBSan + LSan, and TSan for the pthread fan-out of the speculative search
    (OAVIF_PROBE_FANOUT): no report, and the fanned search prints and writes what the sequential one does."""
    import shutil
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("gcc / g++ missing")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "oavif_amd", "csrc")
    inc = os.path.join(root, "include")
    flags = ["-O1", "-g", f"-fsanitize={san}", "-fno-omit-frame-pointer", "-I", inc]
    objs = []
    for src, cc, std in (("oavif_host.c", "gcc", "-std=gnu11"), ("tq.cpp", "g++", "-std=c++17"),
                         ("png_ingest.cpp", "g++", "-std=c++17"),
                         (os.path.join(root, "tests", "c", "stub_scorer.c"), "gcc", "-std=gnu11")):
        path = src if os.path.isabs(src) else os.path.join(csrc, src)
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.run([cc, std, *flags, "-c", path, "-o", obj], check=True, capture_output=True)
        objs.append(obj)
    exe = str(tmp_path / "host_stub")
    subprocess.run(["g++", f"-fsanitize={san}", *objs, "-o", exe, "-ldl", "-lm", "-lz", "-lpthread"], check=True,
                   capture_output=True)
    _ref, png, p = _inputs(tmp_path, 160, 120, seed=5)
    env = dict(ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1", TSAN_OPTIONS="halt_on_error=1",
               OAVIF_HOST_ATEXIT="1", OAVIF_HOST_TIMES="1")
    outs = {}
    for src, args in ((png, ["--score-tgt", "91", "--tolerance", "1", "--max-pass", "8", "--tenbit", "0", "-s", "10"]),
                      (p, ["-t", "85", "--tolerance", "1", "--tenbit", "0", "-s", "10", "--quality-alpha", "60"])):
        for fan in ("1", "4", "16"):
            out = tmp_path / f"o_{src.suffix[1:]}_{fan}.avif"
            r = subprocess.run([exe, *args, str(src), str(out)], capture_output=True, text=True, timeout=600,
                               env=_env(OAVIF_PROBE_FANOUT=fan, **env))
            assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (fan, r.stderr[-3000:])
            assert r.returncode == 0, r.stderr[-1500:]
            lines = [l for l in r.stderr.splitlines() if not l.startswith("  [")]
            found = [l for l in lines if l.startswith("Found q")]
            assert len(found) == 1
            outs[(src, fan)] = (found[0], [l for l in lines if l.startswith("Compressed to")], out.read_bytes())
            if fan != "1":
                assert any(l.startswith("speculative: ") for l in lines)
        assert outs[(src, "1")] == outs[(src, "4")] == outs[(src, "16")]
        assert int(re.search(r"(\d+) passes", outs[(src, "1")][0]).group(1)) >= 2      # a real multi-pass search
