"""CLI surface (parse_args.zig / main.zig) and PAM ingest (io.zig:309-406): CPU tests."""
import numpy as np
import pytest

from oavif_amd import cli, pam, synth


def parse(*argv):
    return cli.parse_args(list(argv))


def test_defaults_match_parse_args_zig():
    o, i, out = parse("in.png", "out.avif")
    assert (i, out) == ("in.png", "out.avif")
    # parse_args.zig:48-63 (code, not README: quality_alpha 0, tenbit on)
    assert (o.quality_alpha, o.speed, o.max_threads, o.tile_rows_log2, o.tile_cols_log2) == (0, 9, 1, 0, 0)
    assert (o.auto_tiling, o.score_tgt, o.tenbit, o.tune, o.tolerance, o.max_pass) == \
        (True, 80.0, True, "iq", 2.0, 6)
    assert o.quality is None
    assert (o.color_primaries, o.transfer_characteristics, o.matrix_coefficients) == (2, 2, 2)


def test_all_flags_and_short_forms():
    o, i, out = parse("-s", "4", "-t", "72.5", "--quality-alpha", "99", "--max-threads", "8",
                      "--tile-rows-log2", "2", "--tile-cols-log2", "6", "--auto-tiling", "0",
                      "--tune", "ssimulacra2", "--tenbit", "0", "--tolerance", "1.5", "--max-pass", "12",
                      "a.jpg", "-q", "100", "--color-primaries", "1", "--transfer-characteristics", "13",
                      "--matrix-coefficients", "0", "b.avif")
    assert (o.speed, o.score_tgt, o.quality_alpha, o.max_threads) == (4, 72.5, 99, 8)
    assert (o.tile_rows_log2, o.tile_cols_log2, o.auto_tiling, o.tune, o.tenbit) == (2, 6, False, "ssimulacra2", False)
    assert (o.tolerance, o.max_pass, o.quality) == (1.5, 12, 100)
    assert (o.color_primaries, o.transfer_characteristics, o.matrix_coefficients) == (1, 13, 0)
    assert (i, out) == ("a.jpg", "b.avif")


@pytest.mark.parametrize("argv,err,msg", [
    (["--speed"], "MissingOptionValue", "Error: Missing --speed value"),
    (["-s", "-1"], "MissingOptionValue", "Error: Missing --speed value"),         # '-' => missing
    (["--speed", "11"], "InvalidOptionValue", "Error: --speed must be between 0 and 10"),
    (["-t", "29.9"], "InvalidOptionValue", "Error: --score-tgt must be between 30 and 100"),
    (["--quality-alpha", "100"], "InvalidOptionValue", "Error: --quality-alpha must be between 0 and 99"),
    (["--max-threads", "0"], "InvalidOptionValue", "Error: --max-threads must be between 1 and 255"),
    (["--tolerance", "0.5"], "InvalidOptionValue", "Error: --tolerance must be between 1 and 100"),
    (["--max-pass", "13"], "InvalidOptionValue", "Error: --max-pass must be between 1 and 12"),
    (["--tenbit", "2"], "InvalidOptionValue", "Error: --tenbit must be 0 or 1"),
    (["--tune", "psnr"], "InvalidOptionValue", "Error: --tune must be one of: ssim, iq, ssimulacra2"),
    (["--matrix-coefficients", "15"], "InvalidOptionValue", "Error: --matrix-coefficients must be between 0 and 14"),
    (["a", "b", "c"], "UnexpectedArgument", "Error: Unexpected argument: c"),
    (["--speed", "x"], "InvalidCharacter", None),
])
def test_error_names_and_messages(argv, err, msg, capsys):
    with pytest.raises(cli.CliError) as ei:
        cli.parse_args(argv)
    assert ei.value.name == err
    if msg:
        assert capsys.readouterr().err.strip().splitlines()[-1] == msg


def test_main_missing_paths_and_help(capsys):
    assert cli.main(["only_input.png"]) == 1
    err = capsys.readouterr().err
    assert err.startswith("\x1b[31moavif\x1b[0m | ") and "error: MissingInputOrOutput" in err
    assert cli.main(["-h", "x", "y"]) == 0
    err = capsys.readouterr().err
    assert "usage:  oavif [options] <in> <out.avif>" in err
    assert "target SSIMULACRA2 score (0..100) [80]" in err
    assert "maximum search passes (1..12) [6]" in err
    assert "Input image formats: PNG, PAM, JPEG, WebP, or AVIF" in err
    # -h is only honoured while it is a leading argument (main.zig:50-59)
    assert cli.main(["in.png", "-h"]) == 1


def test_quality_bypass_writes_avif_without_gpu(tmp_path, capsys):
    """BASELINE configs[0]: -q bypasses the search and the scorer (main.zig:93-100)."""
    from PIL import Image
    from oavif_amd import synth
    if not synth.have_avif():
        pytest.skip("no AVIF codec")
    src = tmp_path / "in.png"
    Image.fromarray(synth.make_ref(64, 48, 1)).save(src)
    out = tmp_path / "out.avif"
    assert cli.main(["-q", "60", str(src), str(out)]) == 0
    err = capsys.readouterr().err.splitlines()
    assert err[1].startswith("Read 64x48, RGB, 8-bit, ")
    assert err[2] == "Encoding [q60, speed 9, 8-bit]"   # Pillow writes 8-bit: the line says what is written
    assert err[3].startswith("Compressed to ") and err[3].endswith(" bpp)")
    assert "passes" not in "\n".join(err)          # measure.py then records passes = None
    assert synth.avif_decode(out.read_bytes()).shape == (48, 64, 3)


# ---- PAM ---------------------------------------------------------------------------------------

def test_pam_roundtrip_all_depths():
    rng = np.random.default_rng(0)
    for c in (1, 2, 3, 4):
        img = rng.integers(0, 256, (5, 7, c), dtype=np.uint8)
        data, w, h, ch = pam.load_pam(pam.write_pam(img))
        assert (w, h, ch) == (7, 5, c) and data == img.tobytes()


def test_pam_header_variants():
    body = bytes(range(12))
    ok = b"P7\n# a comment\nWIDTH 2\r\nHEIGHT   2\nDEPTH\t3\nMAXVAL 255\nTUPLTYPE rgb\nENDHDR\n" + body + b"extra"
    data, w, h, ch = pam.load_pam(ok)
    assert (w, h, ch, data) == (2, 2, 3, body)
    blank = b"P7\nWIDTH 2\nHEIGHT 2\nDEPTH 3\nMAXVAL 255\n\n" + body      # blank line ends the header
    assert pam.load_pam(blank)[0] == body
    unspecified = b"P7\nWIDTH 2\nHEIGHT 2\nDEPTH 3\nMAXVAL 255\nTUPLTYPE FOO\nENDHDR\n" + body
    assert pam.load_pam(unspecified)[3] == 3                                # DEPTH decides


@pytest.mark.parametrize("buf,err", [
    (b"P6\n", "NotAPamFile"),
    (b"P7\nWIDTH 2\nHEIGHT 2", "HeaderNotFound"),
    (b"P7\nWIDTH 2\nHEIGHT 0\nDEPTH 3\nMAXVAL 255\nENDHDR\n", "InvalidPamDimensions"),
    (b"P7\nWIDTH 2\nHEIGHT 2\nDEPTH 3\nENDHDR\n", "InvalidPamDimensions"),
    (b"P7\nWIDTH 2\nHEIGHT 2\nDEPTH 3\nMAXVAL 65535\nENDHDR\n" + bytes(24), "UnsupportedPamMaxVal"),
    (b"P7\nWIDTH 2\nHEIGHT 2\nDEPTH 5\nMAXVAL 255\nENDHDR\n" + bytes(20), "UnsupportedPamDepth"),
    (b"P7\nWIDTH 2\nHEIGHT 2\nDEPTH 4\nMAXVAL 255\nTUPLTYPE RGB\nENDHDR\n" + bytes(16), "PamTupleMismatch"),
    (b"P7\nWIDTH 2\nHEIGHT 2\nDEPTH 1\nMAXVAL 255\nTUPLTYPE BLACKANDWHITE\nENDHDR\n" + bytes(4), "UnsupportedPamTuple"),
    (b"P7\nWIDTH 2\nHEIGHT 2\nDEPTH 3\nMAXVAL 255\nENDHDR\n" + bytes(11), "InsufficientDataInFile"),
    (b"P7\nWIDTH two\nHEIGHT 2\nDEPTH 3\nMAXVAL 255\nENDHDR\n" + bytes(12), "InvalidCharacter"),
])
def test_pam_rejections(buf, err):
    with pytest.raises(pam.PamError) as ei:
        pam.load_pam(buf)
    assert ei.value.name == err


def test_to_rgb8_rules(tmp_path):
    """Image.toRGB8 (io.zig:57-133): alpha dropped, gray replicated."""
    rng = np.random.default_rng(1)
    for c in (1, 2, 3, 4):
        img = rng.integers(0, 256, (6, 4, c), dtype=np.uint8)
        p = tmp_path / f"x{c}.pam"
        p.write_bytes(pam.write_pam(img))
        rgb, src, ch, hbd = cli.load_image(str(p))
        assert ch == c and not hbd and rgb.shape == (6, 4, 3)
        exp = np.repeat(img[..., :1], 3, 2) if c < 3 else img[..., :3]
        assert np.array_equal(rgb, exp)
    with pytest.raises(cli.CliError) as ei:
        cli.load_image("x.bmp")
    assert ei.value.name == "UnsupportedImageFormat"


@pytest.mark.skipif(not synth.have_avif(), reason="Pillow AVIF codec not available")
def test_icc_profile_passes_through_the_quality_bypass(tmp_path, capsys):
    """io.zig:556-560: the source's ICC profile is handed to the encoder unchanged (checked on the
    -q path, which needs no scorer)."""
    from PIL import Image, ImageCms
    icc = ImageCms.ImageCmsProfile(ImageCms.createProfile("sRGB")).tobytes()
    src = tmp_path / "in.png"
    Image.fromarray(synth.make_ref(64, 48, 5)).save(src, icc_profile=icc)
    out = tmp_path / "out.avif"
    assert cli.main(["-q", "70", str(src), str(out)]) == 0
    capsys.readouterr()
    assert Image.open(out).info.get("icc_profile") == icc
    plain = tmp_path / "plain.png"
    Image.fromarray(synth.make_ref(64, 48, 5)).save(plain)
    assert cli.main(["-q", "70", str(plain), str(out)]) == 0
    capsys.readouterr()
    assert not Image.open(out).info.get("icc_profile")


def test_early_error_exit_after_prefetch_does_not_hang(tmp_path):
    """cli.main starts the scorer's background initialisation (ssimu2_prefetch) before it loads
    the input; an early error exit must join it (ssimu2_prefetch_join) and return 1 promptly --
    with or without a GPU."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "oavif_amd.cli", str(tmp_path / "missing.png"), str(tmp_path / "o.avif")],
                       cwd=root, capture_output=True, text=True, timeout=120)
    assert p.returncode == 1
    assert "error: FileNotFound" in p.stderr


def test_blur_mode_selector_from_environment(monkeypatch):
    """OAVIF_SSIMU2_BLUR (host side only; the C library reads no environment): fir / recursive /
    recursive_fma map to ssimu2_ctx_set_blur's modes, anything else is refused.  Unset = the search
    path's default since round 4: the published recursion."""
    from oavif_amd import _lib, cli
    monkeypatch.delenv("OAVIF_SSIMU2_BLUR", raising=False)
    assert cli.blur_from_env() == _lib.BLUR_RECURSIVE
    for text, mode in (("fir", None), ("recursive", _lib.BLUR_RECURSIVE), ("IIR", _lib.BLUR_RECURSIVE),
                       ("recursive_fma", _lib.BLUR_RECURSIVE_FMA), (" recursive-fma ", _lib.BLUR_RECURSIVE_FMA)):
        monkeypatch.setenv("OAVIF_SSIMU2_BLUR", text)
        assert cli.blur_from_env() == mode
    monkeypatch.setenv("OAVIF_SSIMU2_BLUR", "gaussian")
    with pytest.raises(cli.CliError):
        cli.blur_from_env()
    assert (_lib.BLUR_FIR, _lib.BLUR_RECURSIVE, _lib.BLUR_RECURSIVE_FMA) == (0, 1, 2)   # include/ssimu2_hip.h
