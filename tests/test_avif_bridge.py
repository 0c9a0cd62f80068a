"""oavif_amd.avif_bridge: libavif's C API called the way io.zig:452-482,544-666 calls it.

Rows A10 / A11 of SURVEY.md section 8 stay on the CPU and stay libavif's; what is checked here is that the
bridge makes the reference's calls (YUV444, CICP, ICC, qualityAlpha, tune, forced 8-bit RGB decode) on the
library that is present, that its struct-layout guard works, and what that library cannot do (10-bit).
Pillow's own plugin -- another binding of the same libavif -- is the independent cross-check of the pixels.
No GPU."""
import io
import os
import threading

import numpy as np
import pytest

from oavif_amd import avif_bridge as ab
from oavif_amd import cli, synth

pytestmark = pytest.mark.skipif(not ab.available(), reason=f"libavif bridge unavailable: {ab.why_unavailable()}")


def _opts(**kw):
    o = cli.AvifEncOptions()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _pillow_decode(data: bytes, mode="RGB"):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert(mode))


def test_layout_guard_and_versions():
    assert ab.why_unavailable() is None
    v = ab.versions()
    assert v.startswith("libavif 1.") and "aom" in v and "dav1d" in v
    assert ab.supports_depth(8)


def test_layout_guard_refuses_a_library_that_reports_another_layout(monkeypatch):
    """The guard reads the documented defaults back through the offsets the module uses: shift one and the
    check must fail with a message, not encode garbage."""
    L = ab._lib()
    monkeypatch.setattr(ab, "_DEC_IMAGE", 40)       # strictFlags' slot: non-zero after avifDecoderCreate
    assert "avifDecoder" in ab._check_layout(L)
    monkeypatch.undo()
    monkeypatch.setattr(ab, "_IMG_CP", 112)
    assert "avifImage" in ab._check_layout(L)
    monkeypatch.undo()
    monkeypatch.setattr(ab, "_RGB_PIXELS", 40)
    assert "avifRGBImage" in ab._check_layout(L)
    monkeypatch.undo()
    assert ab._check_layout(L) is None


def test_round_trip_matches_pillows_binding_of_the_same_library():
    ref = synth.make_ref(200, 136, 5)       # odd-ish size: rows of libavif's buffer need not be tight
    o = _opts(tune="ssim")
    prev = None
    for q in (30, 60, 90):
        data = ab.encode(ref, 8, o, q)
        dec = ab.decode_rgb8(data)
        assert dec.shape == ref.shape and dec.dtype == np.uint8 and dec.flags.c_contiguous
        assert np.array_equal(dec, _pillow_decode(data))                    # same bitstream, two bindings
        err = np.abs(dec.astype(int) - ref.astype(int)).mean()
        assert err < 6 and (prev is None or err < prev)                     # quality is honoured
        prev = err
        # Pillow's encode with the options it can express (ssim, 4:4:4) decodes identically through the bridge
        pd = synth.avif_encode(ref, q, speed=o.speed, max_threads=1)
        assert np.array_equal(ab.decode_rgb8(pd), _pillow_decode(pd))
    info = ab.probe(data)
    assert info == {"width": 200, "height": 136, "depth": 8, "alpha": False}


def test_encoder_options_of_copy_to_encoder_reach_libaom():
    ref = synth.make_ref(160, 120, 6)
    sizes = {t: len(ab.encode(ref, 8, _opts(tune=t), 60)) for t in cli.TUNE_MODES}   # parse_args.zig:26-45
    assert len(set(sizes.values())) == 3, sizes            # three tunings, three bitstreams (`iq` is the default)
    with pytest.raises(ab.AvifBridgeError) as e:
        ab.encode(ref, 8, _opts(tune="nonsense"), 60)
    assert e.value.name in ("InvalidCodecOption", "AddImageFailed")   # libavif validates the option when it encodes
    s1 = ab.encode(ref, 8, _opts(speed=10), 60)
    s2 = ab.encode(ref, 8, _opts(speed=6), 60)
    assert s1 != s2
    assert ab.encode(ref, 8, _opts(), 60) == ab.encode(ref, 8, _opts(), 60)   # deterministic


def test_cicp_and_icc_are_written():
    ref = synth.make_ref(64, 48, 7)
    data = ab.encode(ref, 8, _opts(color_primaries=1, transfer_characteristics=13, matrix_coefficients=6), 60)
    at = data.index(b"nclx")
    assert data[at + 4:at + 10] == bytes([0, 1, 0, 13, 0, 6])                # colr box: three big-endian u16
    at = ab.encode(ref, 8, _opts(), 60)
    assert at[at.index(b"nclx") + 4:at.index(b"nclx") + 10] == bytes([0, 2, 0, 2, 0, 2])   # parse_args.zig:61-63
    from PIL import Image, ImageCms
    icc = ImageCms.ImageCmsProfile(ImageCms.createProfile("sRGB")).tobytes()
    with_icc = ab.encode(ref, 8, _opts(), 60, icc=icc)
    assert Image.open(io.BytesIO(with_icc)).info.get("icc_profile") == icc    # io.zig:556-560


def test_alpha_plane_and_quality_alpha():
    ref = synth.make_ref(96, 64, 8)
    alpha = np.tile(np.linspace(0, 255, 96, dtype=np.uint8), (64, 1))
    rgba = np.dstack([ref, alpha])
    lo = ab.encode(rgba, 8, _opts(quality_alpha=0), 60)
    hi = ab.encode(rgba, 8, _opts(quality_alpha=99), 60)
    assert len(hi) > len(lo)                                                  # io.zig:626
    assert ab.probe(hi)["alpha"] is True
    with ab.decode_common(hi) as f:                                           # io.zig:473: RGBA iff alpha plane
        assert (f.channels, f.rows.shape[0], f.row_bytes >= 96 * 4) == (4, 64, True)
        tight = f.tight_rgb8()
        a = f.rows[:, :96 * 4].reshape(64, 96, 4)[..., 3]
        assert np.abs(a.astype(int) - alpha).max() <= 3
    assert tight.shape == (64, 96, 3)                                         # io.zig:654-663: alpha dropped
    assert np.array_equal(tight, _pillow_decode(hi, "RGBA")[..., :3])
    assert tight.base is None or tight.flags.owndata                          # a copy: survives close()


def test_ten_bit_is_what_the_reference_asks_for_and_what_this_libaom_refuses():
    """io.zig:546: 10-bit when --tenbit 1 (the default) or the source is 16-bit.  The libaom inside Pillow's
    libavif is built without high-bit-depth support, so the encode fails in aom_codec_enc_init -- recorded
    here as the fact it is; on a library that can, the same call must produce a 10-bit stream."""
    assert ab.output_depth(True, False) == 10 and ab.output_depth(False, True) == 10
    assert ab.output_depth(False, False) == 8
    ref = synth.make_ref(64, 48, 9)
    scaled = ab.prescale_source(ref, 10)
    assert scaled.dtype == np.uint16 and int(scaled.max()) <= 1023
    assert np.array_equal(scaled, (ref.astype(np.uint32) * 1023 + 127) // 255)        # io.zig:572
    if ab.supports_depth(10):
        data = ab.encode(scaled, 10, _opts(), 60)
        assert ab.probe(data)["depth"] == 10
        assert ab.decode_rgb8(data).dtype == np.uint8                                   # io.zig:470-471
        assert cli.codec_depth(True, False) == (10, None)
    else:
        with pytest.raises(ab.AvifBridgeError) as e:
            ab.encode(scaled, 10, _opts(), 60)
        assert e.value.name == "AddImageFailed"
        depth, note = cli.codec_depth(True, False)
        assert depth == 8 and "high-bit-depth" in note and "io.zig:546" in note
    assert cli.codec_depth(False, False) == (8, None)


def test_bad_arguments_fail_with_the_references_error_names():
    ref = synth.make_ref(32, 32, 1)
    with pytest.raises(ab.AvifBridgeError) as e:
        ab.encode(ref[..., :1], 8, _opts(), 60)
    assert e.value.name == "ConvertFailed"
    with pytest.raises(ab.AvifBridgeError):
        ab.encode(ref.astype(np.uint16), 8, _opts(), 60)
    with pytest.raises(ab.AvifBridgeError) as e:
        ab.decode_rgb8(b"not an avif file at all")
    assert e.value.name == "ParseFailed"
    good = ab.encode(ref, 8, _opts(), 60)
    with pytest.raises(ab.AvifBridgeError) as e:
        ab.decode_rgb8(good[: len(good) // 2])
    assert e.value.name in ("ParseFailed", "DecodeImageFailed")


def test_concurrent_encodes_and_decodes_are_independent():
    """The batch driver and the speculative search run the codec from several threads (ctypes drops the GIL)."""
    ref = synth.make_ref(128, 96, 11)
    qs = [35, 50, 65, 80]
    want = {q: ab.encode(ref, 8, _opts(), q) for q in qs}
    got, errs = {}, []

    def work(q):
        try:
            for _ in range(3):
                d = ab.encode(ref, 8, _opts(), q)
                got[q] = (d, ab.decode_rgb8(d))
        except Exception as ex:   # pragma: no cover
            errs.append(ex)
    ts = [threading.Thread(target=work, args=(q,)) for q in qs]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    for q in qs:
        assert got[q][0] == want[q] and np.array_equal(got[q][1], ab.decode_rgb8(want[q]))


def test_cli_uses_the_bridge_and_falls_back_to_pillow_when_told(tmp_path, monkeypatch, capsys):
    from PIL import Image
    ref = synth.make_ref(64, 48, 12)
    src = tmp_path / "in.png"
    Image.fromarray(ref).save(src)
    assert cli.main(["-q", "61", "--tenbit", "0", str(src), str(tmp_path / "a.avif")]) == 0
    err = capsys.readouterr().err
    assert "Encoding [q61, speed 9, 8-bit]" in err and "note:" not in err
    assert (tmp_path / "a.avif").read_bytes() == ab.encode(ref, 8, _opts(tenbit=False), 61)
    # gray sources are expanded, not handed over with a wrong row stride (io.zig:564)
    Image.fromarray(ref[..., 0]).save(tmp_path / "g.png")
    assert cli.main(["-q", "61", "--tenbit", "0", str(tmp_path / "g.png"), str(tmp_path / "g.avif")]) == 0
    assert ab.probe((tmp_path / "g.avif").read_bytes())["width"] == 64
    capsys.readouterr()
    monkeypatch.setenv("OAVIF_CODEC", "pillow")
    assert not ab.available() and ab.why_unavailable() == "OAVIF_CODEC=pillow"
    assert cli.main(["-q", "61", str(src), str(tmp_path / "p.avif")]) == 0
    assert "bridge is off (OAVIF_CODEC=pillow)" in capsys.readouterr().err
    assert _pillow_decode((tmp_path / "p.avif").read_bytes()).shape == (48, 64, 3)


def test_search_with_decoded_frame_hand_off_equals_the_tight_copy_path(monkeypatch):
    """tq.search_hip_frames hands libavif's rows (RGBA, padded or not) to the scorer; the frames it hands over
    unpack to exactly what io.decodeAvifToRgb's copy loop produces, and every frame is closed."""
    from oavif_amd import tq
    ref = synth.make_ref(96, 64, 13)
    rgba = np.dstack([ref, np.full(ref.shape[:2], 180, np.uint8)])
    o = _opts(tenbit=False)
    seen = []

    class FakeScorer:
        def set_reference(self, r):
            assert np.array_equal(r, ref)

        def score_decoded_against_reference(self, flat, row_bytes, channels):
            a = np.asarray(flat).reshape(64, row_bytes)[:, :96 * channels].reshape(64, 96, channels)[..., :3]
            seen.append(a.copy())
            return 100.0 - float(np.abs(a.astype(int) - ref).mean()) * 4

    def codec_frame(q):
        data = ab.encode(rgba, 8, o, q)
        f = ab.decode_common(data)
        frames.append(f)
        return f, len(data)
    frames = []
    r = tq.search_hip_frames(FakeScorer(), ref, codec_frame, score_tgt=92.0, tolerance=1.0, max_pass=4)
    assert r.num_pass == len(seen) == len(frames) >= 2
    assert all(f._rgb is None and f.rows is None for f in frames)          # every frame was closed after its score
    for (q, _s), a in zip(r.history, seen):
        assert np.array_equal(a, ab.decode_rgb8(ab.encode(rgba, 8, o, int(q))))


@pytest.mark.gpu
@pytest.mark.parametrize("alpha", [False, True])
def test_hand_off_search_equals_tight_copy_search_on_the_device(scorer, alpha):
    """tq.search_hip_frames (libavif's RGB / RGBA rows to ssimu2_score_against_reference_strided) against
    tq.search_hip (io.decodeAvifToRgb's tight copy to ssimu2_score_against_reference): the same probes with
    the same score bits, the same quantizer."""
    from oavif_amd import tq
    ref = synth.make_ref(333, 211, 21)          # rows of 999 / 1332 bytes: not multiples of anything
    src = np.dstack([ref, np.tile(np.linspace(30, 255, 333, dtype=np.uint8), (211, 1))]) if alpha else ref
    o = _opts(tenbit=False, quality_alpha=85)
    data = {}

    def enc(q):
        if q not in data:
            data[q] = ab.encode(src, 8, o, q)
        return data[q]
    for tgt in (70.0, 84.0, 93.0):
        a = tq.search_hip(scorer, ref, lambda q: (ab.decode_rgb8(enc(q)), len(enc(q))), score_tgt=tgt, tolerance=1.0)
        b = tq.search_hip_frames(scorer, ref, lambda q: (ab.decode_common(enc(q)), len(enc(q))), score_tgt=tgt,
                                 tolerance=1.0)
        assert (a.q, a.score, a.num_pass, a.buf_q, a.history) == (b.q, b.score, b.num_pass, b.buf_q, b.history)
        assert a.last_avif_size == b.last_avif_size == len(enc(a.buf_q))


def test_encoder_source_hoists_the_yuv_conversion_without_changing_a_byte():
    """EncoderSource = io.zig:550-623 done once per image (avifImageCreate, CICP, ICC, avifImageRGBToYUV); every
    encode from it equals the one-call form that rebuilds the image per pass, as the reference does -- also from
    several threads at once (the probes of a speculative search share one source)."""
    ref = synth.make_ref(160, 120, 14)
    rgba = np.dstack([ref, np.tile(np.linspace(0, 255, 160, dtype=np.uint8), (120, 1))])
    o = _opts(quality_alpha=70, color_primaries=1)
    qs = [20, 45, 70, 95]
    want = {q: ab.encode(rgba, 8, o, q) for q in qs}
    with ab.EncoderSource(rgba, 8, o) as src:
        assert (src.width, src.height, src.channels, src.depth) == (160, 120, 4, 8)
        for q in qs + qs[::-1]:
            assert src.encode(o, q) == want[q]
        got = {}
        ts = [threading.Thread(target=lambda q=q: got.__setitem__(q, src.encode(o, q))) for q in qs]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert got == want
    with pytest.raises(ab.AvifBridgeError):
        src.encode(o, 50)                               # closed
    prepared = cli.encoder_input(rgba[..., :1], _opts(tenbit=False), None)   # gray -> RGB, made once
    assert (prepared.channels, prepared.depth) == (3, 8)
    assert cli._encode(None, _opts(tenbit=False), 50, icc=None, prepared=prepared) == \
        ab.encode(np.repeat(rgba[..., :1], 3, axis=2), 8, _opts(tenbit=False), 50)
    prepared.close()


def test_a_missing_or_foreign_library_turns_the_bridge_off_with_a_reason(tmp_path):
    """OAVIF_LIBAVIF names the library; one that cannot be opened, or that is not libavif, leaves available()
    False with the reason, and the CLI mirror then encodes through Pillow's plugin and says so."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from oavif_amd import avif_bridge as ab, cli; print(ab.available()); print(ab.why_unavailable()); "
            "print(cli.codec_depth(True, False)[1])")
    for lib, needle in (("/nonexistent/libavif.so", "cannot open shared object"), ("libz.so.1", "undefined symbol")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root,
                           env=dict(os.environ, OAVIF_LIBAVIF=lib, PYTHONPATH=root))
        out = r.stdout.splitlines()
        assert r.returncode == 0 and out[0] == "False", r.stderr
        assert needle in out[1] and "bridge is off" in out[2], out
