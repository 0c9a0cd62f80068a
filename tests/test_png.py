"""Native PNG ingest (oavif_amd/csrc/png_ingest.cpp through oavif_amd.png) against the output
rules of the reference's loader, /root/reference/src/io.zig:242-307 (libspng, flags 0):
16-bit -> RGBA16 + hbd, 8-bit truecolour -> RGB8, everything else -> RGBA8.  Decode flags are 0 (io.zig:285), so a
tRNS chunk is not applied: files without an alpha channel decode opaque (libspng applies tRNS only under
SPNG_DECODE_TRNS; libspng is not importable here, so that reading is this repo's -- unpinned).

The expected pixels are computed here from the arrays the test files are made of (a small PNG
WRITER below covers every colour type, bit depth, row filter, Adam7 and the ancillary chunks);
files written by Pillow are decoded by both and compared as a second opinion.  CPU only: host code.
"""
import struct
import zlib

import numpy as np
import pytest

from oavif_amd import png


# ---- a PNG writer for the tests ---------------------------------------------------------------------
def _chunk(kind: bytes, data: bytes, bad_crc: bool = False) -> bytes:
    crc = zlib.crc32(kind + data) & 0xFFFFFFFF
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", crc ^ (1 if bad_crc else 0))


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def _filter_row(ftype: int, row: bytes, prev: bytes, bpp: int) -> bytes:
    out = bytearray(len(row))
    for i, v in enumerate(row):
        a = row[i - bpp] if i >= bpp else 0
        b = prev[i]
        c = prev[i - bpp] if i >= bpp else 0
        pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[ftype] if ftype < 5 else 0   # 5+: an invalid filter byte
        out[i] = (v - pred) & 255
    return bytes([ftype]) + bytes(out)


def _pack_rows(samples: np.ndarray, depth: int) -> list:
    """samples: (h, w, s) integers -> list of packed row bytes (big-endian / MSB-first)."""
    h, w, s = samples.shape
    rows = []
    for y in range(h):
        flat = samples[y].reshape(-1)
        if depth == 16:
            rows.append(flat.astype(">u2").tobytes())
        elif depth == 8:
            rows.append(flat.astype(np.uint8).tobytes())
        else:
            bits = "".join(format(int(v), f"0{depth}b") for v in flat)
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def write_png(samples: np.ndarray, ctype: int, depth: int, interlace: bool = False, plte=None, trns=None,
              icc=None, filters=(0, 1, 2, 3, 4), idat_split: int = 0, extra_chunks=(), bad_idat_crc=False) -> bytes:
    h, w, s = samples.shape
    bpp = max(1, s * depth // 8)
    raw = bytearray()
    passes = ADAM7 if interlace else [(0, 0, 1, 1)]
    k = 0
    for x0, y0, dx, dy in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for row in _pack_rows(sub, depth):
            f = filters[k % len(filters)]
            k += 1
            raw += _filter_row(f, row, prev if prev is not None else bytes(len(row)), bpp)
            prev = row
    comp = zlib.compress(bytes(raw), 6)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, int(interlace)))
    if icc is not None:
        out += _chunk(b"iCCP", b"test profile\x00\x00" + zlib.compress(icc))
    if plte is not None:
        out += _chunk(b"PLTE", np.asarray(plte, np.uint8).tobytes())
    if trns is not None:
        out += _chunk(b"tRNS", trns)
    for kind, data in extra_chunks:
        out += _chunk(kind, data)
    pieces = [comp] if not idat_split else [comp[i:i + idat_split] for i in range(0, len(comp), idat_split)]
    for i, piece in enumerate(pieces):
        out += _chunk(b"IDAT", piece, bad_crc=bad_idat_crc and i == 0)
    return out + _chunk(b"IEND", b"")


# ---- the reference's output rules (io.zig:270-290) restated for the expectation ----------------------
def expected(samples, ctype, depth, plte=None, trns=None):
    trns = None  # spng_decode_image(..., flags = 0) at io.zig:285: no SPNG_DECODE_TRNS, the chunk is not applied
    h, w, s = samples.shape
    smp = samples.astype(np.int64)
    if depth == 16:
        out = np.empty((h, w, 4), np.uint16)
        if ctype in (0, 4):
            out[..., :3] = smp[..., :1]
        else:
            out[..., :3] = smp[..., :3]
        if ctype in (4, 6):
            out[..., 3] = smp[..., -1]
        else:
            a = np.full((h, w), 65535)
            if trns is not None:
                key = np.frombuffer(trns, ">u2").astype(np.int64)
                a[np.all(smp[..., :len(key)] == key, axis=-1)] = 0
            out[..., 3] = a
        return out, 4, True
    if ctype == 2:
        return smp.astype(np.uint8), 3, False
    out = np.empty((h, w, 4), np.uint8)
    if ctype == 0:
        out[..., :3] = (smp[..., :1] * (255 // ((1 << depth) - 1))).astype(np.uint8)
        a = np.full((h, w), 255)
        if trns is not None:
            a[smp[..., 0] == (struct.unpack(">H", trns)[0] & ((1 << depth) - 1))] = 0
        out[..., 3] = a
    elif ctype == 3:
        pal = np.asarray(plte, np.uint8).reshape(-1, 3)
        out[..., :3] = pal[smp[..., 0]]
        alpha = np.full(len(pal), 255, np.uint8)
        if trns is not None:
            alpha[:len(trns)] = np.frombuffer(trns, np.uint8)
        out[..., 3] = alpha[smp[..., 0]]
    elif ctype == 4:
        out[..., :3] = smp[..., :1]
        out[..., 3] = smp[..., 1]
    else:
        out[...] = smp
    return out, 4, False


CASES = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16),
         (6, 8), (6, 16)]
NSAMP = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("ctype,depth", CASES)
def test_every_colour_type_depth_filter_and_adam7(hip_lib, ctype, depth, interlace):
    rng = np.random.default_rng(ctype * 100 + depth + int(interlace))
    for (w, h) in ((1, 1), (13, 9), (8, 8), (37, 5), (5, 37)):
        hi = 1 << depth
        plte = trns = None
        if ctype == 3:
            n = min(hi, 256)
            plte = rng.integers(0, 256, (n, 3))
            hi = n
            trns = bytes(rng.integers(0, 256, n // 2 + 1).astype(np.uint8)) if (w + h) % 2 else None
        smp = rng.integers(0, hi, (h, w, NSAMP[ctype]))
        if ctype == 0 and (w * h) % 2:
            trns = struct.pack(">H", int(smp[0, 0, 0]))
        if ctype == 2 and (w * h) % 2:
            trns = struct.pack(">HHH", *[int(v) for v in smp[h // 2, w // 2]])
        data = write_png(smp, ctype, depth, interlace, plte, trns, idat_split=7 if w > 8 else 0)
        exp, ch, hbd = expected(smp, ctype, depth, plte, trns)
        pix, c, hb, icc = png.load_png(data)
        assert (c, hb, icc) == (ch, hbd, None) and pix.dtype == exp.dtype and pix.shape == exp.shape
        assert np.array_equal(pix, exp), (ctype, depth, interlace, w, h)
        info = png.png_info(data)
        assert (info.width, info.height, info.bit_depth, info.color_type, info.interlaced) == (w, h, depth, ctype, int(interlace))


@pytest.mark.parametrize("ctype,depth", [(2, 8), (6, 8), (4, 16), (2, 16), (0, 8)])   # 3-, 4-, 4-, 6- and 1-byte pixels
@pytest.mark.parametrize("ftype", [1, 2, 3, 4])
def test_each_filter_on_every_row_and_strips_of_rows(hip_lib, ctype, depth, ftype):
    """Every row of the file under ONE filter type (the vector forms of Sub / Average / Paeth for 3- and 4-byte
    pixels, the byte-wise forms for the rest), smooth and noisy content, widths around the vector step, and
    images taller than one inflate strip (256 KB) so that the row above a strip's first row is carried over."""
    rng = np.random.default_rng(1000 + ctype * 10 + ftype + depth)
    hi = 1 << depth
    for (w, h) in ((1, 3), (2, 2), (3, 7), (31, 4), (257, 5), (640, 300)):
        noise = rng.integers(0, hi, (h, w, NSAMP[ctype]))
        ramp = ((np.arange(w)[None, :, None] * 3 + np.arange(h)[:, None, None] * 5 + np.arange(NSAMP[ctype])[None, None, :] * 40)
                * (hi // 256)) % hi
        for smp in (noise, ramp, np.where(noise > hi // 2, hi - 1, 0)):     # random, smooth, extremes (ties of Paeth)
            data = write_png(smp, ctype, depth, filters=(ftype,))
            exp, ch, hbd = expected(smp, ctype, depth)
            pix, c, hb, _icc = png.load_png(data)
            assert (c, hb) == (ch, hbd) and np.array_equal(pix, exp), (ctype, depth, ftype, w, h)


def test_icc_profile_is_handed_on_decompressed(hip_lib):
    smp = np.arange(4 * 5 * 3).reshape(5, 4, 3) % 256
    profile = bytes(range(256)) * 5
    pix, c, hbd, icc = png.load_png(write_png(smp, 2, 8, icc=profile))
    assert icc == profile and c == 3
    # ancillary chunks are skipped, also with a bad CRC; a broken iCCP stream only loses the profile
    extra = [(b"tEXt", b"Comment\x00hello"), (b"gAMA", struct.pack(">I", 45455))]
    pix2, *_ = png.load_png(write_png(smp, 2, 8, extra_chunks=extra))
    assert np.array_equal(pix, pix2)
    bad = write_png(smp, 2, 8, icc=profile).replace(zlib.compress(profile)[:8], b"\x00" * 8)
    pix3, _c, _h, icc3 = png.load_png(bad)      # the chunk's CRC no longer matches: discarded, pixels intact
    assert icc3 is None and np.array_equal(pix3, pix)


def test_matches_pillow_on_pillow_written_files(hip_lib, tmp_path):
    """Second opinion: files written by Pillow's own encoder, decoded by Pillow and by this loader."""
    from PIL import Image
    rng = np.random.default_rng(7)
    h, w = 31, 45
    imgs = {
        "RGB": Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)),
        "RGBA": Image.fromarray(rng.integers(0, 256, (h, w, 4), dtype=np.uint8)),
        "L": Image.fromarray(rng.integers(0, 256, (h, w), dtype=np.uint8)),
        "LA": Image.fromarray(rng.integers(0, 256, (h, w, 2), dtype=np.uint8), "LA"),
        "P": Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).quantize(37),
        "1": Image.fromarray(rng.integers(0, 2, (h, w), dtype=np.uint8) * 255).convert("1"),
        "I;16": Image.fromarray(rng.integers(0, 65536, (h, w)).astype(np.uint16)),
    }
    for mode, im in imgs.items():
        p = tmp_path / f"{mode.replace(';', '_')}.png"
        im.save(p, optimize=(mode == "RGB"))
        pix, c, hbd, _ = png.load_png(p.read_bytes())
        ref = Image.open(p)
        if mode == "I;16":
            g = np.asarray(ref).astype(np.uint16)
            assert hbd and c == 4 and np.array_equal(pix[..., 0], g) and np.array_equal(pix[..., 2], g)
            assert (pix[..., 3] == 65535).all()
        elif mode == "RGB":
            assert not hbd and c == 3 and np.array_equal(pix, np.asarray(ref))
        else:
            assert not hbd and c == 4 and np.array_equal(pix, np.asarray(ref.convert("RGBA"))), mode


def test_errors_carry_the_references_names(hip_lib):
    smp = np.arange(6 * 7 * 3).reshape(7, 6, 3) % 256
    good = write_png(smp, 2, 8)

    def err(buf):
        with pytest.raises(png.PngError) as ei:
            png.load_png(buf)
        return ei.value.name
    assert err(b"not a png at all, but long enough to look at........") == "GetHeaderFailed"
    assert err(good[:20]) == "GetHeaderFailed"
    assert err(good[:8] + _chunk(b"IHDR", struct.pack(">IIBBBBB", 0, 7, 8, 2, 0, 0, 0)) + good[33:]) == "GetHeaderFailed"
    assert err(good[:8] + _chunk(b"IHDR", struct.pack(">IIBBBBB", 6, 7, 3, 2, 0, 0, 0)) + good[33:]) == "GetHeaderFailed"
    assert err(write_png(smp, 2, 8, bad_idat_crc=True)) == "DecodeFailed"
    assert err(good[:-20]) == "DecodeFailed"                                   # truncated file
    assert err(write_png(smp, 2, 8, extra_chunks=[(b"ABCD", b"x")])) == "DecodeFailed"   # unknown critical chunk
    assert err(write_png(smp, 2, 8, filters=(5,))) == "DecodeFailed"           # filter type 5
    idx = np.full((3, 3, 1), 9)
    assert err(write_png(idx, 3, 8, plte=np.zeros((4, 3)))) == "DecodeFailed"  # palette index 9 of 4 entries
    assert err(write_png(idx, 3, 8)) == "DecodeFailed"                         # palette image without PLTE
    # a zlib stream that ends before the last scanline
    raw = zlib.compress(bytes(3 * (1 + 6 * 3)))            # three of the seven rows
    assert err(good[:33] + _chunk(b"IDAT", raw) + _chunk(b"IEND", b"")) == "DecodeFailed"


def test_a_header_that_lies_about_its_size_fails_at_once(hip_lib):
    """a 70-byte file whose IHDR claims 60000 x 60000 must not cost 10 GB or 100 s -- neither
    in the info call (callers size their buffer from it) nor in the decode; libspng fails such a file as
    soon as its stream ends."""
    import time
    smp = np.zeros((7, 6, 3), np.int64)
    good = write_png(smp, 2, 8)
    for (w, h, ctype, depth) in ((60000, 60000, 2, 8), (0x7FFFFFFF, 3, 6, 16), (200000, 200000, 0, 1)):
        lie = good[:8] + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + good[33:]
        t = time.perf_counter()
        with pytest.raises(png.PngError) as ei:
            png.png_info(lie)
        assert ei.value.name in ("DecodeFailed", "ImageSizeFailed")
        with pytest.raises(png.PngError):
            png.load_png(lie)
        assert time.perf_counter() - t < 1.0
    # a stream that could hold the claimed image by the 1032 : 1 bound but ends early: fails at the row
    # where it ends, with two rows of memory (the decode inflates scanline by scanline)
    w, h = 20000, 20000
    body = zlib.compress(bytes(1000 * (1 + 3 * w)), 9)           # 1000 of the 20000 rows, ~60 KB
    pad = _chunk(b"IDAT", body) + _chunk(b"IDAT", bytes(1 + (h * (1 + 3 * w)) // 1032 - len(body)))
    lie = good[:8] + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + pad + _chunk(b"IEND", b"")
    assert png.png_info(lie).data_bytes == 3 * w * h
    t = time.perf_counter()
    with pytest.raises(png.PngError) as ei:
        png.load_png(lie)
    assert ei.value.name == "DecodeFailed" and time.perf_counter() - t < 5.0


def test_an_unallocatable_output_is_out_of_memory_not_a_traceback(hip_lib, monkeypatch):
    smp = np.zeros((7, 6, 3), np.int64)
    monkeypatch.setattr(png.np, "empty", lambda *a, **k: (_ for _ in ()).throw(MemoryError()))
    with pytest.raises(png.PngError) as ei:
        png.load_png(write_png(smp, 2, 8))
    assert ei.value.name == "OutOfMemory"


def test_cli_loads_png_through_the_native_decoder(hip_lib, tmp_path, monkeypatch):
    """cli.load_source: .png goes through oavif_png_decode (not Pillow); the scorer's reference is
    Image.toRGB8 of the reference's Image (io.zig:57-133): >> 8 for 16-bit, alpha dropped."""
    from oavif_amd import cli
    rng = np.random.default_rng(3)
    smp = rng.integers(0, 65536, (9, 11, 3))
    p = tmp_path / "deep.png"
    p.write_bytes(write_png(smp, 2, 16, icc=b"PROFILE" * 9))
    import PIL.Image
    monkeypatch.setattr(PIL.Image, "open", lambda *a, **k: (_ for _ in ()).throw(AssertionError("Pillow used for a PNG")))
    s = cli.load_source(str(p))
    assert s.hbd and s.channels == 4 and s.icc == b"PROFILE" * 9
    assert np.array_equal(s.rgb, (smp >> 8).astype(np.uint8))
    assert s.pixels.dtype == np.uint8 and s.pixels.shape == (9, 11, 4) and (s.pixels[..., 3] == 255).all()
    g = rng.integers(0, 4, (6, 5, 1))
    p2 = tmp_path / "gray2.png"
    p2.write_bytes(write_png(g, 0, 2))
    s2 = cli.load_source(str(p2))
    assert not s2.hbd and s2.channels == 4 and np.array_equal(s2.rgb, np.repeat(g * 85, 3, 2).astype(np.uint8))


def test_concurrent_decodes_do_not_share_state(hip_lib):
    """The batch driver loads PNGs from 16 worker threads at once (ctypes drops the GIL); the inflater's tables
    live in thread-local storage.  Eight threads, different files, many rounds: every decode equals the
    single-threaded one."""
    import threading
    rng = np.random.default_rng(77)
    files = []
    for k in range(8):
        w, h = 50 + 37 * k, 40 + 11 * k
        ctype, depth = [(2, 8), (6, 8), (0, 8), (3, 8), (2, 16), (4, 8), (6, 16), (0, 4)][k]
        hi = 1 << depth
        plte = rng.integers(0, 256, (256, 3)) if ctype == 3 else None
        smp = rng.integers(0, min(hi, 256) if ctype == 3 else hi, (h, w, NSAMP[ctype]))
        if k % 2:
            smp = (smp // (hi // 4 or 1)) * (hi // 4 or 1)      # few levels: long matches in the stream
        files.append(write_png(smp, ctype, depth, interlace=bool(k & 1), plte=plte))
    want = [png.load_png(f)[0] for f in files]
    bad = []

    def work(i):
        for _ in range(25):
            if not np.array_equal(png.load_png(files[i])[0], want[i]):
                bad.append(i)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad
