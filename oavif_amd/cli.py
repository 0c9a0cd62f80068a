"""`oavif` command-line surface over the MI355X scorer.

    python -m oavif_amd.cli [options] <in> <out.avif>

Mirrors /root/reference/src/main.zig (driver flow, stderr line formats) and
/root/reference/src/parse_args.zig (flags, ranges, defaults, error messages) so that tools
written against the reference -- scripts/measure.py parses "N passes" from stderr
(measure.py:27) -- keep working.  The code's behaviour wins where the reference's README
disagrees with it: --quality-alpha default 0 / max 99, --score-tgt min 30.

What runs where: argument parsing and the search control are this repo's (the search is
oavif_amd.tq over the C ABI); the SSIMULACRA2 score of every pass runs on the GPU; the AVIF
encode / decode stay on the CPU in libavif (aom / dav1d), called through oavif_amd.avif_bridge
with the reference's own sequence of libavif calls and arguments (io.zig:544-666: YUV444, CICP,
ICC, qualityAlpha, `tune`, tiling; decode forced to 8-bit RGB).  The library is the libavif
1.4.1 that Pillow bundles (the image has no libavif headers or development package).  Its libaom
is built WITHOUT high-bit-depth support, so no 10-bit AVIF can be written on this image:
`--tenbit 1` (the default) is accepted and reported, the bitstream is 8-bit, and the run says
so; with an aom that has it (OAVIF_LIBAVIF=/path/to/libavif.so) the same code writes 10-bit.
The scorer input is 8-bit RGB either way (io.zig:470-471).  OAVIF_CODEC=pillow (or a libavif
whose struct layout fails the bridge's check) falls back to Pillow's plugin: 8-bit, no
`tune=iq`, no CICP.
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass
from typing import List, Optional, Tuple

VERSION = "oavif_amd-0.1"


class CliError(Exception):
    """Carries the reference's Zig error name (e.g. MissingOptionValue)."""

    def __init__(self, name: str):
        super().__init__(name)
        self.name = name


def blur_from_env():
    """The blur of the SEARCH PATH (this CLI, the batch driver; the Zig shim's `blur` says the same):
    the published recursive Gaussian, `recursive`, unless OAVIF_SSIMU2_BLUR says `fir` or `recursive_fma`.

    Why the recursion is the default here since round 4: fssimu2's source is not available, so which fp32
    evaluation of the blur it uses is unknown (parity unpinned); what IS known is that the published
    SSIMULACRA2 code blurs recursively, that the FIR form differs from it by a median 0.5 / up to 2.4
    points at 3840x2160 and ends 8 of 24 4K searches on another quantizer (profiles/
    r04_4k_search_both_modes.json), and that following the published arithmetic costs 0.38 instead of
    0.16 ms of a ~140 ms pass.  With the cost at 0.2 % the search follows the published order; `fir` stays
    the throughput mode (what bench.py's `value` measures, what a context defaults to at the C ABI).
    tests/golden/pin_kit + scripts/pin_blur_mode.py settle the question for anyone who can run fssimu2.
    Read by the host side only: the C library itself reads no environment."""
    from . import _lib
    v = os.environ.get("OAVIF_SSIMU2_BLUR", "").strip().lower()
    if v in ("", "recursive", "iir"):
        return _lib.BLUR_RECURSIVE
    if v == "fir":
        return None
    if v in ("recursive_fma", "recursive-fma", "iir_fma"):
        return _lib.BLUR_RECURSIVE_FMA
    raise CliError(f"OAVIF_SSIMU2_BLUR={v!r}: expected 'fir', 'recursive' or 'recursive_fma'")


def eprint(s: str = "", end: str = "\n") -> None:
    sys.stderr.write(s + end)


@dataclass
class AvifEncOptions:  # parse_args.zig:48-63
    quality_alpha: int = 0
    speed: int = 9
    max_threads: int = 1
    tile_rows_log2: int = 0
    tile_cols_log2: int = 0
    auto_tiling: bool = True
    score_tgt: float = 80.0
    tenbit: bool = True
    tune: str = "iq"
    tolerance: float = 2.0
    max_pass: int = 6
    quality: Optional[int] = None
    color_primaries: int = 2
    transfer_characteristics: int = 2
    matrix_coefficients: int = 2


TUNE_MODES = ("ssim", "iq", "ssimulacra2")  # parse_args.zig:26-45


def _fmt_num(v) -> str:
    """Zig's {d} on an f64 bound prints 30 for 30.0."""
    return str(int(v)) if float(v).is_integer() else str(v)


def _value(args: List[str], i: int, name: str) -> str:
    # parse_args.zig:126,140,154,168: a value that starts with '-' counts as missing
    if i >= len(args) or args[i].startswith("-"):
        eprint(f"Error: Missing {name} value")
        raise CliError("MissingOptionValue")
    return args[i]


def _int_arg(args, i, lo, hi, name) -> int:
    s = _value(args, i, name)
    try:
        v = int(s, 10)
    except ValueError:
        raise CliError("InvalidCharacter")
    if v < lo or v > hi:
        eprint(f"Error: {name} must be between {lo} and {hi}")
        raise CliError("InvalidOptionValue")
    return v


def _float_arg(args, i, lo, hi, name) -> float:
    s = _value(args, i, name)
    try:
        v = float(s)
    except ValueError:
        raise CliError("InvalidCharacter")
    if v < lo or v > hi:
        eprint(f"Error: {name} must be between {_fmt_num(lo)} and {_fmt_num(hi)}")
        raise CliError("InvalidOptionValue")
    return v


def _bool_arg(args, i, name) -> bool:
    s = _value(args, i, name)
    try:
        v = int(s, 10)
    except ValueError:
        raise CliError("InvalidCharacter")
    if v not in (0, 1):
        eprint(f"Error: {name} must be 0 or 1")
        raise CliError("InvalidOptionValue")
    return v == 1


def parse_args(argv: List[str]) -> Tuple[AvifEncOptions, Optional[str], Optional[str]]:
    """parse_args.zig:76-122.  argv excludes the program name."""
    o = AvifEncOptions()
    inp = out = None
    i = 0
    while i < len(argv):
        a = argv[i]
        i += 1
        if a in ("-s", "--speed"):
            o.speed = _int_arg(argv, i, 0, 10, "--speed"); i += 1
        elif a in ("-t", "--score-tgt"):
            o.score_tgt = _float_arg(argv, i, 30.0, 100.0, "--score-tgt"); i += 1
        elif a == "--quality-alpha":
            o.quality_alpha = _int_arg(argv, i, 0, 99, a); i += 1
        elif a == "--max-threads":
            o.max_threads = _int_arg(argv, i, 1, 255, a); i += 1
        elif a == "--tile-rows-log2":
            o.tile_rows_log2 = _int_arg(argv, i, 0, 6, a); i += 1
        elif a == "--tile-cols-log2":
            o.tile_cols_log2 = _int_arg(argv, i, 0, 6, a); i += 1
        elif a == "--auto-tiling":
            o.auto_tiling = _bool_arg(argv, i, a); i += 1
        elif a == "--tune":
            s = _value(argv, i, a)
            if s not in TUNE_MODES:
                eprint(f"Error: {a} must be one of: ssim, iq, ssimulacra2")
                raise CliError("InvalidOptionValue")
            o.tune = s; i += 1
        elif a == "--tenbit":
            o.tenbit = _bool_arg(argv, i, a); i += 1
        elif a == "--tolerance":
            o.tolerance = _float_arg(argv, i, 1.0, 100.0, a); i += 1
        elif a == "--max-pass":
            o.max_pass = _int_arg(argv, i, 1, 12, a); i += 1
        elif a in ("-q", "--quality"):
            o.quality = _int_arg(argv, i, 0, 100, "--quality"); i += 1
        elif a == "--color-primaries":
            o.color_primaries = _int_arg(argv, i, 1, 22, a); i += 1
        elif a == "--transfer-characteristics":
            o.transfer_characteristics = _int_arg(argv, i, 1, 18, a); i += 1
        elif a == "--matrix-coefficients":
            o.matrix_coefficients = _int_arg(argv, i, 0, 14, a); i += 1
        elif inp is None:
            inp = a
        elif out is None:
            out = a
        else:
            eprint(f"Error: Unexpected argument: {a}")
            raise CliError("UnexpectedArgument")
    return o, inp, out


def print_usage() -> None:  # parse_args.zig:180-238
    d = AvifEncOptions()
    eprint()
    eprint(f"""usage:  oavif [options] <in> <out.avif>

options:
 -h, --help
    show this help
 -v, --version
    show version information
 -s, --speed u8
    encoder speed (0..10) [{d.speed}]
 -t, --score-tgt f64
    target SSIMULACRA2 score (0..100) [{d.score_tgt:.0f}]
 --quality-alpha u8
    quality factor for alpha (0..100=lossless) [{d.quality_alpha}]
 --max-threads u8
    maximum number of threads to use (1..255) [{d.max_threads}]
 --tile-rows-log2 u8
    tile rows log2 (0..6) [{d.tile_rows_log2}]
 --tile-cols-log2 u8
    tile columns log2 (0..6) [{d.tile_cols_log2}]
 --auto-tiling 0/1
    enable automatic tiling [{int(d.auto_tiling)}]
 --tune str
    libaom tuning mode (ssim, iq, ssimulacra2) [{d.tune}]
 --tenbit 0/1
    force 10-bit AVIF output [{int(d.tenbit)}]
 --tolerance f64
    target quality error tolerance (1..100) [{d.tolerance:.0f}]
 --max-pass u8
    maximum search passes (1..12) [{d.max_pass}]
 -q, --quality u8
    quantizer (0..100), bypasses search
 --color-primaries u8
    color primaries (1..22) [{d.color_primaries}]
 --transfer-characteristics u8
    transfer characteristics (1..18) [{d.transfer_characteristics}]
 --matrix-coefficients u8
    matrix coefficients (0..14) [{d.matrix_coefficients}]""", end="")
    eprint("\n\n\x1b[37mInput image formats: PNG, PAM, JPEG, WebP, or AVIF\x1b[0m")


def _bridge_on() -> bool:
    from . import avif_bridge
    return avif_bridge.available()


def print_version() -> None:
    import oavif_amd
    from . import avif_bridge
    eprint(f"oavif {VERSION}")
    eprint(f"scorer {oavif_amd.version()}")
    if avif_bridge.available():
        eprint(f"{avif_bridge.versions()} (C API through oavif_amd.avif_bridge; 10-bit encode: "
               f"{'yes' if avif_bridge.supports_depth(10) else 'no, aom built without high bit depth'})")
    else:
        from PIL import features
        eprint(f"libavif {features.version('avif')} (via Pillow's plugin: {avif_bridge.why_unavailable()})")


# ---- image I/O (CPU; counterpart of io.zig) -------------------------------------------------------

class Source:
    """One loaded input image (io.zig's `Image`, io.zig:42-55): the pixels as decoded
    (`pixels`: (h, w, channels) u8, what the encoder gets, alpha included), the scorer's
    reference `rgb` ((h, w, 3) u8, Image.toRGB8), and the ICC profile to pass through."""
    __slots__ = ("rgb", "pixels", "channels", "hbd", "icc")

    def __init__(self, rgb, pixels, channels, hbd, icc):
        self.rgb, self.pixels, self.channels, self.hbd, self.icc = rgb, pixels, channels, hbd, icc


def load_source(path: str) -> Source:
    """io.loadImage + Image.toRGB8 (io.zig:57-150).  16-bit sources are truncated with >> 8,
    alpha is dropped from the scorer's reference (but kept in `pixels`), gray is replicated."""
    import numpy as np
    from PIL import Image
    ext = os.path.splitext(path)[1].lower()
    if ext not in (".jpg", ".jpeg", ".png", ".pam", ".webp", ".avif"):
        raise CliError("UnsupportedImageFormat")
    icc = None
    if ext == ".pam":
        from .pam import load_pam
        data, w, h, ch = load_pam(open(path, "rb").read())
        arr = np.frombuffer(data, np.uint8).reshape(h, w, ch)
        hbd = False
    elif ext == ".png":
        # io.loadPNG (io.zig:242-307) through the native decoder of the C ABI (oavif_png_decode):
        # 16-bit -> RGBA16 + hbd, 8-bit truecolour -> RGB8, everything else -> RGBA8
        from .png import PngError, load_png
        try:
            arr, _ch, hbd, icc = load_png(open(path, "rb").read())
        except PngError as e:
            raise CliError(e.name)
        if hbd and codec_depth(True, True)[0] != 10:
            # an encoder that cannot write 10-bit gets what Image.toRGB8 makes of 16 bits (>> 8, io.zig:602);
            # one that can keeps the u16 samples and encodes `>> 6` (io.zig:587)
            arr = (arr >> 8).astype(np.uint8)
    else:
        im = Image.open(path)
        hbd = im.mode in ("I;16", "I;16B", "I;16L", "I")
        if hbd:
            arr = (np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)[..., None]
        else:
            if im.mode not in ("L", "LA", "RGB", "RGBA"):
                im = im.convert("RGBA" if "A" in im.getbands() or "transparency" in im.info else "RGB")
            arr = np.asarray(im)
            if arr.ndim == 2:
                arr = arr[..., None]
        icc = im.info.get("icc_profile")  # Image.icc (io.zig:48): handed to the encoder unchanged (io.zig:556-560)
    ch = arr.shape[2]
    if ch == 1 or ch == 2:
        rgb = np.repeat(arr[..., :1], 3, axis=2)
    else:
        rgb = arr[..., :3]
    if rgb.dtype == np.uint16:   # Image.toRGB8: 16-bit -> >> 8 truncation (io.zig:63-95)
        rgb = (rgb >> 8).astype(np.uint8)
    return Source(np.ascontiguousarray(rgb), arr, ch, hbd, icc)


def load_image(path: str):
    """-> (rgb8 (h,w,3) u8, source pixels, channels, hbd); the ICC profile of the last image
    loaded this way is what `_encode` passes through when none is given (the CLI handles one
    image per process; the batch driver uses `load_source` and passes `icc` explicitly)."""
    global _src_icc
    s = load_source(path)
    _src_icc = s.icc
    return s.rgb, s.pixels, s.channels, s.hbd


_src_icc = None
_USE_CLI_ICC = object()


def codec_depth(tenbit: bool, hbd: bool):
    """-> (depth the encoder will write, note or None).  The reference writes 10-bit when --tenbit 1 or the
    source is 16-bit (io.zig:546); here that needs a libaom with high-bit-depth support behind the bridge."""
    from . import avif_bridge
    want = avif_bridge.output_depth(tenbit, hbd)
    if want == 8:
        return 8, None
    if not avif_bridge.available():
        return 8, ("note: the reference would write 10-bit here (--tenbit 1 / 16-bit source, io.zig:546-548); the "
                   f"libavif bridge is off ({avif_bridge.why_unavailable()}) and Pillow's plugin writes 8-bit, so "
                   "sizes and the chosen q are not comparable with oavif's own output")
    if not avif_bridge.supports_depth(10):
        return 8, ("note: the reference would write 10-bit here (--tenbit 1 / 16-bit source, io.zig:546-548); this "
                   "image's libaom has no high-bit-depth support (aom_codec_enc_init: \"Codec does not implement "
                   "requested capability\"), so the bitstream is 8-bit and sizes / the chosen q are not comparable "
                   "with oavif's own output; --tenbit 0 runs are the reference's calls exactly")
    return 10, None


def encoder_input(pixels, o: AvifEncOptions, icc=_USE_CLI_ICC):
    """The source as the encoder gets it, made ONCE per image instead of on every pass: rescaled to the encoder's
    depth (io.zig:566-617; SURVEY.md 8f rank 4, the loops are oavif_prescale_* of the C ABI), then wrapped,
    tagged and converted to YUV444 (io.zig:550-623: avifImageCreate + avifImageRGBToYUV, 15 % of a 4K encode) --
    only `quality` changes between the passes of a search (io.zig:625).  -> avif_bridge.EncoderSource.  Gray
    sources are expanded to RGB(A): the reference hands them to libavif as if they were RGB (io.zig:564), a
    row-stride bug this mirror does not reproduce."""
    import numpy as np
    from . import avif_bridge
    if icc is _USE_CLI_ICC:
        icc = _src_icc
    hbd = pixels.dtype == np.uint16
    depth, _note = codec_depth(o.tenbit, hbd)
    ch = pixels.shape[2]
    if ch == 1:
        pixels = np.repeat(pixels, 3, axis=2)
    elif ch == 2:
        pixels = np.concatenate([np.repeat(pixels[..., :1], 3, axis=2), pixels[..., 1:]], axis=2)
    scaled = avif_bridge.prescale_source(np.ascontiguousarray(pixels), depth)
    return avif_bridge.EncoderSource(scaled, depth, o, icc)


def _encode(src, o: AvifEncOptions, q: int, icc=_USE_CLI_ICC, prepared=None) -> bytes:
    """io.encodeAvifToBuffer (io.zig:544-636).  `prepared` = encoder_input(src, o, icc), hoisted by the callers
    that encode one source many times (then `src` and `icc` are not looked at again)."""
    from . import avif_bridge
    if icc is _USE_CLI_ICC:
        icc = _src_icc
    if avif_bridge.available():
        if prepared is not None:
            return prepared.encode(o, q)
        with encoder_input(src, o, icc) as once:
            return once.encode(o, q)
    import io as _io
    from PIL import Image
    if src.dtype != "uint8":
        src = (src >> 8).astype("uint8")
    mode = {1: "L", 2: "LA", 3: "RGB", 4: "RGBA"}[src.shape[2]]
    im = Image.fromarray(src[..., 0] if src.shape[2] == 1 else src, mode)
    buf = _io.BytesIO()
    extra = {"icc_profile": icc} if icc else {}
    im.save(buf, format="AVIF", quality=int(q), subsampling="4:4:4", speed=o.speed,
            max_threads=o.max_threads, **extra, tile_rows=o.tile_rows_log2, tile_cols=o.tile_cols_log2,
            autotiling=o.auto_tiling, advanced={"tune": o.tune} if o.tune == "ssim" else None)
    return buf.getvalue()


def _decode_rgb(data: bytes):
    """io.decodeAvifToRgb: 8-bit, alpha dropped (io.zig:638-666)."""
    from . import avif_bridge
    if avif_bridge.available():
        return avif_bridge.decode_rgb8(data)
    from . import synth
    return synth.avif_decode(data)


# ---- driver (main.zig:37-117) -----------------------------------------------------------------------

def main(argv: Optional[List[str]] = None, scorer=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    eprint(f"\x1b[31moavif\x1b[0m | {VERSION}")
    show_help = show_version = False
    for a in argv:  # only honoured while they are the leading arguments (main.zig:50-59)
        if a in ("--help", "-h"):
            show_help = True
        elif a in ("--version", "-v"):
            show_version = True
        else:
            break
    if show_help:
        print_usage()
        return 0
    if show_version:
        print_version()
        return 0
    own_scorer = False
    prefetched = None
    prepared = None
    try:
        o, inp, out = parse_args(argv)
        if scorer is None and o.quality is None and inp is not None and out is not None:
            # a search will need the scorer: start its once-per-process initialisation (HIP
            # runtime, code object: 0.15-0.35 s) now, behind the image load and the first encode
            if "torch" not in sys.modules:
                os.environ.setdefault("OAVIF_AMD_NO_TORCH", "1")  # a CLI run shares nothing with torch
            from . import _lib
            prefetched = int(os.environ.get("LOCAL_RANK", "0"))
            _lib.lib().ssimu2_prefetch(prefetched)
        if inp is None or out is None:
            raise CliError("MissingInputOrOutput")
        rgb, src, channels, hbd = load_image(inp)
        h, w, _ = rgb.shape
        eprint(f"Read {w}x{h}, {'RGBA' if channels > 3 else 'RGB'}, {16 if hbd else 8}-bit, "
               f"{os.path.getsize(inp)} bytes")
        # The reference encodes 10-bit when --tenbit 1 or the source is 16-bit (io.zig:546-548): say what
        # IS written (the note goes after the reference's own lines, so that their order -- main.zig:78-116,
        # what tools parse -- is kept)
        out_depth, depth_note = codec_depth(o.tenbit, hbd)
        prepared = encoder_input(src, o) if _bridge_on() else None   # once per image, not once per pass
        if o.quality is not None:  # bypass the search (main.zig:93-100)
            eprint(f"Encoding [q{o.quality}, speed {o.speed}, {out_depth}-bit]")
            data = _encode(src, o, o.quality, prepared=prepared)
            open(out, "wb").write(data)
            eprint(f"Compressed to {len(data)} bytes ({len(data) * 8 / (w * h):.3f} bpp)")
            if depth_note:
                eprint(depth_note)
            return 0

        eprint(f"Searching [tgt {_fmt_num(o.score_tgt)}±{o.tolerance:.1f}, speed {o.speed}, {out_depth}-bit]")
        from . import tq
        if scorer is None:
            from . import Ssimu2
            scorer = Ssimu2(int(os.environ.get("LOCAL_RANK", "0")), blur=blur_from_env())
            own_scorer = True
        cache = {}

        def codec(q: int):
            data = _encode(src, o, q, prepared=prepared)
            cache.clear()
            cache[q] = data  # EncBuffer keeps only the last probe (tq.zig:31-35)
            return _decode_rgb(data), len(data)

        fan = int(os.environ.get("OAVIF_PROBE_FANOUT", "1") or 1)
        if fan > 1 and own_scorer:
            # probes of the search fanned over `fan` scorer contexts (HIP streams) and host
            # threads (include/oavif_tq.h); same q, score and pass count as the plain search.
            # Not a CLI flag: the option surface stays the reference's (parse_args.zig:76-122).
            from . import Ssimu2
            dev = int(os.environ.get("LOCAL_RANK", "0"))
            ctxs = [scorer] + [Ssimu2(dev, blur=blur_from_env()) for _ in range(min(fan, 16) - 1)]

            def codec_keep(q: int):
                data = _encode(src, o, q, prepared=prepared)
                cache[q] = data  # every probe of a wave is kept: any of them may be the answer
                return _decode_rgb(data), len(data)
            try:
                r, _stats, _sizes = tq.search_speculative_hip(
                    ctxs, rgb, codec_keep, score_tgt=o.score_tgt, tolerance=o.tolerance,
                    max_pass=o.max_pass)
            finally:
                for c in ctxs[1:]:
                    c.close()
            r.buf_q = r.q if r.q in cache else r.buf_q
        elif _bridge_on():
            # decoded-frame hand-off: libavif's own RGB(A) rows go to the device as they
            # are; the alpha-dropping copy loop of io.decodeAvifToRgb (io.zig:654-663) does not run on the host
            from . import avif_bridge

            def codec_frame(q: int):
                data = _encode(src, o, q, prepared=prepared)
                cache.clear()
                cache[q] = data
                return avif_bridge.decode_common(data), len(data)
            r = tq.search_hip_frames(scorer, rgb, codec_frame, score_tgt=o.score_tgt, tolerance=o.tolerance,
                                     max_pass=o.max_pass)
        else:
            r = tq.search_hip(scorer, rgb, codec, score_tgt=o.score_tgt, tolerance=o.tolerance,
                              max_pass=o.max_pass)
        eprint(f"Found q{r.q} (score {r.score:.2f}, {r.num_pass} passes)")
        data = cache.get(r.q) if r.buf_q == r.q else None
        if data is None:  # main.zig:109-113
            data = _encode(src, o, r.q, prepared=prepared)
        open(out, "wb").write(data)
        eprint(f"Compressed to {len(data)} bytes ({len(data) * 8 / (w * h):.3f} bpp)")
        if depth_note:
            eprint(depth_note)
        return 0
    except CliError as e:
        eprint(f"error: {e.name}")
        return 1
    except FileNotFoundError:
        eprint("error: FileNotFound")
        return 1
    except Exception as e:  # codec / scorer failures end the run like any Zig error (main.zig:103)
        name = getattr(e, "name", type(e).__name__)
        eprint(f"error: {name}: {e}")
        return 1
    finally:
        if prepared is not None:
            prepared.close()
        if own_scorer and scorer is not None:
            scorer.close()
        elif prefetched is not None:
            # an early error exit (missing input, unsupported format, codec error) before any
            # context was created: let the background HIP start-up finish before the process
            # tears down under it
            from . import _lib
            _lib.lib().ssimu2_prefetch_join(prefetched)


if __name__ == "__main__":
    sys.exit(main())
