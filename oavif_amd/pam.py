"""PAM (Netpbm P7) ingest with the acceptance rules of /root/reference/src/io.zig:309-406.

Dependency-free loader so that synthetic frames can be fed to the CLI / batch driver without
libspng.  Rules kept from the reference:
  * magic "P7" (io.zig:318); header ends at the first "ENDHDR\\n", else at the first blank line
    (io.zig:320-331), else HeaderNotFound;
  * header lines are split on CR/LF, '#' lines are comments, keys are matched as prefixes
    (WIDTH / HEIGHT / DEPTH / MAXVAL / TUPLTYPE), the first whitespace-separated token after the
    key is the value (io.zig:343-365);
  * all four of WIDTH, HEIGHT, DEPTH, MAXVAL must be non-zero (io.zig:367-368); MAXVAL must be
    255 (io.zig:369); DEPTH must be 1..4 (io.zig:370-371);
  * TUPLTYPE (case-insensitive) GRAYSCALE / GRAYSCALE_ALPHA / RGB / RGB_ALPHA must agree with
    DEPTH (io.zig:374-386); BLACKANDWHITE is rejected (io.zig:387-390); any other tuple type
    (including none) is accepted and DEPTH decides the channel count;
  * the raster must hold width*height*depth bytes after the header (io.zig:393-395); extra
    bytes are ignored.
Errors carry the reference's Zig error names.
"""
from __future__ import annotations

import re


class PamError(Exception):
    def __init__(self, name: str):
        super().__init__(name)
        self.name = name


def _first_token(rest: bytes):
    toks = re.split(rb"[ \t]+", rest.strip(b" \t"))
    return toks[0] if toks and toks[0] else None


def _parse_usize(tok: bytes) -> int:
    try:
        if not re.fullmatch(rb"\+?[0-9_]+", tok) or tok.strip(b"+_") == b"":
            raise ValueError
        return int(tok.replace(b"_", b""), 10)
    except ValueError:
        raise PamError("InvalidCharacter")


def load_pam(buf: bytes):
    """-> (raster bytes, width, height, channels)"""
    if len(buf) < 3 or not buf.startswith(b"P7"):
        raise PamError("NotAPamFile")
    i = buf.find(b"ENDHDR\n")
    if i >= 0:
        header_end = i + 7
    else:
        j = buf.find(b"\n\n")
        if j < 0:
            raise PamError("HeaderNotFound")
        header_end = j + 2
    width = height = depth = maxval = 0
    tuple_type = b"UNSPECIFIED"
    for line in re.split(rb"[\r\n]+", buf[:header_end]):
        if not line or line[:1] == b"#":
            continue
        if line.startswith(b"WIDTH"):
            t = _first_token(line[5:])
            if t is not None:
                width = _parse_usize(t)
        elif line.startswith(b"HEIGHT"):
            t = _first_token(line[6:])
            if t is not None:
                height = _parse_usize(t)
        elif line.startswith(b"DEPTH"):
            t = _first_token(line[5:])
            if t is not None:
                depth = _parse_usize(t)
        elif line.startswith(b"MAXVAL"):
            t = _first_token(line[6:])
            if t is not None:
                maxval = _parse_usize(t)
        elif line.startswith(b"TUPLTYPE"):
            t = _first_token(line[8:])
            if t is not None:
                tuple_type = t
        elif line == b"ENDHDR":
            break
    if width == 0 or height == 0 or depth == 0 or maxval == 0:
        raise PamError("InvalidPamDimensions")
    if maxval != 255:
        raise PamError("UnsupportedPamMaxVal")
    if depth not in (1, 2, 3, 4):
        raise PamError("UnsupportedPamDepth")
    channels = depth
    tt = tuple_type.upper()
    want = {b"GRAYSCALE": 1, b"GRAYSCALE_ALPHA": 2, b"RGB": 3, b"RGB_ALPHA": 4}
    if tt in want:
        if depth != want[tt]:
            raise PamError("PamTupleMismatch")
        channels = want[tt]
    elif tt == b"BLACKANDWHITE":
        raise PamError("UnsupportedPamTuple")
    size = width * height * channels
    if header_end + size > len(buf):
        raise PamError("InsufficientDataInFile")
    return buf[header_end:header_end + size], width, height, channels


def write_pam(rgb) -> bytes:
    """(h, w, c) uint8 -> PAM bytes (c = 1..4), for tests and synthetic inputs."""
    h, w, c = rgb.shape
    tt = {1: "GRAYSCALE", 2: "GRAYSCALE_ALPHA", 3: "RGB", 4: "RGB_ALPHA"}[c]
    hdr = f"P7\nWIDTH {w}\nHEIGHT {h}\nDEPTH {c}\nMAXVAL 255\nTUPLTYPE {tt}\nENDHDR\n"
    return hdr.encode() + rgb.tobytes()
