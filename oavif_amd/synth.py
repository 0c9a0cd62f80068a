"""Seeded synthetic RGB8 frames for parity tests and bench.py.

`make_ref`  : low-frequency colour field + band-limited noise + a few hard-edged shapes,
              so every pyramid scale and both the SSIM and the edge-difference terms are
              non-degenerate (not flat, not white noise).
`distort`   : deterministic distortion ladder spanning scores from ~30 to ~95:
              8x8 block quantisation, additive noise, box blur, or banding.
`avif_roundtrip`: real libavif/aom encode + dav1d decode through Pillow (CPU), the same
              kind of `dist` the reference produces at io.zig:544 / io.zig:638.

numpy only; nothing here touches the GPU or the oracle.
"""
from __future__ import annotations

import io

import numpy as np


def _upsample_bilinear(small: np.ndarray, h: int, w: int) -> np.ndarray:
    """(sh, sw, c) float32 -> (h, w, c) float32 by separable linear interpolation."""
    sh, sw = small.shape[:2]
    ys = np.linspace(0, sh - 1, h, dtype=np.float32)
    xs = np.linspace(0, sw - 1, w, dtype=np.float32)
    y0 = np.minimum(ys.astype(np.int32), sh - 2)
    x0 = np.minimum(xs.astype(np.int32), sw - 2)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    rows = small[y0] * (1 - fy) + small[y0 + 1] * fy  # (h, sw, c)
    return rows[:, x0] * (1 - fx) + rows[:, x0 + 1] * fx


def make_ref(w: int, h: int, seed: int = 0) -> np.ndarray:
    """Return an (h, w, 3) uint8 frame; same (w, h, seed) -> same bytes."""
    rng = np.random.default_rng(1234 + seed)
    # low-frequency colour field: ~64 px features
    gh, gw = max(h // 64, 2) + 1, max(w // 64, 2) + 1
    low = _upsample_bilinear(rng.random((gh, gw, 3), dtype=np.float32), h, w)
    # mid-frequency texture: ~8 px features, lower amplitude
    mh, mw = max(h // 8, 2) + 1, max(w // 8, 2) + 1
    mid = _upsample_bilinear(rng.random((mh, mw, 3), dtype=np.float32) - 0.5, h, w)
    # band-limited fine noise: 2 px features
    fh, fw = max(h // 2, 2) + 1, max(w // 2, 2) + 1
    fine = _upsample_bilinear(rng.random((fh, fw, 1), dtype=np.float32) - 0.5, h, w)
    img = 0.15 + 0.7 * low + 0.25 * mid + 0.08 * fine
    # hard-edged shapes (rectangles) so the edge-difference maps have real edges
    nshapes = 6 + (w * h) // (256 * 256)
    for _ in range(min(nshapes, 160)):
        x0 = int(rng.integers(0, w))
        y0 = int(rng.integers(0, h))
        sw = int(rng.integers(max(w // 40, 2), max(w // 6, 4)))
        sh = int(rng.integers(max(h // 40, 2), max(h // 6, 4)))
        col = rng.random(3, dtype=np.float32)
        a = float(rng.uniform(0.4, 1.0))
        sl = img[y0:y0 + sh, x0:x0 + sw]
        sl *= (1 - a)
        sl += a * col
    return np.ascontiguousarray(np.clip(img * 255.0 + 0.5, 0, 255).astype(np.uint8))


def distort(ref: np.ndarray, kind: str = "blockq", strength: int = 2, seed: int = 0) -> np.ndarray:
    """Deterministic distortions of `ref` ((h, w, 3) uint8) -> (h, w, 3) uint8."""
    h, w, _ = ref.shape
    f = ref.astype(np.float32)
    if kind == "blockq":  # 8x8 block mean mixed in + coarse quantisation
        q = [4, 8, 16, 32, 48][strength]
        bh, bw = (h + 7) // 8, (w + 7) // 8
        pad = np.pad(f, ((0, bh * 8 - h), (0, bw * 8 - w), (0, 0)), mode="edge")
        blocks = pad.reshape(bh, 8, bw, 8, 3)
        mean = blocks.mean(axis=(1, 3), keepdims=True)
        resid = np.round((blocks - mean) / q) * q
        out = (mean + resid).reshape(bh * 8, bw * 8, 3)[:h, :w]
    elif kind == "noise":
        sigma = [1, 2, 4, 8, 16][strength]
        rng = np.random.default_rng(99 + seed)
        out = f + rng.normal(0, sigma, f.shape).astype(np.float32)
    elif kind == "blur":
        k = [1, 2, 3, 4, 6][strength]
        out = f.copy()
        for _ in range(k):
            out[1:-1] = (out[:-2] + 2 * out[1:-1] + out[2:]) * 0.25
            out[:, 1:-1] = (out[:, :-2] + 2 * out[:, 1:-1] + out[:, 2:]) * 0.25
    elif kind == "band":
        q = [2, 4, 8, 16, 32][strength]
        out = np.floor(f / q) * q + q / 2
    else:
        raise ValueError(f"unknown distortion {kind!r}")
    return np.ascontiguousarray(np.clip(out + 0.5, 0, 255).astype(np.uint8))


def have_avif() -> bool:
    try:
        from PIL import features
        return bool(features.check("avif"))
    except Exception:
        return False


def avif_encode(rgb: np.ndarray, quality: int, speed: int = 9, max_threads: int | None = None) -> bytes:
    """CPU libavif/aom encode (YUV444) through Pillow; counterpart of io.zig:544.
    max_threads: encoder threads (oavif's --max-threads, default 1 there: parse_args.zig:51);
    None leaves Pillow's default (all cores)."""
    from PIL import Image
    buf = io.BytesIO()
    kw = {} if max_threads is None else {"max_threads": int(max_threads)}
    Image.fromarray(rgb).save(buf, format="AVIF", quality=int(quality), subsampling="4:4:4",
                              speed=int(speed), **kw)
    return buf.getvalue()


def avif_decode(data: bytes) -> np.ndarray:
    """CPU decode to tight RGB8; counterpart of io.zig:638-666."""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))


def avif_roundtrip(rgb: np.ndarray, quality: int, speed: int = 9, max_threads: int | None = None):
    data = avif_encode(rgb, quality, speed, max_threads)
    return avif_decode(data), len(data)
