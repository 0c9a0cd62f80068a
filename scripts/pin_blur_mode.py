#!/usr/bin/env python3
"""Which blur -- and which STAGE -- does fssimu2 follow?  

The scorer has three blur modes -- `fir` (the 9-tap impulse response of the published recursion), `recursive`
(the published fp32 recursion, scalar order) and `recursive_fma` (its multiply-subtract fused) -- which differ
by 0.1 to 3.8 points on the pairs of tests/golden/pin_kit/, far more than the +-0.01 north_star allows.  Which
one fssimu2 0.1.1 agrees with cannot be found out in this repository's build environment (no Zig, no fssimu2
source).  And fssimu2 may differ from the published algorithm somewhere else: the kit therefore also records,
per pair, the CPU checker's score with ONE stage switched to a plausible alternative (blur edge rule, a true
Gaussian instead of the recursion's impulse response, products rounded first, XYB-domain or floor-sized
downsampling, the size test after downsampling, fp32 transfer curve, libm cube root, fp32 map sums: 22 entries,
`--variants` lists them), so that one run of fssimu2 names the stage that differs instead of "no mode matches".
Someone who can run fssimu2 does it like this:

  1. python3 scripts/pin_blur_mode.py --write-pairs DIR       (the kit's pairs as PNG files: the committed
                                                               ones copied, the full-size ones regenerated
                                                               from their seeds and checked against sha256)
  2. score every DIR/ref_*.png against its DIR/dist_*.png with fssimu2 (e.g. `fssimu2 ref.png dist.png`, or
     computeSsimu2 on the decoded RGB8 buffers), and write one line per pair:   name,score
  3. python3 scripts/pin_blur_mode.py results.txt             prints the distance to every variant, nearest
                                                               first, and a verdict: MATCH (a blur mode of the
                                                               scorer within +-0.01 on every pair), STAGE (a
                                                               variant the scorer does not implement is within
                                                               +-0.01: that stage differs), or NO VARIANT MATCHES

  python3 scripts/pin_blur_mode.py --list                      the pairs and the recorded scores
  python3 scripts/pin_blur_mode.py --variants                  the catalogue: name, stage, what differs

No dependency beyond numpy + zlib (the PNG reader / writer below handles 8-bit RGB only, which is what the kit
holds); --write-pairs needs the repo (oavif_amd.synth) for the generated pairs."""
import hashlib
import json
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "tests", "golden", "pin_kit")
MODES = ("fir", "recursive", "recursive_fma")

# full-size pairs, regenerated from seeds: oavif_amd.synth.make_ref + a deterministic numpy distortion
GENERATED = {
    "g1080_noise1": {"w": 1920, "h": 1080, "seed": 7001, "kind": "noise", "strength": 0},
    "g1080_blockq1": {"w": 1920, "h": 1080, "seed": 7003, "kind": "blockq", "strength": 1},
    "g4k_noise1": {"w": 3840, "h": 2160, "seed": 7002, "kind": "noise", "strength": 0},
    "g4k_blockq2": {"w": 3840, "h": 2160, "seed": 7004, "kind": "blockq", "strength": 2},
}


def generate(name):
    sys.path.insert(0, ROOT)
    from oavif_amd import synth
    g = GENERATED[name]
    ref = synth.make_ref(g["w"], g["h"], g["seed"])
    return ref, synth.distort(ref, g["kind"], g["strength"], seed=g["seed"])


def sha256_pixels(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, np.uint8).tobytes()).hexdigest()


def png_rgb8(px: np.ndarray) -> bytes:
    """(h, w, 3) uint8 -> a PNG file (filter 0 on every row, one IDAT)."""
    h, w, _ = px.shape

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), np.ascontiguousarray(px, np.uint8).reshape(h, w * 3)], axis=1).tobytes()
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def read_png_rgb8(path: str) -> np.ndarray:
    """The reader for what png_rgb8 writes (8-bit RGB, filter type 0 rows, no interlace)."""
    buf = open(path, "rb").read()
    assert buf[:8] == b"\x89PNG\r\n\x1a\n", path
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(buf):
        n, kind = struct.unpack(">I4s", buf[pos:pos + 8])
        data = buf[pos + 8:pos + 8 + n]
        if kind == b"IHDR":
            w, h, depth, ctype, _, _, il = struct.unpack(">IIBBBBB", data)
            assert (depth, ctype, il) == (8, 2, 0), "the kit's reader handles 8-bit RGB only"
        elif kind == b"IDAT":
            idat += data
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert not rows[:, 0].any(), "the kit's reader handles filter type 0 only"
    return rows[:, 1:].reshape(h, w, 3).copy()


def load_kit():
    return json.load(open(os.path.join(KIT, "pin_kit.json")))


def rank_variants(kit, results):
    """results: {pair name: score}.  -> [(variant name, worst |d| over the given pairs, mean |d|, stage)], nearest first.
    Pairs recorded before the catalogue existed (no `variant_scores`) count for the three blur modes only."""
    by_name = {p["name"]: p for p in kit["pairs"]}
    out = []
    for v, meta in kit.get("variants", {m: {"stage": "blur"} for m in MODES}).items():
        ds = []
        for name, score in results.items():
            rec = by_name[name].get("variant_scores") or by_name[name]["scores"]
            if v in rec:
                ds.append(abs(score - rec[v]))
        if ds:
            out.append((v, max(ds), sum(ds) / len(ds), meta.get("stage", "?")))
    return sorted(out, key=lambda t: (t[1], t[2]))


def classify(kit, results):
    """results: {pair name: score}.  -> (verdict line, per-pair rows, per-mode worst distance)."""
    tol = float(kit.get("tolerance", 0.01))
    rows, worst = [], {m: 0.0 for m in MODES}
    by_name = {p["name"]: p for p in kit["pairs"]}
    for name, score in results.items():
        if name not in by_name:
            raise SystemExit(f"unknown pair {name!r}; the kit has: {', '.join(by_name)}")
        rec = by_name[name]["scores"]
        d = {m: abs(score - rec[m]) for m in MODES}
        for m in MODES:
            worst[m] = max(worst[m], d[m])
        rows.append((name, score, d, min(MODES, key=lambda m: d[m])))
    match = [m for m in MODES if worst[m] <= tol]
    libm = kit.get("libm_dependent_variants", {})     # variants whose recorded scores depend on the recording host's libm
    ranked = rank_variants(kit, results) if rows else []
    close = [r for r in ranked if r[1] <= tol and r[0] not in MODES]     # variants the scorer does not implement
    if not rows:
        verdict = "no results given"
    elif len(match) == 1:
        same = [r[0] for r in close]
        verdict = (f"MATCH: {match[0]} (every pair within +-{tol}); set the shim's `blur` / OAVIF_SSIMU2_BLUR to it and re-pin the oracle"
                   + (f" [also within +-{tol}, i.e. not told apart from it by these pairs and not needing to be: {', '.join(same)}]" if same else ""))
    elif match:
        verdict = f"AMBIGUOUS: {', '.join(match)} all within +-{tol} -- score the full-size pairs too (the modes are 0.4-2 points apart there)"
    elif close:
        v, w_, _, stage = close[0]
        what = kit.get("variants", {}).get(v, {}).get("what", "")
        others = [r[0] for r in close[1:]]
        verdict = (f"STAGE: no blur mode of the scorer matches, but the checker's variant `{v}` does (every pair within +-{tol}, worst "
                   f"{w_:.4f}): fssimu2 differs from the published algorithm in the {stage.upper()} stage -- {what}.  The HIP scorer does "
                   f"not implement that variant; the oracle does (oracle/ssimu2_oracle.c OR_VAR_*): it is the specification of the change"
                   + (f" [also within +-{tol}: {', '.join(others)}]" if others else "")
                   + (f" [INDICATIVE ONLY: `{v}` calls the host libm (powf / cbrtf), whose last bits differ between glibc versions; its "
                      f"scores were recorded with {libm.get('recorded_with', 'an unrecorded libm')}]" if v in libm.get("names", ()) else ""))
    else:
        near = min(MODES, key=lambda m: worst[m])
        v, w_, _, stage = ranked[0]
        verdict = (f"NO MODE MATCHES within +-{tol}: nearest mode is {near} (worst pair {worst[near]:.4f} away); nearest variant of the "
                   f"catalogue is `{v}` ({stage} stage, worst pair {w_:.4f} away).  "
                   + ("A distance of a few hundredths to a recursive variant is what a last-bit difference BEFORE the blur looks like "
                      "(the recursion amplifies it: see the srgb_powf / cbrt_libm rows of --variants); " if w_ <= 0.1 and v.startswith("recursive") else "")
                   + "fssimu2's source is then the only way to +-0.01")
    return verdict, rows, worst


def main(argv):
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 0
    kit = load_kit()
    if argv[0] == "--list":
        for p in kit["pairs"]:
            s = p["scores"]
            print(f"{p['name']:16s} {p['width']}x{p['height']:<5d} {p['kind']:9s} fir {s['fir']:.4f}  recursive {s['recursive']:.4f}  "
                  f"recursive_fma {s['recursive_fma']:.4f}   ({p['distortion']})")
        return 0
    if argv[0] == "--write-pairs":
        out = argv[1]
        os.makedirs(out, exist_ok=True)
        for p in kit["pairs"]:
            if p["kind"] == "committed":
                ref, dst = read_png_rgb8(os.path.join(KIT, p["ref"])), read_png_rgb8(os.path.join(KIT, p["dist"]))
            else:
                ref, dst = generate(p["name"])
            if sha256_pixels(ref) != p["sha256_ref"] or sha256_pixels(dst) != p["sha256_dist"]:
                raise SystemExit(f"{p['name']}: the pixels do not have the recorded sha256 (numpy / synth changed?) -- do not score this pair")
            open(os.path.join(out, f"ref_{p['name']}.png"), "wb").write(png_rgb8(ref))
            open(os.path.join(out, f"dist_{p['name']}.png"), "wb").write(png_rgb8(dst))
            print(f"wrote {p['name']}: ref_{p['name']}.png dist_{p['name']}.png ({p['width']}x{p['height']})")
        return 0
    if argv[0] == "--variants":
        for v, meta in kit.get("variants", {}).items():
            gaps = []
            for p in kit["pairs"]:
                vs = p.get("variant_scores", {})
                base = "recursive" if v.startswith("recursive") else "fir"
                if v in vs and base in vs:
                    gaps.append(abs(vs[v] - vs[base]))
            print(f"{v:28s} {meta['stage']:8s} {'[HIP mode] ' if meta.get('implemented_by_the_hip_scorer') else ''}{meta['what']}"
                  + (f"   (moves the kit's scores by {min(gaps):.4f} .. {max(gaps):.4f} against `{'recursive' if v.startswith('recursive') else 'fir'}`)" if gaps and max(gaps) > 0 else "")
                  + ("   [indicative only: host libm, recorded with " + kit["libm_dependent_variants"].get("recorded_with", "?") + "]"
                     if v in kit.get("libm_dependent_variants", {}).get("names", ()) else ""))
        return 0
    results = {}
    for ln in open(argv[0]):
        ln = ln.strip()
        if not ln or ln.startswith("#"):
            continue
        name, score = ln.replace(";", ",").replace("\t", ",").split(",")[:2]
        results[name.strip()] = float(score)
    verdict, rows, worst = classify(kit, results)
    for name, score, d, nearest in rows:
        print(f"{name:16s} given {score:9.4f}   |d| fir {d['fir']:.4f}  recursive {d['recursive']:.4f}  recursive_fma {d['recursive_fma']:.4f}   nearest: {nearest}")
    print("worst distance per mode: " + "  ".join(f"{m} {worst[m]:.4f}" for m in MODES))
    print("every variant of the catalogue, nearest first (worst / mean |d| over the given pairs):")
    for v, w_, mean, stage in rank_variants(kit, results):
        print(f"  {v:28s} {stage:8s} worst {w_:8.4f}  mean {mean:8.4f}")
    print(verdict)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
