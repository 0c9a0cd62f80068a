#!/usr/bin/env python3
"""Fit the CPU checker to fssimu2's scores of the pin kit: which COMBINATION of stage alternatives explains them?

scripts/pin_blur_mode.py ranks the 22 recorded entries of the kit (the three blur modes + single-stage variants)
against a results file and needs nothing but numpy.  If fssimu2 differs from the published algorithm in more than
one stage at once -- say the recursion AND a libm cube root AND fp32 map sums -- no single entry matches.  This tool
(test infrastructure: it runs oracle/ssimu2_oracle.c, so it lives under tests/) searches the combinations: for each
blur base (fir / fir with products first / recursive / recursive_fma) a greedy forward selection over the OR_VAR_*
bits that are legal on that base, scoring each candidate by its worst |score - given| over the pairs of the results
file, until nothing improves.  It prints every step, the best combination per base and the overall best, and says
whether that is within the kit's tolerance (+-0.01) on every pair -- in which case `oracle.compute_ssimu2_variant(
ref, dist, <blur>, <bits>)` IS a bit-level specification of what fssimu2 computes on these pairs, and the kernels can
be changed to follow it.  The fit is A combination within tolerance, not necessarily the only one: on the small
pairs the last-bit alternatives (srgb_powf / cbrt_libm / sums_f32) move a score by less than the tolerance and stand in
for each other; the full-size pairs (--full) separate them under a recursive base (0.02-0.05 each there).

    python3 tests/tools/pin_fit.py results.txt            # the small committed pairs of results.txt (seconds each)
    python3 tests/tools/pin_fit.py results.txt --full     # also the 1080p / 4K pairs (a few minutes)

results.txt: one `name,score` line per pair, as for scripts/pin_blur_mode.py.  Parity stays UNPINNED until someone
runs fssimu2; this only shortens the way from its scores to a matching checker."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ssimu2_oracle as orc  # noqa: E402  (checker: test infrastructure)

_spec = importlib.util.spec_from_file_location("pin_blur_mode", os.path.join(ROOT, "scripts", "pin_blur_mode.py"))
kit = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(kit)

STAGE_BITS = {"downsample_xyb": orc.VAR_DOWNSAMPLE_XYB, "downsample_floor": orc.VAR_DOWNSAMPLE_FLOOR,
              "size_test_after": orc.VAR_SIZE_TEST_AFTER, "srgb_powf": orc.VAR_SRGB_POWF, "cbrt_libm": orc.VAR_CBRT_LIBM,
              "sums_f32": orc.VAR_SUMS_F32}
BLUR_BITS = {"edge_clamp": orc.VAR_EDGE_CLAMP, "edge_mirror": orc.VAR_EDGE_MIRROR, "gauss9": orc.VAR_GAUSS9,
             "gauss11": orc.VAR_GAUSS11}
BASES = {"fir": (orc.BLUR_FIR, False), "fir_prodfirst": (orc.BLUR_FIR_PRODFIRST, True),
         "recursive": (orc.BLUR_IIR, False), "recursive_fma": (orc.BLUR_IIR_FMA, False)}
EXCLUSIVE = [{"edge_clamp", "edge_mirror"}, {"gauss9", "gauss11"}]
MIN_GAIN = 1e-3   # a further stage must lower the worst distance by this much (a tenth of the kit's tolerance) to be named


def load_pairs(doc, names, full):
    out = {}
    for p in doc["pairs"]:
        if p["name"] not in names or (p["kind"] != "committed" and not full):
            continue
        if p["kind"] == "committed":
            ref, dst = kit.read_png_rgb8(os.path.join(kit.KIT, p["ref"])), kit.read_png_rgb8(os.path.join(kit.KIT, p["dist"]))
        else:
            ref, dst = kit.generate(p["name"])
        out[p["name"]] = (ref, dst)
    return out


class Fitter:
    def __init__(self, pairs, given, omp=True):
        self.pairs, self.given, self.omp, self.cache = pairs, given, omp, {}

    def distance(self, blur, bits):
        """(worst, mean) |score - given| over the pairs for the checker in (blur, bits)."""
        key = (blur, bits)
        if key not in self.cache:
            d = [abs(orc.compute_ssimu2_variant(r, t, blur, bits, omp=self.omp) - self.given[n]) for n, (r, t) in self.pairs.items()]
            self.cache[key] = (max(d), sum(d) / len(d))
        return self.cache[key]

    def fit_base(self, base, log=print, max_bits=5, good_enough=0.005):
        blur, with_blur_bits = BASES[base]
        cand = dict(STAGE_BITS)
        if with_blur_bits:
            cand.update(BLUR_BITS)
        chosen, bits = [], 0
        best = self.distance(blur, 0)
        log(f"  {base:14s} (no variant)                     worst {best[0]:8.4f}  mean {best[1]:8.4f}")
        while len(chosen) < max_bits and best[0] > good_enough:   # half the tolerance: explained, stop naming stages
            step = None
            for name, bit in cand.items():
                if name in chosen or any(name in ex and ex & set(chosen) for ex in EXCLUSIVE):
                    continue
                d = self.distance(blur, bits | bit)
                if d < (step[1] if step else best):
                    step = (name, d, bit)
            if step is None or step[1][0] > best[0] - MIN_GAIN:   # parsimony: a stage is named only if it buys something
                break
            chosen.append(step[0])
            bits |= step[2]
            best = step[1]
            log(f"  {base:14s} + {' + '.join(chosen):32s} worst {best[0]:8.4f}  mean {best[1]:8.4f}")
        return {"base": base, "blur": blur, "bits": bits, "stages": chosen, "worst": best[0], "mean": best[1]}


def fit(doc, results, full=False, log=print, omp=None):
    """omp: OpenMP build of the checker (default: only with --full; on the small pairs one thread is faster)."""
    omp = full if omp is None else omp
    pairs = load_pairs(doc, set(results), full)
    if not pairs:
        raise SystemExit("none of the pairs of the results file is a " + ("pair of the kit" if full else "small committed pair (try --full)"))
    f = Fitter(pairs, results, omp)
    log(f"fitting on {len(pairs)} pair(s): {', '.join(pairs)}")
    tol = float(doc.get("tolerance", 0.01))
    fits = [f.fit_base(b, log, good_enough=tol / 2) for b in BASES]
    best = min(fits, key=lambda x: (x["worst"] > tol / 2, len(x["stages"]) if x["worst"] <= tol / 2 else 0, x["worst"], x["mean"]))
    best["within_tolerance"] = best["worst"] <= tol
    best["evaluations"] = len(f.cache)
    return best, fits


def main(argv):
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 0
    results = {}
    for ln in open(argv[0]):
        ln = ln.strip()
        if ln and not ln.startswith("#"):
            name, score = ln.replace(";", ",").replace("\t", ",").split(",")[:2]
            results[name.strip()] = float(score)
    orc.build()
    doc = kit.load_kit()
    best, _ = fit(doc, results, full="--full" in argv)
    what = " + ".join([best["base"]] + best["stages"])
    print(f"best: {what}   worst {best['worst']:.4f}  mean {best['mean']:.4f} over the pairs given ({best['evaluations']} checker runs)")
    if best["within_tolerance"]:
        print(f"WITHIN +-{doc['tolerance']} on every pair: oracle.compute_ssimu2_variant(ref, dist, blur={best['blur']}, variant={best['bits']:#x}) "
              "reproduces these scores; " + ("that is a mode of the HIP scorer as it is." if not best["stages"] and best["base"] != "fir_prodfirst"
                                             else "the stages named are what the kernels would have to change."))
    else:
        print(f"NOT within +-{doc['tolerance']}: no combination of the catalogue's alternatives explains these scores; fssimu2's source is the way.")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
