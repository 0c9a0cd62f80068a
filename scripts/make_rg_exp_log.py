"""profiles/r04_rg_chain_vs_bytes.log from the summaries scripts/gpu_rg_exp.sh left under gpurun_out/ (round 4)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
runs = [("r4a", ["base", "h_nostore", "h_sametile", "h_chain", "v_samerows", "v_reverse", "h_ntld"]),
        ("r4b", ["h_chain_instr + OAVIF_RG_LDS_H=30000 (one k_rg_h workgroup per CU)", "base_instr + OAVIF_RG_LDS_H=30000"]),
        ("r4b2", ["base_instr + OAVIF_RG_LDS_V=70000 (one k_rg_v workgroup per CU)"]),
        ("r4e", ["p_chain", "p_nostore", "p_sametile"]),
        ("r4f", ["p_chain_long", "p_chain_fill", "p_long"]),
        ("r4h", ["n_base", "n_pers", "base"])]
print("""# Round 4: what bounds the recursive-mode kernels at 3840x2160 (one reference-cached pass).
# rocprofv3 --kernel-trace --stats over scripts/gpu_rg_bench.py (scripts/gpu_rg_exp.sh; builds by
# scripts/build_variant.sh); us = average over 154 launches.  Boxes of the pool differ by a few per cent;
# compare within a block (one gpurun call = one box).
# Diagnosis builds (their results are garbage, the timings are the point):
#   h_nostore   k_rg_h without its global stores        h_sametile  k_rg_h loading tile 0 over and over (cache-fed)
#   h_chain     both: no HBM traffic at all             v_samerows  k_rg_v loading rows 0-9 over and over (cache-fed)
#   v_reverse   k_rg_v column groups right to left      h_ntld      nontemporal loads in k_rg_h
#   OAVIF_RG_LDS_H / _V = N   N bytes of dynamic LDS added to the launch: one workgroup per CU (a knob of the
#               instrumented build at that commit; the persistent k_rg_v made it permanent, the knob is gone)
#   p_*         the PERSISTENT k_rg_h (one 8-wave workgroup per CU, two job queues, class by SIMD):
#               p_chain / p_nostore / p_sametile as above; p_chain_long = only the long class (972 full-resolution
#               chains + 52, one per SIMD) without HBM; p_chain_fill = only the filler class; p_long = long class, HBM-fed
#   n_base      round 4 as committed (k_rg_h per row group, products formed in staging; persistent k_rg_v)
#   n_pers      the same with -DRG_H_PERSISTENT=1        base = the round-3 kernels
""")
for tag, names in runs:
    path = os.path.join(G, tag, "summary.txt")
    if not os.path.exists(path):
        continue
    lines = [ln.strip() for ln in open(path) if ln.strip()]
    print(f"## {tag}")
    blocks, cur = [], []
    for ln in lines:
        cur.append(ln)
        if "k_finalize" in ln:
            blocks.append(cur)
            cur = []
    for name, blk in zip(names, blocks):
        vals = {}
        for ln in blk:
            m = re.match(r'"(?:void )?([a-z_0-9]+)(<[^>]*>)?.*?",(\d+),(\d+)', ln)
            if m:
                vals[m.group(1) + (m.group(2) or "")] = int(m.group(4)) / int(m.group(3)) / 1e3
        b = open(os.path.join(G, tag, f"bench_{name.split()[0]}.log")).read()
        mm = re.search(r"cached pass ([0-9.]+) ms", b)
        hp = vals.get("k_rg_h<false, false>", vals.get("k_rg_h_persistent<false, false>", 0))
        print(f"{name:72s} k_rg_h {hp:6.1f}  k_rg_v {vals.get('k_rg_v<false>', 0):6.1f}  convert "
              f"{vals.get('k_pyramid_bands_xyb', 0):5.1f} us   cached pass {mm.group(1) if mm else '?'} ms")
print("""
# Reading: k_rg_h without any HBM access (h_chain) takes 139 of its 177 us; a full-resolution chain alone on a
# SIMD takes 101 us (p_chain_long: 26 ns per step -- ~9 instructions at ~7 cycles each: a dependent chain does
# not issue every 4), two on one SIMD about twice that (one workgroup per CU: 181 us in two rounds).  Loads
# alone (+5) or stores alone (+12) cost little, both together +35-40 us: 0.70 GB of interleaved reads and writes
# at 4.0 TB/s.  The persistent form has the same no-HBM time (137) and is 30-50 us SLOWER HBM-fed.
# k_rg_v: 147 us cache-fed, 193 HBM-fed (0.93 GB), 170-185 with one workgroup per CU (persistent since round 4).
# Row pitch (a separate run, 3856 / 3904 / 3776 / 4096 wide): widths whose rows are not 512-byte multiples cost
# k_rg_v 30-45 % per pixel, k_rg_h up to 35 % when not a multiple of 64 columns.""")
