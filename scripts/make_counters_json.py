"""profiles/counters.json from one round's rocprofv3 PMC run.

    python scripts/make_counters_json.py gpurun_out/TAG/pmc_summary.txt [gpurun_out/valu_rate.txt] > profiles/counters.json

Input: the per-kernel counter medians scripts/gpu_pmc_sets.sh writes (scripts/pmc_summary.py
format) over scripts/gpu_kbench.py -- 4K, rotating over 8 pairs, so FETCH_SIZE is HBM traffic --
and, optionally, the VALU issue-rate table of scripts/ubench/valu_rate.  Output: per-launch HBM
bytes and instruction counts of the marching kernels, stamped with the hash of the kernel
sources they were measured on (bench.py marks the file stale when the sources change).
FETCH_SIZE is doubled: on gfx950 it reports half the bytes of a coalesced read
(MI355X_MICROARCH.md, HBM; confirmed by scripts/ubench/fetch_calib).  Units: FETCH_SIZE /
WRITE_SIZE are KiB per dispatch.
"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_summary(path):
    out, cur = {}, None
    for ln in open(path):
        if not ln.strip():
            continue
        if not ln.startswith(" "):
            cur = ln.strip()
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+median=(\S+)\s+max=(\S+)", ln)
            if m and cur is not None:
                out[cur][m.group(1)] = float(m.group(3))
    return out


def source_hash():
    h = hashlib.sha256()
    for name in ("ssimu2_kernels.h", "ssimu2_hip.hip"):
        with open(os.path.join(ROOT, "oavif_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main():
    summ = parse_summary(sys.argv[1])

    def kern(name):
        for k, v in summ.items():
            if k.split("::")[-1] == name:
                return v
        return {}
    march, refb = kern("k_march"), kern("k_march_refblur")
    doc = {
        "source": f"{os.path.relpath(sys.argv[1], ROOT)} (rocprofv3 --pmc passes of scripts/gpu_pmc_sets.sh over "
                  "scripts/gpu_kbench.py: one 3840x2160 workload rotating over 8 pairs); made by "
                  "scripts/make_counters_json.py",
        "kernel_source_hash": source_hash(),
        "workload": "3840x2160 RGB8 pairs, 8 distinct pairs in rotation",
        "simds": 1024,
        "fetch_size_correction": 2.0,
    }
    for tag, c in (("march", march), ("march_cached_reference", refb)):
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            doc[f"{tag}_fetch_KiB_reported"] = c["FETCH_SIZE"]
            doc[f"{tag}_write_KiB_reported"] = c["WRITE_SIZE"]
            doc[f"{tag}_hbm_bytes_per_launch"] = int(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024)
        for ctr, key in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_SALU", "salu"), ("SQ_INSTS_LDS", "lds"),
                         ("SQ_INSTS_VMEM", "vmem")):
            if ctr in c:
                doc[f"{tag}_{key}_wave_instructions_per_launch"] = c[ctr]
        for ctr in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_LDS_BANK_CONFLICT",
                    "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_TRANS_F32"):
            if ctr in c:
                doc[f"{tag}_{ctr}"] = c[ctr]
    if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
        # peak a plain VOP2 stream reaches at 8 waves per SIMD on this chip
        for ln in open(sys.argv[2]):
            m = re.match(r"v_mul/v_add\s+waves/SIMD=8\s+\S+ ms\s+->\s+(\S+) ns", ln)
            if m:
                doc["measured_peak_valu_wave_instructions_per_ns_per_simd"] = round(1.0 / float(m.group(1)), 4)
                doc["measured_peak_source"] = (f"{os.path.relpath(sys.argv[2], ROOT)}: v_mul/v_add stream, 8 waves per SIMD "
                                               "(scripts/ubench/valu_rate)")
    json.dump(doc, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
