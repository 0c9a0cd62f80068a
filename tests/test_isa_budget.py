"""Register / LDS budgets of the gfx950 kernels, read from the compiler's own metadata (hipcc
cross-compiles without a GPU).  The occupancy arguments of DESIGN.md section 4 rest on them:
the marching kernels need <= 80 VGPRs and ~50 KB of LDS for three 8-wave workgroups per CU (six
waves per SIMD); the recursive horizontal pass needs its LDS under a third of the CU's 160 KB;
nothing may spill.  CPU only (one device-only compile of the scorer translation unit, ~40 s)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "oavif_amd", "csrc", "ssimu2_hip.hip")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc missing")
    out = tmp_path_factory.mktemp("isa") / "scorer.s"
    # the flags of oavif_amd/build.py that shape device code
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                    "-S", "--cuda-device-only", "-o", str(out), SRC], check=True, capture_output=True)
    text = open(out).read()
    meta = {}
    for block in text.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        meta[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", block).group(1))
                      for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size")}
    return meta, text


def _one(meta, fragment):
    hits = [v for k, v in meta.items() if fragment in k]
    assert hits, fragment
    return hits


def test_marching_kernels_keep_six_waves_per_simd(kernels):
    meta, text = kernels
    for frag in ("7k_marchE", "15k_march_refblur", "10k_ref_blur"):
        for k in _one(meta, frag):
            assert k["vgpr_count"] <= 80 and k["vgpr_spill_count"] == 0, (frag, k)
            assert 3 * k["group_segment_fixed_size"] <= 160 * 1024, (frag, k)      # three workgroups per CU
    body = text.split("_ZN6ssimu27k_marchENS_9MarchPlanE:")[1].split("s_endpgm")[0]
    assert "v_mfma" not in body and "v_pk_" not in body        # stencil work: no matrix ops, no packed math (slower here)
    assert "scratch_" not in body


def test_recursive_kernels_do_not_spill_and_fit_three_workgroups(kernels):
    meta, _ = kernels
    for k in _one(meta, "k_rg_hILb"):
        assert k["vgpr_spill_count"] == 0 and k["vgpr_count"] <= 256, k
        assert 3 * k["group_segment_fixed_size"] <= 160 * 1024, k                  # 642 workgroups at 4K need 2.5 per CU
    for frag in ("k_rg_vILb", "k_rg_v_emitILb", "k_pyramid_bands"):
        for k in _one(meta, frag):
            assert k["vgpr_spill_count"] == 0, (frag, k)
    for k in _one(meta, "k_rg_vILb"):
        assert k["vgpr_count"] <= 128, k                                           # 8-wave workgroups, two per CU
