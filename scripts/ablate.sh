#!/bin/bash
# Build timing-only ablation variants of the library (results are WRONG by construction;
# only the kernel time matters).  Run on the build container: scripts/ablate.sh ; then on the
# GPU box:  for v in base noconv novmaps nobarrier; do OAVIF_AMD_LIB=... python scripts/gpu_kbench.py; done
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_ablate
for v in base:"" noconv:-DABL_NOCONV novmaps:-DABL_NOVMAPS nobarrier:-DABL_NOBARRIER nohvconv:"-DABL_NOCONV -DABL_NOVMAPS"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize $flags \
     -o gpurun_ablate/lib_$name.so oavif_amd/csrc/ssimu2_hip.hip oavif_amd/csrc/tq.cpp
done
ls -la gpurun_ablate
