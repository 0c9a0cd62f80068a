#!/bin/bash
# Like build_variant.sh, for the INSTRUMENTED library (timing hooks; scripts/gpu_ab.py needs them):
#   scripts/build_variant_instr.sh NAME [-DFLAG ...]  -> gpurun_ablate/NAME/liboavif_hip_instr.so
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_ablate/$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize \
  -fvisibility=hidden -fvisibility-inlines-hidden -Wall -Wno-unused-function "$@" \
  -o $ROOT/gpurun_ablate/$NAME/liboavif_hip_instr.so \
  $ROOT/oavif_amd/csrc/ssimu2_instrument.hip $ROOT/oavif_amd/csrc/tq.cpp $ROOT/oavif_amd/csrc/png_ingest.cpp -lz
echo built gpurun_ablate/$NAME instr
