#!/bin/bash
# A/B build of the product library with extra hipcc flags:  scripts/build_variant.sh NAME [-DFLAG ...]
# -> gpurun_ablate/NAME/liboavif_hip.so (git-ignored; travels to the GPU box; select with OAVIF_AMD_LIB)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_ablate/$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize \
  -fvisibility=hidden -fvisibility-inlines-hidden -Wall -Wno-unused-function "$@" \
  -o $ROOT/gpurun_ablate/$NAME/liboavif_hip.so \
  $ROOT/oavif_amd/csrc/ssimu2_hip.hip $ROOT/oavif_amd/csrc/tq.cpp $ROOT/oavif_amd/csrc/png_ingest.cpp -lz
echo built gpurun_ablate/$NAME
