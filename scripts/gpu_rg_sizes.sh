#!/bin/bash
# The recursive mode at sizes other than 3840x2160 (row-pitch sensitivity, BASELINE frame sizes):
#   scripts/gpu_rg_sizes.sh TAG [LIBDIR...]     LIBDIR = a dir under gpurun_ablate/ ("-" = the in-tree library)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
LIBS=${@:--}
for lib in $LIBS; do
  if [ "$lib" = "-" ]; then unset OAVIF_AMD_LIB; else export OAVIF_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_ablate/$lib/liboavif_hip.so; fi
  for wh in "3840 2160" "3856 2160" "3904 2160" "3776 2160" "4000 3000" "6000 4000" "1920 1080" "7680 4320" "512 512"; do
    set -- $wh
    r=$(timeout -k 10 200 python3 scripts/gpu_rg_bench.py $1 $2 2>/dev/null | grep "rg_bench:" | sed 's/  device memory.*//')
    echo "$lib ${1}x${2}  $r" | tee -a $OUT/sizes.log
  done
done
