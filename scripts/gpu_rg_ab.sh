#!/bin/bash
# Same-box A/B of recursive-mode library builds: scripts/gpu_rg_ab.sh DIR...  (each DIR holds a liboavif_hip.so)
for rep in 1 2; do
  for d in "$@"; do
    OAVIF_AMD_LIB=$GRAFT_REPO_ROOT/$d/liboavif_hip.so timeout -k 10 120 python3 scripts/gpu_rg_bench.py 2>/dev/null | grep "rg_bench:" | sed "s|^|$d  |"
  done
done
