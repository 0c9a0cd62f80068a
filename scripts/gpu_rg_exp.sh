#!/bin/bash
# Per-kernel durations of recursive-mode A/B builds on the GPU box (rocprofv3 --kernel-trace --stats over
# scripts/gpu_rg_bench.py):  scripts/gpu_rg_exp.sh TAG NAME...   (NAME = a dir under gpurun_ablate/)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in "$@"; do
  export OAVIF_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_ablate/$d/liboavif_hip.so
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$d -- python3 $GRAFT_REPO_ROOT/scripts/gpu_rg_bench.py > $OUT/bench_$d.log 2>&1 || { echo "$d: rocprof failed"; tail -5 $OUT/bench_$d.log; exit 1; }
  f=$(find $OUT/prof_$d -name "*kernel_stats.csv" | head -1)
  echo "== $d: $(grep 'rg_bench:' $OUT/bench_$d.log)"
  grep -E "k_rg_|k_pyramid_bands_xyb|k_finalize|k_rgw" $f | cut -d, -f1-4,6 | sed 's/ssimu2:://; s/(ssimu2::RgPlan)//' | tee -a $OUT/summary.txt
  rm -rf $OUT/prof_$d
done
