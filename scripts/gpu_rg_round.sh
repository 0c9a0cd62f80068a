#!/bin/bash
# Recursive-mode check on the GPU box: parity tests, then scripts/gpu_rg_bench.py under
# rocprofv3 --kernel-trace --stats (per-kernel split).  Usage (repo root): scripts/gpu_rg_round.sh TAG
TAG=${1:-rg}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_recursive.py -x -q > $OUT/tests.log 2>&1; echo "pytest rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/scripts/gpu_rg_bench.py > $OUT/bench.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
grep rg_bench $OUT/bench.log
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv && rm -rf $OUT/prof
cut -c1-120 $OUT/kernel_stats.csv | head -14
