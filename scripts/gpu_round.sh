#!/bin/bash
# One recorded GPU-box round: smoke, GPU tests, bench, rocprof kernel stats, PMC traffic.
# Usage (on the box, from the repo root): scripts/gpu_round.sh TAG     (outputs in gpurun_out/TAG/)
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
export OMP_NUM_THREADS=16
mkdir -p $OUT
python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
tail -2 $OUT/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --streams 1 > $OUT/prof.log 2>&1; echo "rocprof stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/scripts/gpu_kbench.py > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/scripts/gpu_kbench.py > $OUT/pmc_write.log 2>&1; echo "pmc write rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_calib -- $GRAFT_REPO_ROOT/scripts/ubench/fetch_calib > $OUT/pmc_calib.log 2>&1; echo "pmc calib rc=$?"
cd $GRAFT_REPO_ROOT
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 scripts/pmc_summary.py $OUT/pmc_fetch/ $OUT/pmc_write/ $OUT/pmc_calib/ > $OUT/pmc_summary.txt
cat $OUT/kernel_stats.csv | cut -c1-150
cat $OUT/pmc_summary.txt
rm -rf $OUT/prof/*/*kernel_trace.csv $OUT/pmc_*/*/*kernel_trace.csv
