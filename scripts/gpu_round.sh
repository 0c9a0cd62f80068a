#!/bin/bash
# One recorded GPU-box round: smoke, GPU tests, rocprof kernel stats, PMC counters, VALU
# microbenchmark, counters.json, and last the bench (so that its counter-derived fields are this build's).   Usage (on the box, repo root): scripts/gpu_round.sh TAG
# Outputs in gpurun_out/TAG/; the summaries to keep are then copied into profiles/ by hand
# (kernel_stats.csv -> profiles/TAG_kernel_stats.csv, pmc_summary.txt, counters.json, bench.json).
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
export OMP_NUM_THREADS=16
mkdir -p $OUT
python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
tail -2 $OUT/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
# per-kernel durations with the scores of one stream never overlapping (two streams stretch them); 3840x2160 launches only
# (--no-by-resolution: the other sizes of the line's by_resolution extra would be averaged into the same kernel names)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 512 --warmup 64 --no-cpu-baseline --streams 1 --no-by-resolution > $OUT/prof.log 2>&1; echo "rocprof stats rc=$?"
cd $GRAFT_REPO_ROOT
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cut -c1-150 $OUT/kernel_stats.csv
rm -rf $OUT/prof
scripts/gpu_pmc_sets.sh $TAG
timeout -k 10 200 scripts/ubench/valu_rate > $OUT/valu_rate.txt
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_calib -- $GRAFT_REPO_ROOT/scripts/ubench/fetch_calib > $OUT/pmc_calib.log 2>&1; echo "pmc calib rc=$?"
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT/pmc_calib/ > $OUT/pmc_calib_summary.txt; rm -rf $OUT/pmc_calib
# the counter files under the names they are committed under, so that bench.py (which reads
# profiles/counters.json and checks its source hash) runs LAST, on counters of these very sources
cp $OUT/pmc_summary.txt profiles/${TAG}_pmc_summary.txt
cp $OUT/valu_rate.txt profiles/${TAG}_valu_rate.txt
python3 scripts/make_counters_json.py profiles/${TAG}_pmc_summary.txt profiles/${TAG}_valu_rate.txt > $OUT/counters.json
cp $OUT/counters.json profiles/counters.json
cat $OUT/counters.json | head -12
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json
