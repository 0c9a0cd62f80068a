#!/bin/bash
# One GPU-box round: smoke, GPU tests, bench, rocprof kernel stats.  Usage: scripts/gpu_round.sh TAG
TAG=${1:-r01}
export OMP_NUM_THREADS=16
mkdir -p gpurun_out
python __graft_entry__.py --smoke > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log
tail -2 gpurun_out/smoke_$TAG.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -15 gpurun_out/pytest_gpu_$TAG.log
timeout -k 10 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head -3
cat $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) | head -20
