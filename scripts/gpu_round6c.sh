#!/bin/bash
# Second half of round 6's records (after scripts/gpu_round.sh r06), on the final sources: the recursive mode's kernel
# stats (the rocprofv3 cross-check of the live per-kernel times on the bench line) and counter passes, the hipGraph A/B
# (profiles/r06_graph_ab.log), one randomised parity campaign per blur mode (every score kernel now goes through launch()),
# scripts/gpu_scale.sh with the BARE commands as far as this box has devices, and the N > 1 rehearsals over gloo: bare
# `bench.py --gpus 4` (the command starts its own ranks) and the same two ranks under torch.distributed.run.
# Usage (GPU box, repo root): scripts/gpu_round6c.sh TAG
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
export OMP_NUM_THREADS=16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (the first half's kernel trace was taken before bench.py had --no-by-resolution: its averages mixed four frame sizes)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 512 --warmup 64 --no-cpu-baseline --streams 1 --no-by-resolution > $OUT/prof.log 2>&1; echo "rocprof stats rc=$?"
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv && rm -rf $OUT/prof
cut -c1-150 $OUT/kernel_stats.csv | head -9
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rgprof -- python3 $GRAFT_REPO_ROOT/scripts/gpu_rg_bench.py > $OUT/rg_bench.log 2>&1; echo "rg rocprof rc=$?"
cd $GRAFT_REPO_ROOT
grep rg_bench $OUT/rg_bench.log
cp $(find $OUT/rgprof -name "*kernel_stats.csv" | head -1) $OUT/rg_kernel_stats.csv && rm -rf $OUT/rgprof
scripts/gpu_pmc_sets.sh ${TAG}_rgpmc scripts/gpu_rg_bench.py > $OUT/rg_pmc.log 2>&1; cp gpurun_out/${TAG}_rgpmc/pmc_summary.txt $OUT/rg_pmc_summary.txt; tail -3 $OUT/rg_pmc.log | cut -c1-120
timeout -k 10 500 python3 scripts/gpu_graph_ab.py > $OUT/graph_ab.log 2> $OUT/graph_ab.err; echo "graph A/B rc=$?"; tail -12 $OUT/graph_ab.log
timeout -k 10 300 python tests/tools/gpu_fuzz.py 3000 2026 recursive > $OUT/fuzz_3000_recursive.log 2>&1; tail -1 $OUT/fuzz_3000_recursive.log
timeout -k 10 400 python tests/tools/gpu_fuzz.py 6000 556 > $OUT/fuzz_6000_entrypoints.log 2>&1; tail -1 $OUT/fuzz_6000_entrypoints.log
timeout -k 10 900 scripts/gpu_scale.sh ${TAG}_scale 96 400 > $OUT/scale.log 2>&1; echo "scale rc=$?"; cp profiles/scale.json $OUT/scale.json
OAVIF_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 4 --steps 20 --warmup 5 > $OUT/bench_n4_gloo_bare.json 2> $OUT/bench_n4_gloo_bare.err; echo "bare bench n4 gloo rc=$?"; cut -c1-300 $OUT/bench_n4_gloo_bare.json
OAVIF_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_n2_gloo_torchrun.json 2> $OUT/bench_n2_gloo_torchrun.err; echo "torchrun bench n2 gloo rc=$?"; cut -c1-300 $OUT/bench_n2_gloo_torchrun.json
timeout -k 10 400 python scripts/gpu_soak.py 150 > $OUT/soak.log 2>&1; echo "soak rc=$?"; tail -1 $OUT/soak.log
