#!/bin/bash
# Second half of round 5's records (after scripts/gpu_round.sh TAG), on the final sources: the recursive mode's kernel
# stats and counter passes (the search path's default pass), one randomised parity campaign per blur mode over every
# entry point (the pinned-buffer entry included through the soak), the pinned-vs-pageable pass, the batch demo with
# the C host decoding into pinned frames, scripts/gpu_scale.sh as far as this box has devices, and a FOUR-rank gloo
# rehearsal of bench.py's N > 1 path (ranks pinned to disjoint cores, the collective record gathered over the
# group; the pool allows six processes on a card).   Usage (GPU box, repo root): scripts/gpu_round5b.sh TAG
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
export OMP_NUM_THREADS=16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rgprof -- python3 $GRAFT_REPO_ROOT/scripts/gpu_rg_bench.py > $OUT/rg_bench.log 2>&1; echo "rg rocprof rc=$?"
cd $GRAFT_REPO_ROOT
grep rg_bench $OUT/rg_bench.log
cp $(find $OUT/rgprof -name "*kernel_stats.csv" | head -1) $OUT/rg_kernel_stats.csv && rm -rf $OUT/rgprof
scripts/gpu_pmc_sets.sh ${TAG}_rgpmc scripts/gpu_rg_bench.py > $OUT/rg_pmc.log 2>&1; cp gpurun_out/${TAG}_rgpmc/pmc_summary.txt $OUT/rg_pmc_summary.txt; tail -3 $OUT/rg_pmc.log | cut -c1-120
timeout -k 10 300 python tests/tools/gpu_fuzz.py 3000 2025 recursive > $OUT/fuzz_3000_recursive.log 2>&1; tail -1 $OUT/fuzz_3000_recursive.log
timeout -k 10 400 python tests/tools/gpu_fuzz.py 6000 555 > $OUT/fuzz_6000_entrypoints.log 2>&1; tail -1 $OUT/fuzz_6000_entrypoints.log
timeout -k 10 300 python scripts/gpu_pinned_ab.py --out $OUT/pinned_ab.json > $OUT/pinned_ab.log 2>&1; echo "pinned ab rc=$?"
timeout -k 10 900 scripts/gpu_batch_demo.sh 96 > $OUT/batch_demo.log 2>&1; tail -32 $OUT/batch_demo.log
timeout -k 10 900 scripts/gpu_scale.sh ${TAG}_scale 96 400 > $OUT/scale.log 2>&1; echo "scale rc=$?"; cp profiles/scale.json $OUT/scale.json
OAVIF_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 4 --steps 20 --warmup 5 > $OUT/bench_n4_gloo.json 2> $OUT/bench_n4_gloo.err; echo "bench n4 gloo rc=$?"; cut -c1-300 $OUT/bench_n4_gloo.json
