#!/bin/bash
# Second half of a recorded round (after scripts/gpu_round.sh TAG): the recursive mode's kernel stats and PMC passes,
# the randomised parity campaigns and the search campaigns in both blur modes, other frame sizes, the batch demo,
# a two-rank gloo rehearsal of bench.py's N > 1 path.   Usage (GPU box, repo root): scripts/gpu_round4b.sh TAG
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
export OMP_NUM_THREADS=16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rgprof -- python3 $GRAFT_REPO_ROOT/scripts/gpu_rg_bench.py > $OUT/rg_bench.log 2>&1; echo "rg rocprof rc=$?"
cd $GRAFT_REPO_ROOT
grep rg_bench $OUT/rg_bench.log
cp $(find $OUT/rgprof -name "*kernel_stats.csv" | head -1) $OUT/rg_kernel_stats.csv && rm -rf $OUT/rgprof
scripts/gpu_pmc_sets.sh ${TAG}_rgpmc scripts/gpu_rg_bench.py > $OUT/rg_pmc.log 2>&1; cp gpurun_out/${TAG}_rgpmc/pmc_summary.txt $OUT/rg_pmc_summary.txt; tail -3 $OUT/rg_pmc.log | cut -c1-120
timeout -k 10 300 python tests/tools/gpu_fuzz.py 3000 12345 recursive > $OUT/fuzz_3000_recursive.log 2>&1; tail -1 $OUT/fuzz_3000_recursive.log
timeout -k 10 400 python tests/tools/gpu_fuzz.py 6000 777 > $OUT/fuzz_6000_entrypoints.log 2>&1; tail -1 $OUT/fuzz_6000_entrypoints.log
timeout -k 10 400 python tests/tools/gpu_search_campaign.py 60 recursive > $OUT/search_campaign_360_recursive.log 2>&1; tail -1 $OUT/search_campaign_360_recursive.log
timeout -k 10 400 python tests/tools/gpu_search_campaign.py 60 > $OUT/search_campaign_360.log 2>&1; tail -1 $OUT/search_campaign_360.log
# the same campaigns with the probes made by libavif's C API under the reference's calls and defaults, the HIP side on the CLI / batch path's search
timeout -k 10 400 python tests/tools/gpu_search_campaign.py 60 recursive bridge > $OUT/search_campaign_360_recursive_bridge.log 2>&1; tail -1 $OUT/search_campaign_360_recursive_bridge.log
timeout -k 10 400 python tests/tools/gpu_search_campaign.py 60 fir bridge > $OUT/search_campaign_360_bridge.log 2>&1; tail -1 $OUT/search_campaign_360_bridge.log
for wh in "7680 4320" "1920 1080" "512 512"; do
  set -- $wh
  timeout -k 10 300 python bench.py --width $1 --height $2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d.get('recursive_blur_mode',{})
print('$1x$2: value', d['value'], 'MP/s  ms_per_step', d['ms_per_step'], ' one stream', d.get('score_roofline',{}).get('ms_per_score_one_stream'), ' cached FIR pass', d.get('cached_reference',{}).get('ms_per_score'), ' recursive cached pass', r.get('cached_reference',{}).get('ms_per_pass'), ' recursive pair', r.get('ms_per_score'))" | tee -a $OUT/sizes.log
done
timeout -k 10 900 scripts/gpu_batch_demo.sh 96 > $OUT/batch_demo.log 2>&1; tail -32 $OUT/batch_demo.log
timeout -k 10 200 scripts/gpu_host_timeline.sh > $OUT/host_timeline.log 2>&1; grep -E "^==|process wall" $OUT/host_timeline.log
OAVIF_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err; echo "bench n2 gloo rc=$?"; cut -c1-300 $OUT/bench_n2_gloo.json
