#!/bin/bash
# Copy the summaries of a recorded round from gpurun_out/TAG/ (scripts/gpu_round.sh TAG + scripts/gpu_round5b.sh TAG, merged
# back by gpurun) into profiles/ under the names they are committed under.   Usage (repo root): scripts/collect_round_records.sh r05
set -e
TAG=${1:-r05}
O=gpurun_out/$TAG
cp $O/kernel_stats.csv profiles/${TAG}_kernel_stats.csv
cp $O/pmc_summary.txt profiles/${TAG}_pmc_summary.txt
cp $O/valu_rate.txt profiles/${TAG}_valu_rate.txt
cp $O/pmc_calib_summary.txt profiles/${TAG}_pmc_calib_summary.txt
cp $O/counters.json profiles/counters.json
grep '^{' $O/bench.json | tail -1 > profiles/${TAG}_bench.json
for f in rg_kernel_stats.csv rg_pmc_summary.txt fuzz_3000_recursive.log fuzz_6000_entrypoints.log pinned_ab.json batch_demo.log graph_ab.log soak.log; do
  [ -f $O/$f ] && cp $O/$f profiles/${TAG}_$f
done
[ -f $O/scale.json ] && cp $O/scale.json profiles/scale.json
[ -f $O/bench_n4_gloo.json ] && grep '^{' $O/bench_n4_gloo.json | tail -1 > profiles/${TAG}_bench_n4_gloo_rehearsal.json
# round 6: the bare command (bench.py launches its own ranks) and the same ranks under torch.distributed.run
[ -f $O/bench_n4_gloo_bare.json ] && grep '^{' $O/bench_n4_gloo_bare.json | tail -1 > profiles/${TAG}_bench_n4_gloo_bare_rehearsal.json
[ -f $O/bench_n2_gloo_torchrun.json ] && grep '^{' $O/bench_n2_gloo_torchrun.json | tail -1 > profiles/${TAG}_bench_n2_gloo_torchrun_rehearsal.json
python3 - <<PY
import json
d = json.load(open("profiles/${TAG}_bench.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline.frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"],
      "counters stale", d["roofline"].get("counters", {}).get("stale"), "cpu_baseline", d.get("cpu_baseline", {}).get("value"),
      "search default ms", d["default_search_mode"]["ms_per_pass"], "collective", d["collective"]["backend"], d["collective"]["world_size"], d["collective"]["problems"])
PY
