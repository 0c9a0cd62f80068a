#!/bin/bash
# Round 6, second GPU session: the hipGraph A/B (scripts/gpu_graph_ab.py) and a short bench line with the per-kernel
# timestamps.     Usage (repo root on the box): scripts/gpu_round6b.sh [TAG]
TAG=${1:-r06b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 500 python3 scripts/gpu_graph_ab.py > $OUT/graph_ab.log 2> $OUT/graph_ab.err; echo "graph A/B rc=$?"
cat $OUT/graph_ab.log; tail -5 $OUT/graph_ab.err
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_short.json 2> $OUT/bench_short.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_short.json").read().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
for r in d["by_resolution"].get("sizes", []): print(json.dumps(r))
print(d["by_resolution"].get("seconds"), d["by_resolution"].get("error"))
k = d["recursive_blur_mode"]["kernels"]
print({x: k[x] for x in k if x not in ("kernels", "note", "source")})
print([(x["kernel"], x["ms"], x.get("ms_rocprofv3_committed")) for x in k["kernels"]])
PY
