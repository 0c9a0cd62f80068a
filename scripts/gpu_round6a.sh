#!/bin/bash
# Round 6, first GPU session: does the new plumbing work on the box?  smoke, the IPC-mode probe, a
# short bench line (by_resolution, live recursive-pass stage times), the bare two-rank commands (gloo rehearsal -> rc 0,
# RCCL on a one-GPU box -> rc 4), then the GPU tests.      Usage (repo root on the box): scripts/gpu_round6a.sh [TAG]
TAG=${1:-r06a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export OMP_NUM_THREADS=16
python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
{
  echo "# scripts/ubench/ipc_probe on $(hostname), $(date -u +%FT%TZ); the box's own environment has HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY-(unset)}"
  env -u HSA_ENABLE_IPC_MODE_LEGACY timeout -k 5 60 scripts/ubench/ipc_probe; echo "rc=$?"
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 5 60 scripts/ubench/ipc_probe; echo "rc=$?"
  HSA_ENABLE_IPC_MODE_LEGACY=1 timeout -k 5 60 scripts/ubench/ipc_probe; echo "rc=$?"
} > $OUT/ipc_mode_probe.txt 2>&1
cat $OUT/ipc_mode_probe.txt
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err; echo "bench (driver shape) rc=$?"
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_driver_shape.json").read().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
print(json.dumps(d.get("by_resolution"), indent=0)[:3000])
print(json.dumps(d["recursive_blur_mode"].get("kernels"), indent=0)[:2500])
print(json.dumps(d["collective"].get("ranks"), indent=0)[:1200])
PY
OAVIF_BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_bare_n2_gloo.json 2> $OUT/bench_bare_n2_gloo.err; echo "bare --gpus 2 over gloo rc=$?"
tail -c 600 $OUT/bench_bare_n2_gloo.err
timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_bare_n2_nccl.json 2> $OUT/bench_bare_n2_nccl.err; echo "bare --gpus 2 over RCCL on this box rc=$? (4 = refused: one GPU)"
tail -c 600 $OUT/bench_bare_n2_nccl.err
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
