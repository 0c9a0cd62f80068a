#!/bin/bash
# Sweep an environment knob over bench.py: scripts/gpu_sweep.sh VAR v1 v2 ...
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --no-cpu-baseline --steps 300 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['ms_per_step'], d['stages_ms'])"
done
