#!/bin/bash
# scripts/gpu_sweep2.sh: scale-0 segment x tail segment grid
for s0 in 0 68 108 135; do for t in 40 45 54; do
  env OAVIF_AMD_SEG_ROWS=$s0 OAVIF_AMD_SEG_ROWS_TAIL=$t python bench.py --no-cpu-baseline --steps 300 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seg0=$s0 tail=$t', d['value'], d['ms_per_step'], d['stages_ms']['march'])"
done; done
