#!/bin/bash
# 1080p: scale-0 segment x streams
for s0 in 0 45 68 135; do for n in 2 3 4; do
  env OAVIF_AMD_SEG_ROWS=$s0 python bench.py --no-cpu-baseline --steps 800 --width 1920 --height 1080 --streams $n | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p seg0=$s0 streams=$n', d['value'], d['ms_per_step'], d['stages_ms']['march'])"
done; done
