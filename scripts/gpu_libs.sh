#!/bin/bash
# bench several builds of the library: scripts/gpu_libs.sh name1 name2 ... (gpurun_ablate/lib_NAME.so; "base" = product)
for v in "$@"; do
  if [ "$v" = base ]; then L=""; else L=$PWD/gpurun_ablate/lib_$v.so; fi
  env OAVIF_AMD_LIB=$L python bench.py --no-cpu-baseline --steps 400 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['stages_ms']['march'])"
done
