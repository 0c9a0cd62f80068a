#!/bin/bash
# MP/s by frame size and stream count (inputs resident in HBM); run on the GPU box.
for wh in "512 512" "1920 1080" "3840 2160" "7680 4320"; do set -- $wh
 for n in 1 2 3; do timeout -k 10 120 python bench.py --streams $n --no-cpu-baseline --steps 600 --width $1 --height $2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1x$2 streams', d['config']['streams_per_gpu'], d['value'], 'MP/s', round(d['ms_per_step']*1e3,1), 'us/step', d['stages_ms'])" || exit 1; done; done
