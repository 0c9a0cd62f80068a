#!/bin/bash
for n in 1 2 3 4; do python bench.py --streams $n --no-cpu-baseline --steps 400 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams', d['config']['streams_per_gpu'], d['value'], d['ms_per_step'])"; done
