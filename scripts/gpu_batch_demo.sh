#!/bin/bash
# BASELINE configs[3] on one GPU's shard: N synthetic 1920x1080 PNGs through the batch driver,
# sequential (as scripts/measure.py runs oavif) vs worker threads.
N=${1:-32}
D=/tmp/oavif_batch_demo; rm -rf $D; mkdir -p $D/imgs
python - <<PY
import sys; sys.path.insert(0, "$GRAFT_REPO_ROOT")
from PIL import Image
from oavif_amd import synth
for i in range($N):
    Image.fromarray(synth.make_ref(1920, 1080, 900 + i)).save("$D/imgs/img%03d.png" % i, compress_level=1)
PY
for w in 1 16; do
  echo "== workers=$w"
  python -m oavif_amd.batch $D/imgs $D/out_$w.csv --workers $w --out-dir $D/o$w 2>/dev/null | grep -E "Images:|Total wall|Throughput|Average encoding|Average passes"
done
python - <<PY
import csv
a = [(r[0], r[2], r[6]) for r in csv.reader(open("$D/out_1.csv"))]
b = [(r[0], r[2], r[6]) for r in csv.reader(open("$D/out_16.csv"))]
print("identical results:", a == b)
PY
