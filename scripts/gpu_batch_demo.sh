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
  python -m oavif_amd.batch $D/imgs $D/out_$w.csv --workers $w --out-dir $D/o$w 2>/dev/null | grep -E "Images:|Total wall|Throughput|Average encoding|Average passes|Host cores"
done
echo "== --exec: one process per image, as scripts/measure.py runs an oavif binary (measure.py:151-158), with this repository's compiled C host (oavif_amd/lib/oavif_host) as that binary; then the same with 4 at a time (this pool allows a job 6 processes on a GPU at once)"
for w in 1 4; do
  python -m oavif_amd.batch $D/imgs $GRAFT_REPO_ROOT/oavif_amd/lib/oavif_host $D/out_x$w.csv --exec --workers $w --out-dir $D/ox$w 2>/dev/null | grep -E "Images:|Total wall|Throughput|Average encoding"
done
echo "== workers=default, FIR blur mode (OAVIF_SSIMU2_BLUR=fir; the runs above and below use the search path's default, the published recursion)"
OAVIF_SSIMU2_BLUR=fir python -m oavif_amd.batch $D/imgs $D/out_r.csv --out-dir $D/or 2>/dev/null | grep -E "Images:|Total wall|Throughput|Average passes|Worker threads|Host cores"
for k in 2 4; do
echo "== --procs-per-gpu $k: $k ranks on GPU 0 (the gather over gloo), pinned to disjoint host cores, largest file first"
python -m torch.distributed.run --nnodes=1 --nproc-per-node $k --master-addr 127.0.0.1 --master-port 2953$k \
  -m oavif_amd.batch $D/imgs $D/out_${k}r.csv --out-dir $D/o${k}r --procs-per-gpu $k 2>/dev/null | grep -E "Images:|Ranks|Total wall|Throughput|Host cores|Worker threads"
done
python - <<PY
import csv
a = [(r[0], r[2], r[6]) for r in csv.reader(open("$D/out_1.csv"))]
b = [(r[0], r[2], r[6]) for r in csv.reader(open("$D/out_16.csv"))]
c = [(r[0], r[2], r[6]) for r in csv.reader(open("$D/out_2r.csv"))]
d = [(r[0], r[2], r[6]) for r in csv.reader(open("$D/out_4r.csv"))]
x = [(r[0], r[2], r[6]) for r in csv.reader(open("$D/out_x4.csv"))]
print("identical results (1 worker, 16 workers, 2 ranks, 4 ranks, one C-host process per image):", a == b == c == d == x)
PY
