#!/bin/bash
# The scaling record of BASELINE.json (north_star: "reported at 1, 2, 4 and 8 GPUs ... >= 6x batch throughput at 8
# GPUs"; configs[3]: "Batch of 256x 1920x1080 PNG via scripts/measure.py, per-image shard over 8 MI355X, RCCL
# gather"), runnable as it is on a node with several MI355X -- nothing in this repository has run on more than one.
#
#   scripts/gpu_scale.sh [TAG] [IMAGES] [STEPS]        (repo root; TAG default "scale", IMAGES 256, STEPS 400)
#
# For N in 1 2 4 8 (as far as the node has devices): the BARE commands `python3 bench.py --gpus N` and `python3 -m
# oavif_amd.batch --gpus N ...` (one process per GPU over RCCL; the command starts its own ranks, oavif_amd/launch.py --
# `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N` runs the
# same ranks; which of the two forms a driver uses is the driver's business, both are tested) over IMAGES synthetic
# 1920x1080 PNGs for the batch (the scripts/measure.py counterpart: images dealt largest first, one RCCL all_gather of
# the records).  The environment is handed to the ranks as it is: nothing is set here (HSA_ENABLE_IPC_MODE_LEGACY: see
# profiles/r06_ipc_mode_probe.txt and DESIGN.md section 6).  Every run's
# JSON carries its `collective` record (backend, world size, per rank the device / PCI bus id / NUMA node / pinned
# cores, RCCL version); a run whose ranks share a GPU exits with rc 4 and is recorded as refused.
# Output: gpurun_out/TAG/*.json and profiles/scale.json (scripts/make_scale_json.py: MP/s, speed-up over N = 1,
# images/s, speed-up, the collective records).  No number is claimed by this script's existence.
set -u
TAG=${1:-scale}
IMAGES=${2:-256}
STEPS=${3:-400}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
NDEV=$(python3 -c 'import torch; print(torch.cuda.device_count())')
echo "devices visible: $NDEV" | tee "$OUT/devices.txt"
IMG=/tmp/oavif_scale_imgs_$$
mkdir -p "$IMG"
python3 - <<PY
import sys; sys.path.insert(0, "$ROOT")
from PIL import Image
from oavif_amd import synth
# content of four kinds, so that the searches do not all end on their first pass (the plain synthetic frame does at
# target 80): as it is / with sensor-like noise (scores low at the first guess: the search walks up) / smoothed (scores
# high: walks down) / noise on a smoothed frame -- the passes per image then spread over 1..4 like a folder of photographs
for i in range($IMAGES):
    f = synth.make_ref(1920, 1080, 3000 + i)
    k = i % 4
    if k == 1:
        f = synth.distort(f, "noise", 1 + (i // 4) % 3, seed=i)
    elif k == 2:
        f = synth.distort(f, "blur", 1 + (i // 4) % 3)
    elif k == 3:
        f = synth.distort(synth.distort(f, "blur", 2), "noise", (i // 4) % 2, seed=i)
    Image.fromarray(f).save("$IMG/img%03d.png" % i, compress_level=1)
PY
for N in 1 2 4 8; do
  if [ "$N" -gt "$NDEV" ]; then echo "N=$N skipped: $NDEV device(s)"; continue; fi
  echo "== bench.py --gpus $N"
  EXTRA=""; [ "$N" -eq 1 ] && EXTRA="--no-cpu-baseline"
  timeout -k 10 900 python3 bench.py --gpus "$N" --steps "$STEPS" --warmup 50 $EXTRA > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err"
  echo "rc=$?" | tee "$OUT/bench_n$N.rc"
  echo "== batch of $IMAGES x 1080p on $N GPU(s)"
  OAVIF_GATHER_ALWAYS=1 timeout -k 10 1800 python3 -m oavif_amd.batch --gpus "$N" "$IMG" "$OUT/batch_n$N.csv" --out-dir "/tmp/oavif_scale_out_$$_$N" \
    --collective-json "$OUT/batch_n$N.json" > "$OUT/batch_n$N.log" 2> "$OUT/batch_n$N.err"
  echo "rc=$?" | tee "$OUT/batch_n$N.rc"
  grep -E "Images:|Ranks|Total wall|Throughput:|Average passes|Collective:" "$OUT/batch_n$N.log"
  rm -rf "/tmp/oavif_scale_out_$$_$N"
done
rm -rf "$IMG"
python3 scripts/make_scale_json.py "$OUT" > profiles/scale.json
cat profiles/scale.json | head -60
