#!/bin/bash
# Where a one-image run of the compiled host (oavif_amd/lib/oavif_host) spends its wall time: the phase timeline
# of OAVIF_HOST_TIMES=1 plus the wall clock around the whole process, for a 1080p and a 4K PNG, in-order runs.
# Usage (on the box, repo root): scripts/gpu_host_timeline.sh
D=/tmp/oavif_host_timeline; rm -rf $D; mkdir -p $D
export OAVIF_LIBAVIF=$(python -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); from oavif_amd import avif_bridge as a; print(a._find_library())")
python - <<PY
import sys; sys.path.insert(0, "$GRAFT_REPO_ROOT")
from PIL import Image
from oavif_amd import synth
Image.fromarray(synth.make_ref(1920, 1080, 900)).save("$D/hd.png", compress_level=1)
Image.fromarray(synth.make_ref(3840, 2160, 901)).save("$D/uhd.png", compress_level=1)
PY
for img in hd uhd; do
  for rep in 1 2 3; do
    echo "== $img run $rep"
    s=$(date +%s.%N)
    OAVIF_HOST_TIMES=1 $GRAFT_REPO_ROOT/oavif_amd/lib/oavif_host $D/$img.png $D/$img.avif 2>&1 | grep -v "^.\[31m"
    e=$(date +%s.%N)
    python -c "print('   process wall: %.1f ms' % (($e - $s) * 1e3))"
  done
done
