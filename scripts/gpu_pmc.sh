#!/bin/bash
# PMC pass(es) on the kernel micro-bench.  Usage: scripts/gpu_pmc.sh TAG "CTR1 CTR2 ..." ["CTR..."]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$i -- python3 $GRAFT_REPO_ROOT/scripts/gpu_kbench.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$i.log 2>&1
  echo "pmc set $i rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_*/ 
