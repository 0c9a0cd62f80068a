#!/bin/bash
# SQ / TCC counter passes over the 4K kernel micro-bench (one counter set per rocprofv3 run).
# Usage (GPU box, repo root): scripts/gpu_pmc_sets.sh TAG [workload script, default scripts/gpu_kbench.py]
#   -> gpurun_out/TAG/pmc_summary.txt
TAG=${1:-pmc}
WORK=${2:-scripts/gpu_kbench.py}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INSTS_VALU_INT32" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/set$i -- python3 $GRAFT_REPO_ROOT/$WORK > $OUT/set$i.log 2>&1
  echo "pmc set $i rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT/set*/ > $OUT/pmc_summary.txt
rm -rf $OUT/set*/
head -120 $OUT/pmc_summary.txt
