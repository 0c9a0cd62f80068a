cd /tmp; export TMPDIR=/tmp OAVIF_RG_INSTR=1
for cfg in "0 0" "0 70000" "0 40000" "30000 0" "60000 0" "100000 0"; do set -- $cfg
  export OAVIF_RG_LDS_H=$1 OAVIF_RG_LDS_V=$2
  echo "== LDS_H=$1 LDS_V=$2"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$1_$2 -- python3 $GRAFT_REPO_ROOT/scripts/gpu_rg_bench.py 2>&1 | grep "rg_bench:"
  grep -h "k_rg_h<false, false>\|k_rg_v<false>" $(find /tmp/p_$1_$2 -name "*kernel_stats.csv") | cut -d, -f1-4
done
