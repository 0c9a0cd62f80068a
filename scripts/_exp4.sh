cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for d in m_base m_nt m_a6 m_a8 m_nt_a8; do
  OAVIF_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_ablate/$d/liboavif_hip.so timeout -k 10 120 python3 scripts/gpu_refblur_bench.py 2>/dev/null | grep refblur_bench | sed "s|^|$d  |"
done; done
for rep in 1 2; do
for d in v_pf4 v_pf6 v_pf8; do
  OAVIF_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_ablate/$d/liboavif_hip.so timeout -k 10 120 python3 scripts/gpu_rg_bench.py 2>/dev/null | grep "rg_bench:" | sed "s|^|$d  |"
done; done
