cd $GRAFT_REPO_ROOT
export OAVIF_RG_LDS_H=30000
scripts/gpu_rg_exp.sh r4b h_chain_instr base_instr
export OAVIF_RG_LDS_H=0
export OAVIF_RG_LDS_V=70000
scripts/gpu_rg_exp.sh r4b2 base_instr
