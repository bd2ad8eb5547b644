R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_r4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$O/$tag" -- python3 "$R/tools/tower_only.py" 2300 2 > "$O/$tag.log" 2>&1
    for c in $grp; do for k in "k_conv3x3_f16s_rect<1" "k_conv3x3_f16s<7, 1"; do python3 "$R/tools/pmc_summary.py" "$O/$tag" $c "$k"; done; done
done
