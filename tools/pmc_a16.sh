#!/bin/bash
# SQ counters of the conv launches of the tower with 16-bit activations on BASELINE configs[4]'s canvas (37 x 37, 20 layers):
#   pmc_a16.sh <tag> [bf16|f16a] [games 500]      -> gpurun_out/pmc_<tag>/ + gpurun_out/pmc_<tag>.json (tools/pmc_collect.py)
# one --pmc pass per counter group, the program itself after "--" (run through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
T="$1"; export SNK_CONV_ALGO=${2:-bf16}; G=${3:-500}
O="$R/gpurun_out/pmc_$T"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$O/$tag" -- python3 "$R/tools/tower_only.py" $G 2 19 > "$O/$tag.log" 2>&1
    tail -1 "$O/$tag.log"
done
cd $R && python3 tools/pmc_collect.py "$O" "$R/gpurun_out/pmc_$T.json" "k_conv3x3" "tools/pmc_a16.sh $T: tools/tower_only.py $G 2 19 with SNK_CONV_ALGO=$SNK_CONV_ALGO (two forwards of the 20-layer tower on mid-game 19x19 / 8-snake observations, one chunk)" SQ_
