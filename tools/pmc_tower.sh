#!/bin/bash
# SQ counters of the conv launches of a whole Q-net tower (one --pmc pass per counter group, --kernel-trace only beside it; the
# program itself after "--"; run through gpurun):
#   pmc_tower.sh <tag> <algo f16s|bf16|f16a> <games> <board 11|19>   -> gpurun_out/pmc_<tag>/ + gpurun_out/pmc_<tag>.json
# (replaces pmc_conv_r4.sh = "r4 f16s 2300 11" and pmc_a16.sh = "<tag> bf16 500 19")
R=${GRAFT_REPO_ROOT:-/root/repo}
T="$1"; export SNK_CONV_ALGO=${2:-f16s}; G=${3:-2300}; B=${4:-11}
O="$R/gpurun_out/pmc_$T"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$O/$tag" -- python3 "$R/tools/tower_only.py" $G 2 $B > "$O/$tag.log" 2>&1 || { echo "pass $tag failed"; tail -5 "$O/$tag.log"; exit 1; }
    tail -1 "$O/$tag.log"
done
cd $R && python3 tools/pmc_collect.py "$O" "$R/gpurun_out/pmc_$T.json" "k_conv3x3" "tools/pmc_tower.sh $T $SNK_CONV_ALGO $G $B: tools/tower_only.py $G 2 $B with SNK_CONV_ALGO=$SNK_CONV_ALGO (two forwards of the tower on mid-game observations, one chunk)" SQ_
