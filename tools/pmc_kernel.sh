#!/bin/bash
# SQ instruction / cycle counters of one engine kernel: pmc_kernel.sh <tag> <kernel substring> <python tool> [tool args]
# (one --pmc pass per counter group; run through gpurun; prints per-dispatch averages)
R=${GRAFT_REPO_ROOT:-/root/repo}
T="$1"; K="$2"; shift 2
O="$R/gpurun_out/pmc_$T"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$O/$tag" -- python3 "$R/tools/$1" "${@:2}" > "$O/$tag.log" 2>&1
    for c in $grp; do python3 "$R/tools/pmc_summary.py" "$O/$tag" $c "$K"; done
done
