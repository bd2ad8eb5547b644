#!/bin/bash
# SQ instruction / cycle counters of k_step at 262 144 games (one --pmc pass per group; run through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_step
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$tag -- python3 $R/tools/step_time.py 262144 > $O/$tag.log 2>&1
    for c in $grp; do python3 $R/tools/pmc_summary.py $O/$tag $c k_step; done
done
