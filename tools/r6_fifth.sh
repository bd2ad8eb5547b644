#!/bin/bash
# Round 6, fifth GPU call: the hazard micro-experiment beside a stress kernel, the round's bench lines, the round's fuzz / soak logs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
timeout -k 10 200 tools/micro/store_hazard 2 1 > $O/store_hazard_stress.json 2> $O/store_hazard_stress.err; echo "store_hazard stress rc=$?"
timeout -k 10 900 bash tools/bench_set.sh r6 2>&1 | tail -12
timeout -k 10 900 bash tools/logs.sh r6 2>&1 | tail -30
