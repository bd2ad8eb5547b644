#!/bin/bash
# Round-4 profile set (run on the GPU box through gpurun; summaries are copied into profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of the bench command (per-kernel totals, csv)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of whole forwards on mid-game observations (tools/tower_only.py), turned
#      into a traffic profile that names the sources it was measured on (tools/conv_traffic.py)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-rooflines > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "bench kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/tower_only.py 2300 3 > $O/tower_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/tower_only.py 2300 3 > $O/tower_write.log 2>&1
tail -1 $O/tower_fetch.log
cd $R && python3 tools/conv_traffic.py $O/pmc_fetch $O/pmc_write $O/tower_fetch.log $O/r4_conv_rect_traffic.json "round 4: sub-rectangle form of layers 0-5 (readers take the pixels outside their producer's rectangle from the producer's L2-resident background image; only layer 5 fills the canvas), full form of layers 6-7 (the last with the head's 1x1 stage: its output never goes to HBM)"
find $O -name "*kernel_stats.csv" | head -3
