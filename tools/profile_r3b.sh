#!/bin/bash
# Round-3 profile set after the sub-rectangle form went in (run on the GPU box through gpurun; summaries are copied into
# profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of the bench command (per-kernel totals, csv)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of whole forwards on mid-game observations (tools/tower_only.py):
#      every conv launch of the tower, sub-rectangle and full layers
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r3b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-rooflines > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "bench kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/tower_only.py 2300 3 > $O/tower_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/tower_only.py 2300 3 > $O/tower_write.log 2>&1
tail -1 $O/tower_fetch.log
for k in "k_conv3x3_f16s_rect<1" "k_conv3x3_f16s_rect<2" "k_conv3x3_f16s<7, 1" "k_conv3x3_f16s<7, 3" "k_stem_conv_mfma"; do
  python3 $R/tools/pmc_summary.py $O/pmc_fetch FETCH_SIZE "$k"
  python3 $R/tools/pmc_summary.py $O/pmc_write WRITE_SIZE "$k"
done
find $O -name "*kernel_stats.csv"
