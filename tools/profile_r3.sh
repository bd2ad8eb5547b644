#!/bin/bash
# Round-3 profile set (run on the GPU box through gpurun; summaries are copied into profiles/ afterwards).
#   1. rocprofv3 --kernel-trace --stats of the bench command (per-kernel totals, csv)
#   2. rocprofv3 --kernel-trace --stats of one generation's live training steps (tools/fit_time.py 10)
#   3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the conv kernel alone and of the weight-gradient kernel alone
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-rooflines > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "bench kernel-trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fit -- python3 $R/tools/fit_time.py 10 > $O/fit_under_rocprof.log 2>&1
echo "fit kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/conv_only.py 8192 f16s > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/conv_only.py 8192 f16s > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $O/pmc_fetch FETCH_SIZE k_conv3x3_f16s
python3 $R/tools/pmc_summary.py $O/pmc_write WRITE_SIZE k_conv3x3_f16s
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/wg_fetch -- python3 $R/tools/wgrad_time.py 2048 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/wg_write -- python3 $R/tools/wgrad_time.py 2048 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $O/wg_fetch FETCH_SIZE k_wgrad_f16s
python3 $R/tools/pmc_summary.py $O/wg_write WRITE_SIZE k_wgrad_f16s
find $O -name "*kernel_stats.csv"
