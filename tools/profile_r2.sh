#!/bin/bash
# Round-2 profile set (run on the GPU box through gpurun; summaries are copied into profiles/ afterwards).
#   1. rocprofv3 --kernel-trace --stats of the bench command (per-kernel totals, csv)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the dominant conv kernel alone (tools/conv_only.py)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-rooflines > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/conv_only.py 8192 f16s > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/conv_only.py 8192 f16s > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $O/pmc_fetch FETCH_SIZE k_conv3x3_f16s
python3 $R/tools/pmc_summary.py $O/pmc_write WRITE_SIZE k_conv3x3_f16s
find $O -name "*kernel_stats.csv" | head -3
# 3. BASELINE configs[0] (8 games, 25 sims; eager compacted tick) under the kernel trace: GPU-busy time per rollout tick
rocprofv3 --kernel-trace --stats --output-format csv -d $O/small -- python3 $R/bench.py --games 8 --breadth 25 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --no-conv-timing > $O/small_under_rocprof.json 2> /dev/null
find $O -name "*kernel_stats.csv"
# 4. HBM traffic of the engine kernels (262 144 games): FETCH_SIZE / WRITE_SIZE passes of tools/engine_only.py
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/eng_fetch -- python3 $R/tools/engine_only.py 262144 > $O/engine_only.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/eng_write -- python3 $R/tools/engine_only.py 262144 > /dev/null 2>&1
for k in k_step k_clone k_observe; do python3 $R/tools/pmc_summary.py $O/eng_fetch FETCH_SIZE $k; python3 $R/tools/pmc_summary.py $O/eng_write WRITE_SIZE $k; done
