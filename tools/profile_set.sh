#!/bin/bash
# A round's profile set:   profile_set.sh <tag, e.g. r6>
# (run on the GPU box through gpurun; summaries are copied into profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of the bench command (per-kernel totals, csv) + of the configs[4]-shape bf16 bench + of the fit
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of whole forwards on mid-game observations (tools/tower_only.py) for the judged
#      float32-accurate tower (11x11) and for the bf16 tower on the configs[4] canvas, turned into traffic profiles that name the
#      sources they were measured on (tools/conv_traffic.py)
#   3. SQ counters of both towers' conv launches (tools/pmc_tower.sh), on the sources the library was built from
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r6}
O=$R/gpurun_out/prof_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-rooflines > $O/${T}_bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find $O/bench -name "*kernel_stats.csv" | head -1) $O/${T}_bench_kernel_stats.csv; echo "bench kernel-trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_c4 -- python3 $R/bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --conv-algo bf16 --steps 12 --warmup 10 --no-cpu-baseline --no-kernel-rooflines > $O/${T}_bench_config4_shape_bf16_under_rocprof.json 2> $O/bench_c4_under_rocprof.err
cp $(find $O/bench_c4 -name "*kernel_stats.csv" | head -1) $O/${T}_bench_config4_shape_bf16_kernel_stats.csv; echo "configs[4] kernel-trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fit -- python3 $R/tools/fit_time.py 8 > $O/fit_under_rocprof.log 2>&1
cp $(find $O/fit -name "*kernel_stats.csv" | head -1) $O/${T}_fit_kernel_stats.csv; echo "fit kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/tower_only.py 2300 3 > $O/tower_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/tower_only.py 2300 3 > $O/tower_write.log 2>&1
export SNK_CONV_ALGO=bf16
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch16 -- python3 $R/tools/tower_only.py 500 3 19 > $O/tower_fetch16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write16 -- python3 $R/tools/tower_only.py 500 3 19 > $O/tower_write16.log 2>&1
unset SNK_CONV_ALGO
cd $R
python3 tools/conv_traffic.py $O/pmc_fetch $O/pmc_write $O/tower_fetch.log $O/${T}_conv_rect_traffic.json "round ${T}: the judged float32-accurate tower (kernels unchanged since round 4: sub-rectangle form of layers 0-5, full form of layers 6-7, the last with the head's 1x1 stage)"
SNK_CONV_ALGO=bf16 python3 tools/conv_traffic.py $O/pmc_fetch16 $O/pmc_write16 $O/tower_fetch16.log $O/${T}_conv_a16_traffic.json "round ${T}: the bf16 tower on the configs[4] canvas (19x19 board, 37 x 37 observations, 20 layers: 12 in the sub-rectangle form; the last layer's epilogue feeds the head, no float32 activation in HBM); the 16-bit towers' own block frame"
bash tools/pmc_tower.sh ${T}_bf16 bf16 500 19 | tail -4
cp $R/gpurun_out/pmc_${T}_bf16.json $O/${T}_conv_a16_sq_counters.json
bash tools/pmc_tower.sh ${T}_f16s f16s 2300 11 | tail -4
cp $R/gpurun_out/pmc_${T}_f16s.json $O/${T}_conv_f16s_sq_counters.json
ls $O/*.json $O/*.csv
