#!/bin/bash
# A/B of a library variant on the judged workload (default f16s bench), alternating runs in one call: r5_ab_bench.sh <variant> [steps]
R=${GRAFT_REPO_ROOT:-/root/repo}; V=$1; K=${2:-6}
O=$R/gpurun_out/r5; mkdir -p $O; cd $R
SE=$R/alphasnake-zero_amd/snake_engine
SNK_LIB_PATH=$SE/libsnake_engine_$V.so python3 -m pytest tests/test_net_gpu.py tests/test_rect_conv_gpu.py -x -q -m gpu -k "full_net or rect or conv3x3_layer or fused_head" 2>&1 | tail -2
for rep in 1 2; do
  for lib in "" "_$V"; do
    SNK_LIB_PATH=$SE/libsnake_engine$lib.so python3 bench.py --steps $K --warmup 2 --no-cpu-baseline --no-kernel-rooflines > $O/ab_bench$lib.$rep.json 2> $O/ab_bench$lib.$rep.err
    python3 -c "import json; d=json.loads(open('$O/ab_bench$lib.$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('lib${lib:-_default} run $rep:', round(d['value'],1), 'conv TF', round(r['achieved'],1), 'MHz', round(r['clock_mhz']['median']))"
  done
done
