#!/bin/bash
# A/B of a library variant against the default library on the 16-bit tower, alternating inside one GPU call: r5_ab16.sh <variant> [pytest -k]
R=${GRAFT_REPO_ROOT:-/root/repo}; V=$1
O=$R/gpurun_out/r5; mkdir -p $O; cd $R
SE=$R/alphasnake-zero_amd/snake_engine
if [ -n "$2" ]; then python3 -m pytest tests/test_net_gpu.py tests/test_rect_conv_gpu.py -x -q -m gpu -k "$2" 2>&1 | tail -2; fi
for rep in 1 2; do
  for lib in "" "_$V"; do
    echo "== lib${lib:-_default} (run $rep)"
    SNK_LIB_PATH=$SE/libsnake_engine$lib.so SNK_CONV_ALGO=bf16 python3 tools/a16_layers.py 19 500 5 2>&1 | grep -v amdgpu.ids | tail -1
    SNK_LIB_PATH=$SE/libsnake_engine$lib.so SNK_CONV_ALGO=bf16 python3 tools/a16_layers.py 11 2300 5 2>&1 | grep -v amdgpu.ids | tail -1
    SNK_LIB_PATH=$SE/libsnake_engine$lib.so python3 bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --chunk 4096 --conv-algo bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   bench c4 bf16:', round(d['value']), 'of held clock', round(r['executed_frac_of_held_clock_peak'],3), 'MHz', round(r['clock_mhz']['median']))"
  done
done
SNK_LIB_PATH=$SE/libsnake_engine_dbg.so python3 tools/a16_stamps.py 1024 37 bf16 2>&1 | grep -v amdgpu.ids
