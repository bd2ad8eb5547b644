#!/bin/bash
# Round 6, third GPU call: resident-grid launches of the 16-bit towers (snk_conv_set_persist / SNK_CONV_PERSIST): parity, then A/B
#   noloop = a -DHS_NO_ITEM_LOOP build (round 5's one-item kernels); default library: looped kernels, persist 0 / 2 / 3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
SE=$R/alphasnake-zero_amd/snake_engine
timeout -k 10 600 python -m pytest tests/test_rect_conv_gpu.py tests/test_net_gpu.py -x -q > $O/gpu_tests_b.log 2>&1; rc=$?; tail -3 $O/gpu_tests_b.log
[ $rc -ne 0 ] && { tail -60 $O/gpu_tests_b.log; exit $rc; }
C4="python3 bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --conv-algo bf16 --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-rooflines"
{
echo "== 16-bit tower per layer (a16_layers 19 500 5): no loop vs loop, one workgroup per item"; bash tools/ab.sh "SNK_LIB_PATH=$SE/libsnake_engine_noloop.so" "SNK_CONV_PERSIST=0" -- python3 tools/a16_layers.py 19 500 5
echo "== the same: one workgroup per item vs 2 resident per CU"; bash tools/ab.sh "SNK_CONV_PERSIST=0" "SNK_CONV_PERSIST=2" -- python3 tools/a16_layers.py 19 500 5
echo "== bench configs[4] shape bf16, 25 steps: no loop vs 2 resident per CU"; bash tools/ab.sh "SNK_LIB_PATH=$SE/libsnake_engine_noloop.so" "SNK_CONV_PERSIST=2" -- $C4
echo "== bench configs[4] shape bf16: loop with one workgroup per item vs 3 resident per CU"; bash tools/ab.sh -n 1 "SNK_CONV_PERSIST=0" "SNK_CONV_PERSIST=3" -- $C4
echo "== judged workload, 6 steps: no loop vs default library (its float32-accurate kernels are not looped: must be equal)"; bash tools/ab.sh "SNK_LIB_PATH=$SE/libsnake_engine_noloop.so" "" -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-rooflines
} 2>&1 | tee $O/persist_ab.log
