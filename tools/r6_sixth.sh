#!/bin/bash
# Round 6, sixth GPU call: k_observe with the rings of 16-bit-cell boards left in global memory (live segments to LDS): parity, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
SE=$R/alphasnake-zero_amd/snake_engine
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_mcts_gpu.py tests/test_search_fullsize_gpu.py tests/test_rect_conv_gpu.py -x -q > $O/gpu_tests_d.log 2>&1; rc=$?; tail -3 $O/gpu_tests_d.log
[ $rc -ne 0 ] && { tail -60 $O/gpu_tests_d.log; exit $rc; }
timeout -k 10 300 python3 tools/fuzz_engine.py 9500 60 2>&1 | grep -v amdgpu | tail -2
{
for b in 19 11; do
  echo "== observe_time, board $b, 32768 games: base (whole record in LDS) vs new"
  bash tools/ab.sh "OBS_BOARD=$b OBS_LIB=$SE/libsnake_engine_base.so" "OBS_BOARD=$b" -- python3 tools/observe_time.py 32768 planes
  bash tools/ab.sh -n 1 "OBS_BOARD=$b OBS_LIB=$SE/libsnake_engine_base.so" "OBS_BOARD=$b" -- python3 tools/observe_time.py 32768 mask+key
done
echo "== bench configs[4] shape bf16, 25 steps: base vs new"
bash tools/ab.sh "SNK_LIB_PATH=$SE/libsnake_engine_base.so" "" -- python3 bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --conv-algo bf16 --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-rooflines
} 2>&1 | tee $O/observe_ab.log
