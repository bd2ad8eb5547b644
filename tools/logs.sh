#!/bin/bash
# A round's soak / fuzz / timing logs (run through gpurun; copied into profiles/ afterwards):   logs.sh <tag, e.g. r6>
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r6}; O=$R/gpurun_out/${T}logs; mkdir -p $O; cd $R
python3 tools/fit_time.py 32 2>&1 | grep -v amdgpu | tail -1 > $O/${T}_fit_time.log; cat $O/${T}_fit_time.log
python3 tools/soak_train.py 3 2>&1 | grep -v amdgpu | tail -14 > $O/${T}_soak_train.log; tail -5 $O/${T}_soak_train.log
python3 tools/fuzz_rect.py 1500 150 2>&1 | grep -v amdgpu | tail -2 > $O/${T}_fuzz_rect.log
SNK_CONV_ALGO=bf16 python3 tools/fuzz_rect.py 1700 150 2>&1 | grep -v amdgpu | tail -2 >> $O/${T}_fuzz_rect.log
SNK_CONV_ALGO=f16a python3 tools/fuzz_rect.py 1900 100 2>&1 | grep -v amdgpu | tail -2 >> $O/${T}_fuzz_rect.log
cat $O/${T}_fuzz_rect.log
python3 tools/fuzz_engine.py 9000 120 2>&1 | grep -v amdgpu | tail -2 > $O/${T}_fuzz_engine.log; cat $O/${T}_fuzz_engine.log
python3 tools/fuzz_mcts.py 9000 300 2>&1 | grep -v amdgpu | tail -3 > $O/${T}_fuzz_mcts.log; cat $O/${T}_fuzz_mcts.log
python3 tools/fuzz_pit.py 9000 60 2>&1 | grep -v amdgpu | tail -2 > $O/${T}_fuzz_pit.log; cat $O/${T}_fuzz_pit.log
python3 tools/soak_selfplay.py 1024 50 9 2>&1 | grep -v amdgpu | tail -4 > $O/${T}_soak_selfplay.log; cat $O/${T}_soak_selfplay.log
