#!/bin/bash
# Round 6, seventh GPU call: how many tower layers of the 37 x 37 canvas should take the sub-rectangle form (default 12 of 20)?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
C4="python3 bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --conv-algo bf16 --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-rooflines"
{
echo "== configs[4] shape bf16: default (12 sub-rectangle layers) vs 16"; bash tools/ab.sh "" "SNK_CONV_RECT_LAYERS=16" -- $C4
echo "== default vs 19"; bash tools/ab.sh "" "SNK_CONV_RECT_LAYERS=19" -- $C4
echo "== per layer"; SNK_CONV_ALGO=bf16 python3 tools/a16_layers.py 19 500 5 2>&1 | grep -v amdgpu | tail -30
} 2>&1 | tee $O/rect_layers_ab.log
