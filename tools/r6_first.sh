#!/bin/bash
# Round 6, first GPU call: (1) the store-hazard micro-experiment, (2) how busy the GPU is at the reference's own settings
# (train.py:9-11: 256 games, breadth 128) and at configs[0], (3) a >= 30 s timed line of the configs[4] shape (bf16 tower)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
timeout -k 10 120 tools/micro/store_hazard 3 > $O/store_hazard.json 2> $O/store_hazard.err; echo "store_hazard rc=$?"
timeout -k 10 240 python3 tools/gpu_busy.py $O/busy_ref_defaults -- --games 256 --breadth 128 --steps 6 --warmup 2 | cut -c1-1500 && \
timeout -k 10 240 python3 tools/gpu_busy.py $O/busy_config0 -- --games 8 --breadth 25 --steps 30 --warmup 3 | cut -c1-1500 && \
timeout -k 10 400 python3 bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --chunk 4096 --conv-algo bf16 --steps 150 --warmup 10 --no-cpu-baseline --no-kernel-rooflines > $O/bench_config4_long.json 2> $O/bench_config4_long.err; echo "config4 rc=$?"; cut -c1-400 $O/bench_config4_long.json
