#!/bin/bash
# Round 6, second GPU call: the whole GPU suite on the rebuilt library, the configs[4] shape over a long timed window, the SQ
# counters of the judged f16s tower on the shipped sources
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests_a.log 2>&1; rc=$?; tail -3 $O/gpu_tests_a.log
[ $rc -ne 0 ] && { tail -60 $O/gpu_tests_a.log; exit $rc; }
timeout -k 10 300 python3 bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --chunk 4096 --conv-algo bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-rooflines > $O/bench_config4_long.json 2> $O/bench_config4_long.err; echo "config4 rc=$?"; cut -c1-300 $O/bench_config4_long.json
bash tools/pmc_tower.sh r6_f16s f16s 2300 11 | tail -3
