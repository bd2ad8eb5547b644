#!/bin/bash
# Round 6, fourth GPU call: the whole GPU suite on the final library, then the round's profile set (kernel traces, HBM traffic and
# SQ counters of both towers on the shipped sources)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests_c.log 2>&1; rc=$?; tail -3 $O/gpu_tests_c.log
[ $rc -ne 0 ] && { tail -60 $O/gpu_tests_c.log; exit $rc; }
timeout -k 10 1000 bash tools/profile_set.sh r6 2>&1 | tail -25
