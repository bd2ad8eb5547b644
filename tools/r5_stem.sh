#!/bin/bash
# the deferred stem (round 5): tests, then the fit with and without it alternating in one call
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_train_ops_gpu.py tests/test_next_rows_gpu.py -x -q > $O/stem_tests_$1.log 2>&1; rc=$?; tail -3 $O/stem_tests_$1.log
[ $rc -ne 0 ] && { tail -50 $O/stem_tests_$1.log; exit $rc; }
for v in 0 1 0 1; do echo "== SNK_TRAIN_DEFER_STEM=$v"; SNK_TRAIN_DEFER_STEM=$v timeout -k 10 200 python3 tools/fit_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $O/stem_ab_$1.log
