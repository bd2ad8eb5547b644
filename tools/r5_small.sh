#!/bin/bash
# the stem's sums in its epilogue + the head's 1x1 stage in the last batch-norm kernel (round 5): tests, A/B in one call
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_train_ops_gpu.py tests/test_next_rows_gpu.py -x -q > $O/small_tests_$1.log 2>&1; rc=$?; tail -3 $O/small_tests_$1.log
[ $rc -ne 0 ] && { tail -40 $O/small_tests_$1.log; exit $rc; }
for v in "SNK_TRAIN_HEAD_FUSED=0" "SNK_TRAIN_HEAD_FUSED=1" "SNK_TRAIN_HEAD_FUSED=0" "SNK_TRAIN_HEAD_FUSED=1"; do echo "== $v"; env $v timeout -k 10 200 python3 tools/fit_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $O/small_ab_$1.log
