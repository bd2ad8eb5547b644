#!/bin/bash
# branch-free input-gradient epilogue + the shortcut's gradient through mask bytes (round 5): tests, A/B, kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_train_ops_gpu.py tests/test_next_rows_gpu.py -x -q > $O/resmask_tests_$1.log 2>&1; rc=$?; tail -3 $O/resmask_tests_$1.log
[ $rc -ne 0 ] && { tail -40 $O/resmask_tests_$1.log; exit $rc; }
for v in 0 1 0 1; do echo "== SNK_TRAIN_RES_MASK=$v"; SNK_TRAIN_RES_MASK=$v timeout -k 10 200 python3 tools/fit_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $O/resmask_ab_$1.log
echo "== SNK_TRAIN_DEFER_BN=0 SNK_TRAIN_RES_MASK=0 SNK_TRAIN_BATCH_PREP=0"; SNK_TRAIN_DEFER_BN=0 SNK_TRAIN_RES_MASK=0 SNK_TRAIN_BATCH_PREP=0 timeout -k 10 200 python3 tools/fit_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/resmask_ab_$1.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$1 -- python3 $R/tools/fit_time.py 8 > $O/trace_$1.log 2>&1 || exit 1
f=$(find $O/trace_$1 -name "*kernel_stats.csv" | head -1); cp $f $O/fit_kernel_stats_$1.csv; rm -rf $O/trace_$1
python3 - $O/fit_kernel_stats_$1.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:58]:58s} {int(r['Calls']):5d} x {float(r['AverageNs']) / 1e3:7.1f} us = {float(r['TotalDurationNs']) / 1e6:7.1f} ms")
print("all kernels", sum(float(r['TotalDurationNs']) for r in rows) / 1e6, "ms")
PY
