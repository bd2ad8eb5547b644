#!/bin/bash
# round-5 second half: the whole GPU suite on the final library, then the training logs (fit time, kernel trace, three soak generations)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; rc=$?; tail -3 $O/gputests.log
[ $rc -ne 0 ] && { tail -60 $O/gputests.log; exit $rc; }
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
for i in 1 2; do python3 tools/fit_time.py 32 2>&1 | grep -v amdgpu | tail -1; done > $O/r5b_fit_time.log; cat $O/r5b_fit_time.log
echo "== every switch of the round's second half off"; SNK_TRAIN_DEFER_BN=0 SNK_TRAIN_RES_MASK=0 SNK_TRAIN_BATCH_PREP=0 SNK_TRAIN_HEAD_FUSED=0 python3 tools/fit_time.py 32 2>&1 | grep -v amdgpu | tail -1 | tee $O/r5b_fit_time_off.log
python3 tools/soak_train.py 3 2>&1 | grep -v amdgpu | tail -14 > $O/r5b_soak_train.log; tail -6 $O/r5b_soak_train.log
python3 tools/fuzz_wgrad.py 5 60 2>&1 | grep -v amdgpu | tail -1 > $O/r5b_fuzz_wgrad.log; cat $O/r5b_fuzz_wgrad.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fit -- python3 $R/tools/fit_time.py 8 > $O/trace_fit.log 2>&1 || exit 1
f=$(find $O/trace_fit -name "*kernel_stats.csv" | head -1); cp $f $O/r5b_fit_kernel_stats.csv; rm -rf $O/trace_fit
python3 - $O/r5b_fit_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:58]:58s} {int(r['Calls']):5d} x {float(r['AverageNs']) / 1e3:7.1f} us = {float(r['TotalDurationNs']) / 1e6:7.1f} ms")
print("all kernels", sum(float(r['TotalDurationNs']) for r in rows) / 1e6, "ms")
PY
