#!/bin/bash
# the final library: kernel trace of the fit, the training fuzzers
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
for s in 3 4; do timeout -k 10 250 python3 tools/fuzz_train.py $s 24 2>&1 | grep -v amdgpu | tail -1; done | tee $O/r5b_fuzz_train2.log
timeout -k 10 250 python3 tools/fuzz_wgrad.py 7 60 2>&1 | grep -v amdgpu | tail -1 | tee $O/r5b_fuzz_wgrad2.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fit -- python3 $R/tools/fit_time.py 8 > $O/trace_fit.log 2>&1 || exit 1
f=$(find $O/trace_fit -name "*kernel_stats.csv" | head -1); cp $f $O/r5b_fit_kernel_stats.csv; rm -rf $O/trace_fit
python3 - $O/r5b_fit_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'][:58]:58s} {int(r['Calls']):5d} x {float(r['AverageNs']) / 1e3:7.1f} us = {float(r['TotalDurationNs']) / 1e6:7.1f} ms")
print("all kernels", sum(float(r['TotalDurationNs']) for r in rows) / 1e6, "ms")
PY
