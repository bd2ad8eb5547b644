#!/bin/bash
# kernel traces of the fit with and without the deferred batch norm (one call)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export SNK_TRAIN_DEFER_BN=$v
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_defer$v -- python3 $R/tools/fit_time.py 8 > $O/trace_defer$v.log 2>&1 || exit 1
  f=$(find $O/trace_defer$v -name "*kernel_stats.csv" | head -1); cp $f $O/fit_kernel_stats_defer$v.csv
  rm -rf $O/trace_defer$v
  echo "== defer $v"; head -14 $O/fit_kernel_stats_defer$v.csv | cut -d, -f1-4 | cut -c1-150
done
