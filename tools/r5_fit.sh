#!/bin/bash
# training-step A/B: the slab form of the weight gradient against the window form, same call; then the kernel trace of the fit
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
for v in "SNK_WGRAD=slabs" "SNK_WGRAD_NK=5" "SNK_WGRAD=slabs" "SNK_WGRAD_NK=5"; do echo "== $v"; env $v python3 tools/fit_time.py 10 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $O/fit_ab_$1.log
if [ "$2" = "trace" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fit_$1 -- python3 $R/tools/fit_time.py 8 > $O/trace_fit_$1.log 2>&1
  f=$(find $O/trace_fit_$1 -name "*kernel_stats.csv" | head -1); cp $f $O/fit_kernel_stats_$1.csv; head -30 $f | cut -c1-140
fi
