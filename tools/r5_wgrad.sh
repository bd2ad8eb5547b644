#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
for v in "SNK_WGRAD=slabs" "SNK_WGRAD_NK=5" "SNK_WGRAD_NK=4"; do echo "== $v"; env $v python3 tools/wgrad_time.py 2>&1 | grep -v amdgpu.ids; done | tee $O/wgrad_ab_$1.log
if [ "$2" = "tests" ]; then python3 -m pytest tests/test_train_ops_gpu.py -x -q -m gpu 2>&1 | tail -3; fi
if [ "$3" = "fit" ]; then python3 tools/fit_time.py 8 2>&1 | grep -v amdgpu.ids | tail -8; fi
