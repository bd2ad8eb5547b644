#!/bin/bash
# A/B: the weight-gradient kernel with the second wavefront of every SIMD delayed by s_sleep k after each window's barrier
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
for i in 1 2; do
  for v in "" "$@"; do
    echo "== ${v:-default}"; SNK_LIB_PATH=$R/alphasnake-zero_amd/snake_engine/libsnake_engine${v:+_$v}.so python3 tools/wgrad_time.py 2>&1 | grep -v amdgpu.ids | tail -2
  done
done | tee $O/wgskew_ab.log
