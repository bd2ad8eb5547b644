#!/bin/bash
# A/B: library variants built with -fno-slp-vectorize (no packed-f32 VALU) against the default build, alternating in one call: r5_noslp.sh <variant> ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
for i in 1 2; do
  for v in "" "$@"; do
    echo "== ${v:-default}"; SNK_LIB_PATH=$R/alphasnake-zero_amd/snake_engine/libsnake_engine${v:+_$v}.so python3 tools/fit_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1
  done
done | tee $O/noslp_ab_fit.log
