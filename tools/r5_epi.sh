#!/bin/bash
# A/B: the input-gradient epilogue requesting the next pass's rows row by row (default) against all rows before the exchange (-DHS_EPI_AHEAD=0)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_train_ops_gpu.py -x -q > $O/epi_tests.log 2>&1; rc=$?; tail -2 $O/epi_tests.log
[ $rc -ne 0 ] && { tail -40 $O/epi_tests.log; exit $rc; }
for i in 1 2; do
  for v in "" "ea0"; do
    echo "== ${v:-default (ahead)}"; SNK_LIB_PATH=$R/alphasnake-zero_amd/snake_engine/libsnake_engine${v:+_$v}.so python3 tools/fit_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1
  done
done | tee $O/epi_ab.log
