#!/bin/bash
# A/B of library variants on the 16-bit tower inside one GPU call: r5_ab.sh <variant name> [stamps variant name] [trace]
R=${GRAFT_REPO_ROOT:-/root/repo}; V=$1; S=$2
O=$R/gpurun_out/r5; mkdir -p $O; cd $R
SE=$R/alphasnake-zero_amd/snake_engine
for rep in 1 2; do
  for lib in "" "_$V"; do
    echo "== lib${lib:-_default} (run $rep)"
    SNK_LIB_PATH=$SE/libsnake_engine$lib.so SNK_CONV_ALGO=bf16 python3 tools/a16_layers.py 19 500 5 2>&1 | grep -v amdgpu.ids | tail -1
    SNK_LIB_PATH=$SE/libsnake_engine$lib.so SNK_CONV_ALGO=bf16 python3 tools/a16_layers.py 11 2300 5 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
if [ -n "$S" ] && [ "$S" != "none" ]; then
  SNK_LIB_PATH=$SE/libsnake_engine_$S.so python3 tools/a16_stamps.py 1024 37 bf16 2>&1 | grep -v amdgpu.ids | tee $O/a16_stamps_$S.log
fi
if [ "$3" = "trace" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python3 $R/bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --chunk 4096 --conv-algo bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $O/trace_c4.json 2> $O/trace_c4.err
  f=$(find $O/trace_c4 -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-200
fi
