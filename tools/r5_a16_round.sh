#!/bin/bash
# one A/B round of the 16-bit tower on the GPU box: r5_a16_round.sh <tag> [pytest -k expression | none]
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; K=${2:-none}
O=$R/gpurun_out/r5; mkdir -p $O; cd $R
if [ "$K" != "none" ]; then python3 -m pytest tests/test_net_gpu.py tests/test_rect_conv_gpu.py -x -q -m gpu -k "$K" > $O/tests_$T.log 2>&1; tail -4 $O/tests_$T.log; fi
SNK_CONV_ALGO=bf16 python3 tools/a16_layers.py 19 500 5 > $O/a16_layers_$T.log 2>&1
D=$R/alphasnake-zero_amd/snake_engine/libsnake_engine_dbg.so
SNK_LIB_PATH=$D python3 tools/a16_stamps.py 1024 37 bf16 > $O/a16_stamps_$T.log 2>&1
SNK_LIB_PATH=$D python3 tools/a16_stamps.py 4096 21 bf16 >> $O/a16_stamps_$T.log 2>&1
python3 bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --chunk 4096 --conv-algo bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $O/bench_c4_bf16_$T.json 2> $O/bench_c4_bf16_$T.err
grep -v amdgpu.ids $O/a16_layers_$T.log | tail -4; grep -v amdgpu.ids $O/a16_stamps_$T.log
python3 -c "import json; d=json.loads(open('$O/bench_c4_bf16_$T.json').read().strip().splitlines()[-1]); r=d['roofline']; print('bench c4 bf16:', d['value'], 'frac', r['frac'], 'executed', r['executed_frac'], 'of held clock', r['executed_frac_of_held_clock_peak'], 'MHz', r['clock_mhz']['median'])"
if [ -f $R/alphasnake-zero_amd/snake_engine/libsnake_engine_dbg1.so ]; then
  echo "--- one block per CU (stamps, -DHS_ONE_PER_CU)"
  SNK_LIB_PATH=$R/alphasnake-zero_amd/snake_engine/libsnake_engine_dbg1.so python3 tools/a16_stamps.py 1024 37 bf16 2>&1 | grep -v amdgpu.ids | tee $O/a16_stamps_one_per_cu_$T.log
fi
if [ "$3" = "pmc" ]; then
  cd /tmp && export TMPDIR=/tmp
  SNK_CONV_ALGO=bf16 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_lds_$T -- python3 $R/tools/tower_only.py 500 2 19 > $O/pmc_lds_$T.log 2>&1
  for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do python3 $R/tools/pmc_summary.py $O/pmc_lds_$T $c "k_conv3x3_f16s<"; python3 $R/tools/pmc_summary.py $O/pmc_lds_$T $c "k_conv3x3_f16s_rect<2"; done
fi
