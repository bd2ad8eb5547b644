#!/bin/bash
# full GPU test suite + the default bench line + chunk-size A/B of the bf16 configs[4] shape
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-rooflines > $O/bench_default_short.json 2> $O/bench_default_short.err
python3 -c "import json; d=json.loads(open('$O/bench_default_short.json').read().strip().splitlines()[-1]); r=d['roofline']; print('bench default:', d['value'], 'frac', r['frac'], 'MHz', r['clock_mhz']['median'])"
for c in 1024 2048 8192; do
python3 bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --chunk $c --conv-algo bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $O/bench_c4_bf16_chunk$c.json 2> $O/bench_c4_bf16_chunk$c.err
python3 -c "import json; d=json.loads(open('$O/bench_c4_bf16_chunk$c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('bench c4 bf16 chunk $c:', d['value'], 'frac', r['frac'], 'of held clock', r['executed_frac_of_held_clock_peak'], 'MHz', r['clock_mhz']['median'])"
done
python3 bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --chunk 4096 --conv-algo f16a --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $O/bench_c4_f16a.json 2> $O/bench_c4_f16a.err
python3 -c "import json; d=json.loads(open('$O/bench_c4_f16a.json').read().strip().splitlines()[-1]); r=d['roofline']; print('bench c4 f16a:', d['value'], 'frac', r['frac'], 'of held clock', r['executed_frac_of_held_clock_peak'], 'MHz', r['clock_mhz']['median'])"
