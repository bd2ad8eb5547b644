#!/bin/bash
# A/B inside ONE gpurun call, alternating (boxes differ by ~3 % in the clock they hold; run-to-run spread inside a call is ~0.1-0.5 %):
#     tools/ab.sh [-n reps] "ENV_A" "ENV_B" -- command ...
#   ENV_x: environment assignments for that side, e.g. "SNK_CONV_PERSIST=0" or "SNK_LIB_PATH=$SE/libsnake_engine_alt.so" (a
#   `make variant NAME=alt EXTRA=-D...` build) or "" for the defaults.  A bench.py command is summarised from its JSON line
#   (value, conv TFLOP/s, held clock, executed fraction of the held clock's peak); any other command by its last output line.
# Replaces round 5's eighteen one-shot tools/r5_*.sh (same pattern, different variant names); typical commands:
#     python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-rooflines                         (judged workload)
#     python3 bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --conv-algo bf16 --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-rooflines
#     python3 tools/a16_layers.py 19 500 5      python3 tools/rect_layers.py      python3 tools/fit_time.py 32
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
N=2; [ "$1" = "-n" ] && { N=$2; shift 2; }
A="$1"; B="$2"; shift 2; [ "$1" = "--" ] && shift
for rep in $(seq 1 $N); do
  for side in A B; do
    if [ $side = A ]; then E="$A"; else E="$B"; fi
    out=$(env $E timeout -k 10 900 "$@" 2>/dev/null) || { echo "$side ($E) run $rep FAILED"; exit 1; }
    echo "$out" | python3 -c "
import json, sys
lines = [l for l in sys.stdin.read().strip().splitlines() if l.strip()]
last = lines[-1] if lines else ''
try:
    d = json.loads(last); r = d['roofline']; c = r.get('clock_mhz') or {}
    print('$side [$E] run $rep: %.1f %s | conv %.1f TFLOP/s | %s MHz | executed/held %s | %.1f ms/step' % (d['value'], d['unit'], r['achieved'] or 0,
          round(c['median']) if c else '-', round(r.get('executed_frac_of_held_clock_peak', 0), 3) or '-', d['ms_per_step']))
except Exception:
    print('$side [$E] run $rep:', ' | '.join(l[:200] for l in lines[-2:]))
"
  done
done
