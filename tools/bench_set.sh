#!/bin/bash
# A round's bench lines (run through gpurun; copy what is to be judged into profiles/ afterwards):   bench_set.sh <tag, e.g. r6>
#   driver-style configs[1]; configs[0] on the GPU; 2 ranks over gloo on one GPU; 8 ranks over gloo on one GPU (the launcher's N = 8
#   path, tiny games); the configs[4] shape over a LONG window (its cost per root turn grows as snakes die: DESIGN section 7);
#   the reference's own settings (train.py:9-11) with the GPU-busy fraction from a kernel trace (tools/gpu_busy.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r6}
O=$R/gpurun_out/${T}set
mkdir -p $O; cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/${T}_bench.err && tail -2 $O/${T}_bench.err
python bench.py --games 8 --breadth 25 --steps 40 --warmup 5 --no-conv-timing --no-cpu-baseline --no-kernel-rooflines > $O/${T}_bench_config0.json 2> $O/${T}_bench_config0.err
python bench.py --gpus 2 --steps 3 --warmup 1 > $O/${T}_bench_2ranks_gloo_1gpu.json 2> $O/${T}_bench_2ranks.err
python bench.py --gpus 6 --games 64 --breadth 8 --steps 2 --warmup 1 > $O/${T}_bench_6ranks_gloo_1gpu.json 2> $O/${T}_bench_6ranks.err
python bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --conv-algo bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-rooflines > $O/${T}_bench_config4_shape_bf16_long.json 2> $O/${T}_bench_config4_shape_bf16_long.err
python3 tools/gpu_busy.py $O/${T}_busy_ref_defaults -- --games 256 --breadth 128 --steps 6 --warmup 2 > /dev/null
python - <<P
import json, os
O = "$O"
for f in sorted(os.listdir(O)):
    if not f.endswith(".json") or "_trace" in f: continue
    d = json.loads(open(os.path.join(O, f)).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, round(d["value"], 1), r.get("achieved"), r.get("frac"), r.get("executed_frac_of_held_clock_peak"), r.get("traffic_source"), r.get("counters_source"),
          (r.get("clock_mhz") or {}).get("median"), (d.get("gpu_busy") or {}).get("frac"),
          [(x["rank"], round(x["env_steps_per_s"], 1)) for x in d["ranks"]], d.get("cpu_baseline", {}).get("value"))
P
