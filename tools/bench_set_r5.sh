#!/bin/bash
# the round's bench lines (gpurun): driver-style configs[1], configs[0] on the GPU, 2 ranks over gloo on one GPU, the configs[4] shape
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5set
mkdir -p $O; cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5_bench.json 2> $O/r5_bench.err && tail -2 $O/r5_bench.err
python bench.py --games 8 --breadth 25 --steps 40 --warmup 5 --no-conv-timing --no-cpu-baseline --no-kernel-rooflines > $O/r5_bench_config0.json 2> $O/r5_bench_config0.err
python bench.py --gpus 2 --steps 3 --warmup 1 > $O/r5_bench_2ranks_gloo_1gpu.json 2> $O/r5_bench_2ranks.err
for a in bf16 f16a; do
python bench.py --board 19 --snakes 8 --blocks 10 --games 4096 --chunk 4096 --conv-algo $a --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $O/r5_bench_config4_shape_$a.json 2> $O/r5_bench_config4_shape_$a.err
done
python - <<'P'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r5set")
for f in ("r5_bench.json", "r5_bench_config0.json", "r5_bench_2ranks_gloo_1gpu.json", "r5_bench_config4_shape_bf16.json", "r5_bench_config4_shape_f16a.json"):
    d = json.load(open(os.path.join(O, f))); r = d["roofline"]
    print(f, round(d["value"], 1), r.get("achieved"), r.get("frac"), r.get("executed_frac_of_held_clock_peak"), r.get("traffic"), r.get("traffic_source"), (r.get("clock_mhz") or {}).get("median"),
          [(x["rank"], round(x["env_steps_per_s"], 1), round(x["host_cpu_s"], 1), x["host_threads"]) for x in d["ranks"]], d.get("cpu_baseline", {}).get("value"))
P
