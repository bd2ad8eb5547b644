#!/bin/bash
# the round's bench lines (gpurun): driver-style configs[1], configs[0] on the GPU, 2 ranks over gloo on one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4_bench.json 2> $O/r4_bench.err && tail -2 $O/r4_bench.err
python bench.py --games 8 --breadth 25 --steps 40 --warmup 5 --no-conv-timing --no-cpu-baseline --no-kernel-rooflines > $O/r4_bench_config0.json 2> $O/r4_bench_config0.err
python bench.py --gpus 2 --steps 3 --warmup 1 > $O/r4_bench_2ranks_gloo_1gpu.json 2> $O/r4_bench_2ranks.err
python - <<'P'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r4")
for f in ("r4_bench.json", "r4_bench_config0.json", "r4_bench_2ranks_gloo_1gpu.json"):
    d = json.load(open(os.path.join(O, f))); r = d["roofline"]
    print(f, round(d["value"], 1), r.get("achieved"), r.get("frac"), r.get("traffic"), r.get("traffic_source"), (r.get("clock_mhz") or {}).get("median"),
          [(x["rank"], round(x["env_steps_per_s"], 1), round(x["host_cpu_s"], 1), x["host_threads"]) for x in d["ranks"]], d.get("cpu_baseline", {}).get("value"))
P
