"""How busy is the GPU over bench.py's timed self-play steps?  (the small-batch regime.)

    python3 tools/gpu_busy.py OUT_PREFIX -- --games 256 --breadth 128 --steps 6 --warmup 2

starts `rocprofv3 --kernel-trace -- python3 bench.py <args> --trace-markers ...` as a child (this process never touches the GPU),
finds the timed region in the kernel trace by bench.py's two marker launches (k_clock_probe, csrc/probe.hip) and writes
OUT_PREFIX.json = bench.py's own JSON line plus

    "gpu_busy": {"frac": union of the kernels' [start, end) intervals / marker-to-marker time, "kernel_s": their sum,
                 "launches", "launches_per_step", "launches_per_rollout_tick", "gap_us": quantiles of the idle gaps between consecutive
                 kernels, "top": the ten kernels with the most time}

The trace runs under the profiler: its per-launch host cost (a few microseconds) is inside the measured wall, so `frac` is a
LOWER bound of the un-profiled run's (bench.py's value of the same run, printed beside it, shows what the profiler cost).
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def busy_from_trace(path, marker="k_clock_probe"):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 2:
        raise SystemExit(f"gpu_busy: {len(marks)} marker launches in {path}, need 2")
    a, b = marks[0], marks[1]
    t0, t1 = rows[a][1], rows[b][0]
    inner = rows[a + 1:b]
    busy, cur_s, cur_e, gaps = 0, None, None, []
    for s, e, _ in inner:
        if cur_e is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
    if cur_e is not None:
        busy += cur_e - cur_s
    per = defaultdict(lambda: [0, 0])
    for s, e, k in inner:
        short = k.split("(")[0].replace("void ", "")
        per[short][0] += e - s
        per[short][1] += 1
    top = sorted(per.items(), key=lambda kv: -kv[1][0])[:12]
    gaps.sort()

    def q(p):
        return gaps[min(len(gaps) - 1, int(p * len(gaps)))] / 1e3 if gaps else None
    return {"frac": busy / (t1 - t0), "wall_s": (t1 - t0) / 1e9, "busy_s": busy / 1e9, "kernel_s": sum(e - s for s, e, _ in inner) / 1e9,
            "launches": len(inner), "idle_s": (t1 - t0 - busy) / 1e9,
            "gap_us": {"p10": q(0.1), "p50": q(0.5), "p90": q(0.9), "p99": q(0.99), "count": len(gaps),
                       "sum_s_of_gaps_over_100us": sum(g for g in gaps if g > 100000) / 1e9},
            "top": [{"kernel": k, "s": v[0] / 1e9, "launches": v[1], "avg_us": v[0] / v[1] / 1e3} for k, v in top]}


def main():
    if "--" not in sys.argv or len(sys.argv) < 3:
        raise SystemExit(__doc__)
    cut = sys.argv.index("--")
    prefix, bench_args = os.path.abspath(sys.argv[1]), sys.argv[cut + 1:]        # (the child runs with cwd = /tmp)
    os.makedirs(os.path.dirname(prefix), exist_ok=True)
    out_dir = prefix + "_trace"
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out_dir, "--", sys.executable, os.path.join(REPO, "bench.py"),
           *bench_args, "--trace-markers", "--no-clock-probe", "--no-cpu-baseline", "--no-kernel-rooflines"]
    p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=open(prefix + ".err", "w"), text=True)
    if p.returncode:
        raise SystemExit(f"gpu_busy: bench under rocprofv3 exited {p.returncode}; see {prefix}.err")
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    traces = glob.glob(out_dir + "/**/*kernel_trace.csv", recursive=True)
    trace = max(traces, key=os.path.getsize)
    gb = busy_from_trace(trace)
    gb["launches_per_step"] = gb["launches"] / res["steps"]
    ticks = res.get("config", {}).get("rollout_ticks_per_step")
    if ticks:
        gb["launches_per_rollout_tick"] = gb["launches_per_step"] / ticks
        gb["ms_per_rollout_tick"] = res["ms_per_step"] / ticks
    gb["how"] = "rocprofv3 --kernel-trace of this very run; region = between bench.py's two marker launches; frac = union of kernel intervals / region"
    res["gpu_busy"] = gb
    stats = glob.glob(out_dir + "/**/*kernel_stats.csv", recursive=True)
    if stats:
        os.replace(max(stats, key=os.path.getsize), prefix + "_kernel_stats.csv")
    with open(prefix + ".json", "w") as f:
        f.write(json.dumps(res) + "\n")
    print(json.dumps({"value": res["value"], "ms_per_step": res["ms_per_step"], "gpu_busy": gb}))
    for t in traces:          # the raw traces are large: the summary is what is kept
        os.remove(t)


if __name__ == "__main__":
    main()
