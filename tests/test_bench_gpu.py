"""GPU test of the bench.py contract: one JSON line with the metric, the roofline object and the cpu_baseline object."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_the_contract_fields():
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--games", "16", "--breadth", "16", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"].startswith("f32") and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 2 * 16 / (d["ms_per_step"] * 2e-3)) / d["value"] < 0.05
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and r["kernel"].startswith("k_conv3x3_f16s") and r["achieved"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    sr = r["sub_rectangles"]                                        # the first six tower layers convolve the grown board window only
    assert sr["layers"] == 6 and sr["of"] == 8 and len(sr["tiles_executed_vs_full_per_layer"]) == 6
    assert all(0.3 < v <= 1.0 for v in sr["tiles_executed_vs_full_per_layer"]) and 0.5 < sr["tiles_executed_vs_full"] < 1.0
    assert sr["tiles_executed_vs_full_per_layer"] == sorted(sr["tiles_executed_vs_full_per_layer"])
    assert abs(r["executed_frac"] - r["frac"] * 3.0 * 448 / 441 * sr["tiles_executed_vs_full"]) < 1e-6
    # derived fractions stay fractions: algorithmic <= executed (three MFMAs per product) <= the peak, also at the clock held
    assert 0 < r["frac"] <= r["executed_frac"] <= 1.0 and r["executed_frac"] <= r["executed_frac_of_held_clock_peak"] <= 1.0
    clk = r["clock_mhz"]                                            # the clock the chip held in an untimed extra turn of the same loop
    assert clk["samples"] >= 1 and 500.0 < clk["p10"] <= clk["median"] <= clk["p90"] < 2700.0 and "UNTIMED" in clk["how"]
    # traffic is quoted only from a profile measured on the sources the loaded library was built from (else null + the reason)
    assert (r["traffic"] is None) != str(r["traffic_source"]).endswith(".json"), (r["traffic"], r["traffic_source"])
    assert len(d["ranks"]) == 1 and d["ranks"][0]["env_steps"] == 32 and d["exchange"]["dist_backend"] is None
    rk = d["ranks"][0]
    assert rk["host_threads"] == d["config"]["host_threads_per_rank"] >= 1 and 0 < rk["host_cpu_s"] and 0 < rk["conv_gpu_s"] < rk["self_play_s"]
    assert d["config"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert (r["counters"] is None) != str(r["counters_source"]).endswith(".json"), (r["counters"], r["counters_source"])
    for k in ("step", "clone", "observe", "observe_19x19_8_snakes"):
        e = d["engine_kernels"][k]
        assert e["bound"] == "hbm" and e["unit"] == "GB/s" and 0 < e["frac"] < 1


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns two fresh ranks before touching the GPU and
    relays rank 0's line (RCCL when the box has >= 2 GPUs, otherwise both ranks share GPU 0 and talk over gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--games", "16", "--breadth", "16",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert d["config"]["games_per_gpu"] == 16 and d["config"]["parallelism"] == "games sharded x2"
    # the row count of the exchange follows trainer.py:63-72 on the records of BOTH ranks: (1 warm-up + 2 timed) turns x 16 games
    # x 4 snakes each, fewer than one batch of 2 048 -> all 384 rows form the batch, 192 from each rank
    assert d["config"]["sample_rows_gathered"] == 384 and d["exchange"]["rows_per_rank"] == 192
    assert d["exchange"]["dist_backend"] in ("nccl", "gloo") and d["exchange"]["bytes_per_rank"] == 192 * (21 * 21 * 3 + 3) * 4
    assert d["exchange"]["all_gather_s_max"] > 0 and d["exchange"]["all_reduce_s_max"] > 0
    # one row per rank: what a scaling curve is read with
    assert [r["rank"] for r in d["ranks"]] == [0, 1]
    for r in d["ranks"]:
        assert r["env_steps"] == 32 and r["records"] == 192 and r["net_evals"] > 0
        assert 0 < r["self_play_s"] <= r["wall_s"] <= d["ms_per_step"] * 2e-3 * 1.001
        assert r["all_gather_s"] > 0 and r["all_reduce_s"] > 0 and r["env_steps_per_s"] > 0
    # whole-job value: both ranks' root env-steps (2 turns x 16 games each, nobody dies in two turns) over the slowest rank's time
    assert abs(d["value"] - 2 * 2 * 16 / (d["ms_per_step"] * 2e-3)) / d["value"] < 0.05
    assert abs(d["value"] - sum(r["env_steps"] for r in d["ranks"]) / max(r["wall_s"] for r in d["ranks"])) / d["value"] < 0.05
    assert "cpu_baseline" not in d                                 # rank 0 at N = 1 only
    assert d["config"]["dist_backend"] in ("nccl", "gloo")
    # the ranks share the host: each runs usable cpus // 2 torch threads, and the IPC mode this pool's driver needs is set
    assert all(r["host_threads"] == d["config"]["host_threads_per_rank"] >= 1 and r["host_cpu_s"] > 0 for r in d["ranks"])
    assert d["config"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_five_ranks_share_one_gpu_over_gloo():
    """the launcher's N-rank path beyond two ranks, rehearsed on the one GPU there is.  The pool's process
    guard allows six processes on a card (this pytest process is one of them), so N = 5 here; the 8-rank exchange and fit run over
    gloo on the CPU (tests/test_dist_cpu.py::test_eight_rank_rehearsal_over_gloo).  Checked: one row per rank, every rank's env-steps,
    the per-rank thread cap (usable cpus // ranks), value = sum of env-steps / the slowest rank's wall, equal shares of the exchange."""
    import torch
    if torch.cuda.device_count() >= 5:
        pytest.skip("five GPUs present: this rehearsal is for boxes where ranks must share a card")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "5", "--games", "16", "--breadth", "8",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 5 and d["config"]["parallelism"] == "games sharded x5" and d["config"]["dist_backend"] == "gloo"
    assert [r["rank"] for r in d["ranks"]] == [0, 1, 2, 3, 4]
    threads = d["config"]["host_threads_per_rank"]
    assert threads >= 1 and all(r["host_threads"] == threads for r in d["ranks"])
    for r in d["ranks"]:            # 2 timed turns x 16 games; (1 + 2) turns x 16 games x <= 4 snakes recorded (a snake may die early)
        assert 28 <= r["env_steps"] <= 32 and 160 <= r["records"] <= 192 and r["net_evals"] > 0 and 0 < r["self_play_s"] <= r["wall_s"]
    assert abs(d["value"] - sum(r["env_steps"] for r in d["ranks"]) / max(r["wall_s"] for r in d["ranks"])) / d["value"] < 0.05
    # fewer records than one batch of 2 048: every record is sampled (trainer.py:63-72), a whole number of rows per rank
    total = sum(r["records"] for r in d["ranks"])
    assert d["config"]["sample_rows_gathered"] == total // 5 * 5 and total // 5 <= d["exchange"]["rows_per_rank"] <= 192
    assert "cpu_baseline" not in d


def test_one_rank_under_the_launcher_is_the_plain_line():
    """N = 1 started the way the driver starts N > 1 (python -m torch.distributed.run) describes the same workload as the plain run"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    args = ["--gpus", "1", "--games", "16", "--breadth", "16", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-rooflines",
            "--no-clock-probe"]
    a = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=600)
    assert a.returncode == 0, a.stderr[-3000:]
    b = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    da, db = (json.loads([l for l in o.stdout.splitlines() if l.strip().startswith("{")][-1]) for o in (a, b))
    assert da["config"]["workload"] == db["config"]["workload"] and da["n_gpus"] == db["n_gpus"] == 1
    assert da["config"]["parallelism"] == db["config"]["parallelism"] and da["ranks"][0]["env_steps"] == db["ranks"][0]["env_steps"] == 16
    assert da["metric"] == db["metric"] and da["scaling"] == db["scaling"] and da["dtype"] == db["dtype"]


def test_bench_games_total_splits_the_job_over_the_ranks():
    """--games-total T (BASELINE configs[3] is `--gpus 8 --games-total 262144 --breadth 200`): T / N games per GPU, strong scaling"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--games-total", "24", "--breadth", "8",
                          "--steps", "1", "--warmup", "1", "--no-conv-timing"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    assert d["scaling"] == "strong" and d["config"]["games_per_gpu"] == 12 and d["config"]["games_total"] == 24
    assert sum(r["env_steps"] for r in d["ranks"]) == 24
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--games-total", "25"], capture_output=True,
                         text=True, env=env, timeout=120)
    assert out.returncode != 0 and "does not split evenly" in out.stderr


def test_bench_refuses_a_world_size_mismatch_before_touching_the_gpu():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--games", "16"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_rccl_calls_of_the_exchange_run_on_one_rank():
    """the collectives of the iteration-end exchange, of bench.py's timing reduction and of the trainer's gradient bucket,
    issued to RCCL itself through a one-rank "nccl" process group (what a one-GPU box allows)"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "helpers", "rccl_single_rank.py")], capture_output=True,
                         text=True, env=env, timeout=600)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stderr[-3000:]
