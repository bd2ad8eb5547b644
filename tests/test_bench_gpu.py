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
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and r["kernel"] == "k_conv3x3_f16s" and r["achieved"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    for k in ("step", "clone", "observe"):
        e = d["engine_kernels"][k]
        assert e["bound"] == "hbm" and e["unit"] == "GB/s" and 0 < e["frac"] < 1
