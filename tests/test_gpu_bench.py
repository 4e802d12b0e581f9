"""bench.py contract on the GPU box: one JSON line (last on stdout) with the required keys, the roofline and
cpu_baseline objects; and the RCCL (nccl) batch-sharded path exercised at world_size 1."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(extra_env, *args):
    env = dict(os.environ, **extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    last = [ln for ln in p.stdout.strip().splitlines() if ln.strip()][-1]
    return json.loads(last)


def test_bench_json_contract():
    d = _run({}, "--steps", "210", "--warmup", "21", "--cpu-seconds", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 210 and d["warmup"] == 21 and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and 0 < r["frac"] < 1.5 and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "updates" in c["sample"]
    assert d["value"] > 100 * c["value"]              # sanity: GPU path is the thing measured
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-6


def test_bench_rccl_path_world1():
    d = _run({"GSMVI_BENCH_FORCE_DIST": "1"}, "--steps", "42", "--warmup", "21", "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and "RCCL" in d["config"]["parallelism"] and d["value"] > 0


def test_bench_row_block_path_world1():
    d = _run({"GSMVI_BENCH_FORCE_DIST": "1"}, "--steps", "42", "--warmup", "21", "--no-cpu-baseline", "--shard", "rows")
    assert d["n_gpus"] == 1 and "row blocks" in d["config"]["parallelism"] and d["value"] > 0


def test_bench_in_flight_option():
    """--in-flight 2: two independent updates on two streams / two engine contexts (opt-in secondary figure)."""
    d = _run({}, "--steps", "42", "--warmup", "21", "--no-cpu-baseline", "--in-flight", "2")
    assert d["value"] > 0 and d["value_in_flight"]["in_flight"] == 2 and d["value_in_flight"]["updates_per_s"] > 0
