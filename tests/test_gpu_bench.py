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
    # round 5: the drop-in call path (host numpy score, torch-autograd score) is part of the line
    f = d["fit_iterations_per_s"]
    assert isinstance(f["host_lp_g"], float) and isinstance(f["autograd_lp_g"], float) and f["host_lp_g"] > 0
    assert "c2" in f["call_path"] and f["call_path"]["here"]["auto"]["host_lp_g_diag_target"]["overhead_us"] < 200
    assert d["config"]["workload"].startswith("BASELINE configs[2]")


def test_bench_rccl_path_world1():
    d = _run({"GSMVI_BENCH_FORCE_DIST": "1"}, "--steps", "42", "--warmup", "21", "--no-cpu-baseline", "--no-callpath")
    assert d["n_gpus"] == 1 and "RCCL" in d["config"]["parallelism"] and d["value"] > 0


def test_bench_row_block_path_world1():
    d = _run({"GSMVI_BENCH_FORCE_DIST": "1"}, "--steps", "42", "--warmup", "21", "--no-cpu-baseline", "--no-callpath", "--shard", "rows")
    assert d["n_gpus"] == 1 and "row blocks" in d["config"]["parallelism"] and d["value"] > 0


def test_bench_in_flight_option():
    """--in-flight 2: two independent updates on two streams / two engine contexts (opt-in secondary figure)."""
    d = _run({}, "--steps", "42", "--warmup", "21", "--no-cpu-baseline", "--no-callpath", "--in-flight", "2")
    assert d["value"] > 0 and d["value_in_flight"]["in_flight"] == 2 and d["value_in_flight"]["updates_per_s"] > 0


def test_bench_driver_flags_time_the_graph_path_they_name():
    """The driver's own invocation (--steps 20 --warmup 5, fewer steps than ring instances): every timed step is a
    hipGraph replay and the label says what ran (round-1 verdict: those 20 steps used to be eager launches)."""
    d = _run({}, "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-callpath")
    assert d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["launch"] == "hipGraph(1 x 20 updates/replay)", d["config"]["launch"]
    assert d["config"]["warmup_steps_run"] >= 5
    # (advisor, round 4) a RELATIVE guard in place of an absolute rate: the same command with eager launches, same box, same run.
    # What it can assert is bounded by the host: three launches per 18 us update are within reach of a fast host, so eager
    # has been seen anywhere from 36.5k (rounds 2-4) to 55.7k updates/s (round 5: ABOVE the replayed graph's 52.4k of the same
    # run) while the graph stays at 51-55k.  The guard therefore says "the replayed path is not grossly slower than eager
    # launches" (a graph that re-instantiates or serialises would be), at 210 steps (~4 ms of timed region: the 20-step region
    # is one 0.4 ms replay, too short for any ratio); that the timed steps ARE replays is what the launch label asserts.
    dg = _run({}, "--steps", "210", "--warmup", "21", "--no-cpu-baseline", "--no-callpath", "--no-large-point")
    de = _run({}, "--steps", "210", "--warmup", "21", "--no-cpu-baseline", "--no-callpath", "--no-graph", "--no-large-point")
    assert "hipGraph" in dg["config"]["launch"] and "hipGraph" not in de["config"]["launch"], (dg["config"]["launch"], de["config"]["launch"])
    assert dg["value"] > 0.8 * de["value"], (dg["value"], de["value"])
    d = _run({}, "--steps", "50", "--warmup", "5", "--no-cpu-baseline", "--no-callpath")
    assert d["config"]["launch"] == "hipGraph(2 x 21 updates/replay + 1 x 8)", d["config"]["launch"]
    assert "traffic_source" in d["roofline"] and d["roofline"]["moved_bytes_per_launch"] < \
        d["roofline"]["algorithmic_bytes_per_launch"]


def test_bench_refuses_more_gpus_than_visible():
    """--gpus N without a launcher starts its own N ranks; with fewer than N devices it must fail loudly instead of
    reporting a 1-GPU number as an N-GPU one."""
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "GPU(s) are visible" in p.stderr and '"metric"' not in p.stdout
