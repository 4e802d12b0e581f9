"""CPU: bench.py's call-path estimator.  A host score whose own time differs from RUN to run (a many-threaded numpy GEMM does:
0.9 ... 3 ms per call on the GPU box) must not leak into the engine's overhead: the callable's time has to come from the same
runs as the rate.  (Before this held, one published pass showed an iteration of 47 us for a 2.7 ms callable.)"""
import importlib.util
import os
import time

import pytest

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_marginal_rate_subtracts_the_callable_of_the_same_runs(monkeypatch):
    b = _bench()
    monkeypatch.setattr(b.torch.cuda, "synchronize", lambda *a, **k: None)      # no GPU here: the estimator's arithmetic only
    per_call = iter([0.0008, 0.0030, 0.0011, 0.0027, 0.0009, 0.0033, 0.0010] * 10)   # the callable's time, by RUN
    engine_per_iter = 0.0002

    def busy(dt):
        t_end = time.perf_counter() + dt
        while time.perf_counter() < t_end:
            pass

    state = {"dt": 0.001}

    def host_fn(x):
        busy(state["dt"])
        return x

    timed = b._TimedCallable(host_fn)

    def run(k):
        state["dt"] = next(per_call)
        for _ in range(k):
            busy(engine_per_iter)
            timed(0)

    n = 10
    rate = b._marginal_rate(run, n, timed=timed)
    iteration = 1.0 / rate
    outside = iteration - timed.marginal_fn_s                     # what the table calls overhead + native iteration
    assert outside == pytest.approx(engine_per_iter, rel=0.35), (iteration, timed.marginal_fn_s)
    # the old subtraction (callable averaged over ALL runs) is off by the callable's swing on the same data
    naive = iteration - timed.t / timed.calls
    assert abs(naive - engine_per_iter) > 2 * abs(outside - engine_per_iter)
