#!/usr/bin/env python3
"""U (update-only) and F (fit-iteration) rates for every BASELINE.json config that fits one GPU
(report both, median and min), with the CPU restatement timed beside them on a bounded sample.

  c2  D=256,  B=8    GSM dense update            c4  D=1024, B=128  BaM update (un-sharded on one GPU)
  c3  D=1024, B=32   GSM dense update            c5  D=4096, B=64   ill-conditioned target (cond 1e8):
                                                                    dense update, factor update, both fits
Writes one JSON document (argv[1], default gpurun_out/configs.json).  Update timings: HIP events around
single calls on the current stream after warm-up (median / min over the trials), cache-resident inputs.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gsmvi_amd  # noqa: E402
from oracle import gsm_oracle as orc  # noqa: E402   (checker / CPU baseline only)
from oracle import bam_oracle as borc  # noqa: E402

eng = gsmvi_amd.get_engine()
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/configs.json"


def ev_times(fn, warm, n):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = np.array(ts)
    return {"median_us": float(np.median(ts)), "min_us": float(ts.min()), "n": int(n)}


def graph_time(fn, reps, nrep):
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nrep):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (nrep * reps) * 1e6


def cpu_time(fn, budget_s, nmax=50):
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= nmax:
            return {"ms": el / n * 1e3, "n": n}


def gpu_state(D, B, seed, cond=None):
    """SURVEY 8(d) synthetic inputs, generated on the device (the oracle's make_update_state is O(D^3) numpy)."""
    g = torch.Generator(device=eng.device)
    g.manual_seed(100 + seed)
    kw = dict(dtype=torch.float64, device=eng.device, generator=g)
    m = torch.rand(D, **kw)
    L = torch.randn(D, D, **kw)
    cov_t = L @ L.T + 1e-3 * torch.eye(D, dtype=torch.float64, device=eng.device)
    if cond is not None:
        w, Q = torch.linalg.eigh(cov_t)
        w = torch.logspace(0.0, float(np.log10(cond)), D, dtype=torch.float64, device=eng.device) * w.min().clamp(min=1e-12)
        cov_t = (Q * w[None, :]) @ Q.T
        cov_t = 0.5 * (cov_t + cov_t.T)
    P = torch.linalg.inv(cov_t)
    P = (0.5 * (P + P.T)).contiguous()
    mu0 = torch.randn(D, **kw)
    A = torch.randn(D, D, **kw)
    S0 = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device=eng.device)
    S0 = (0.5 * (S0 + S0.T)).contiguous()
    F0 = torch.linalg.cholesky(S0).T.contiguous()            # S0 = F0^T F0
    Z = torch.randn(B, D, **kw)
    X = (mu0[None, :] + Z @ F0).contiguous()
    G = eng.gaussian_score(X, m, P)
    return dict(m=m, P=P, cov_t=cov_t, mu0=mu0, S0=S0, F0=F0, Z=Z, X=X, G=G)


def fit_rate(D, B, tgt, method, n):
    gsm = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
    gsm.fit(1, niter=5, batch_size=B, verbose=False, rng="device", method=method)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gsm.fit(1, niter=n - 1, batch_size=B, verbose=False, rng="device", method=method)
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    # a fit has fixed costs (initial Cholesky, the D x D buffers, F^T F at the end: ~20 ms at D = 4096): the MARGINAL
    # iteration time from a second fit three times as long is the rate a long fit sees
    t0 = time.perf_counter()
    gsm.fit(1, niter=3 * n - 1, batch_size=B, verbose=False, rng="device", method=method)
    torch.cuda.synchronize()
    t3 = time.perf_counter() - t0
    return {"it_per_s": n / t1, "n": n, "it_per_s_marginal": 2 * n / (t3 - t1), "reverts": int(gsm.n_reverts)}


res = {"device": torch.cuda.get_device_name(0), "host_cpus": os.cpu_count(), "dtype": "f64", "configs": {}}
for name, D, B, cond in (("c2", 256, 8, None), ("c3", 1024, 32, None), ("c5", 4096, 64, 1e8)):
    st = gpu_state(D, B, 0, cond)
    mu, S, Fn = eng.empty(D), eng.empty(D, D), eng.empty(D, D)
    flag = eng.new_flag()
    r = {"D": D, "B": B, "target_cond": cond}
    small = D <= 1024
    f_dense = lambda: eng.gsm_update(st["X"], st["G"], st["mu0"], st["S0"], out=(mu, S))
    f_factor = lambda: eng.gsm_factor_update(st["Z"], st["X"], st["G"], st["mu0"], st["F0"], out=(mu, Fn), flag=flag)
    f_potrf = lambda: eng.potrf(st["S0"], out=Fn, flag=flag)
    r["U_dense_update"] = ev_times(f_dense, 50 if small else 10, 2000 if small else 60)
    r["U_dense_update"]["graph_us"] = graph_time(f_dense, 40 if small else 8, 20 if small else 5)
    r["U_factor_update"] = ev_times(f_factor, 20 if small else 5, 300 if small else 60)
    r["U_factor_update"]["graph_us"] = graph_time(f_factor, 10 if small else 4, 10 if small else 4)
    r["potrf"] = ev_times(f_potrf, 5, 50 if small else 10)
    alg = 24.0 * D * D + 72.0 * B * D
    r["U_dense_update"]["algorithmic_bytes"] = alg
    r["U_dense_update"]["algorithmic_GBs_graph"] = alg / r["U_dense_update"]["graph_us"] / 1e3
    r["U_dense_update"]["updates_per_s_graph"] = 1e6 / r["U_dense_update"]["graph_us"]
    tgt = gsmvi_amd.GaussianTarget(st["m"].cpu().numpy(), precision=st["P"].cpu().numpy())
    r["F_fit_dense"] = fit_rate(D, B, tgt, "dense", 300 if small else 30)
    r["F_fit_factor"] = fit_rate(D, B, tgt, "factor", 600 if small else 100)
    # CPU: the oracle's faithful port (reference operation order) where its B x D x D temporaries fit, and the
    # batched BLAS-3 form ("best-effort CPU")
    h = {k: st[k].cpu().numpy() for k in ("X", "G", "mu0", "S0")}
    if D <= 1024:
        r["cpu_port_update"] = cpu_time(lambda: orc.gsm_update_faithful(h["X"], h["G"], h["mu0"], h["S0"]), 6.0)
    r["cpu_blas3_update"] = cpu_time(lambda: orc.gsm_update_batched(h["X"], h["G"], h["mu0"], h["S0"]), 4.0)
    r["cpu_cholesky"] = cpu_time(lambda: np.linalg.cholesky(h["S0"]), 3.0)
    res["configs"][name] = r
    print(name, json.dumps(r), flush=True)
    del st, mu, S, Fn, tgt
    torch.cuda.empty_cache()

# ---- c4: BaM, D = 1024, B = 128 (and B = 32 for reference) -----------------------------------------------
for name, D, B in (("c4", 1024, 128), ("c4_B32", 1024, 32)):
    st = gpu_state(D, B, 1)
    mu, S = eng.empty(D), eng.empty(D, D)
    flag = eng.new_flag()
    f_bam = lambda: eng.bam_update(st["X"], st["G"], st["mu0"], st["S0"], 1.0, 1e-6, out=(mu, S), flag=flag)
    r = {"D": D, "B": B, "reg": 1.0, "U_bam_update": ev_times(f_bam, 5, 100)}
    tgt = gsmvi_amd.GaussianTarget(st["m"].cpu().numpy(), precision=st["P"].cpu().numpy())
    bam = gsmvi_amd.BaM(D, tgt.lp, tgt.lp_g)
    sched = lambda i: 100.0 / (1 + i)                             # examples/example_bam.py:58
    bam.fit(1, sched, niter=3, batch_size=B, verbose=False, rng="device", method="dense")
    torch.cuda.synchronize()
    n, t0 = 60, time.perf_counter()
    bam.fit(1, sched, niter=n - 1, batch_size=B, verbose=False, rng="device", method="dense")
    torch.cuda.synchronize()
    r["F_fit_bam"] = {"it_per_s": n / (time.perf_counter() - t0), "n": n,
                      "method": "dense (the reference's loop incl. jitter) = the default, method='auto', at the reference's jitter (round 6)"}
    bam.fit(1, sched, niter=3, batch_size=B, verbose=False, rng="device")                       # every argument at its default
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bam.fit(1, sched, niter=3 * n - 1, batch_size=B, verbose=False, rng="device")
    torch.cuda.synchronize()
    r["F_fit_bam_default_args"] = {"it_per_s": 3 * n / (time.perf_counter() - t0), "n": 3 * n, "method_used": bam.method_used}
    if 2 * B <= 256:                                              # factor form: Sigma = F^T F, no D^3 step per iteration (2B <= 256 since round 4)
        F0, _ = eng.potrf(st["S0"])
        Z = eng.normal(B, D, 5, 0)
        Xf = eng.sample(Z, st["mu0"], F0)
        Gf = tgt.lp_g(Xf)
        Fo = eng.empty(D, D)
        f_bamf = lambda: eng.bam_factor_update(Z, Xf, Gf, st["mu0"], F0, 1.0, out=(mu, Fo), flag=flag)
        r["U_bam_factor_update"] = ev_times(f_bamf, 5, 100)
        r["U_bam_factor_update"]["graph_us"] = graph_time(f_bamf, 4, 4)
        for key, jit, what in (("F_fit_bam_factor", 0.0, "factor, jitter = 0 (= method='auto' for a call without jitter)"),
                               ("F_fit_bam_factor_absorbing", 1e-6, "factor, the reference's jitter absorbed every 4 accepted updates (opt-in)")):
            bam.fit(1, sched, niter=3, batch_size=B, verbose=False, rng="device", method="factor", jitter=jit)
            torch.cuda.synchronize()
            n, t0 = (400 if B <= 64 else 150), time.perf_counter()
            bam.fit(1, sched, niter=n - 1, batch_size=B, verbose=False, rng="device", method="factor", jitter=jit)
            torch.cuda.synchronize()
            t1 = time.perf_counter() - t0
            t0 = time.perf_counter()
            bam.fit(1, sched, niter=3 * n - 1, batch_size=B, verbose=False, rng="device", method="factor", jitter=jit)
            torch.cuda.synchronize()
            t3 = time.perf_counter() - t0
            # it_per_s: the whole call (initial Cholesky, buffers, final Gram product included); marginal: the iteration alone
            r[key] = {"it_per_s": n / t1, "it_per_s_marginal": 2 * n / (t3 - t1), "n": n, "n_reverts": bam.n_reverts, "method": what}
    h = {k: st[k].cpu().numpy() for k in ("X", "G", "mu0", "S0")}
    r["cpu_lowrank_update"] = cpu_time(lambda: borc.bam_lowrank_update_exact(h["X"], h["G"], h["mu0"], h["S0"], 1.0), 4.0)
    res["configs"][name] = r
    print(name, json.dumps(r), flush=True)

os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
