#!/usr/bin/env python3
"""Round-5 verdict, item 2: batch sizes beyond the old ceilings.  U (one-shot update, back-to-back median of event pairs and
replayed) for the GSM dense update and the BaM dense update, and F (fit iterations/s, dense loops -- what method="auto"
takes for 2B > 256) at D = 1024 for B = 128, 130, 160, 256, 512 (+ 640, 1024 for the lifted BaM bound), with the kernel
families that ran (gsmvi_last_path) and the per-sample cost relative to B = 128.
usage: bigbatch_bench.py out.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gsmvi_amd  # noqa: E402

eng = gsmvi_amd.get_engine()
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bigbatch.json"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
BS = [int(b) for b in sys.argv[3].split(",")] if len(sys.argv) > 3 else [128, 130, 160, 256, 512, 640, 1024]


def back_to_back(fn, warm, n):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def state(B, seed):
    g = torch.Generator(device=eng.device)
    g.manual_seed(100 + seed)
    kw = dict(dtype=torch.float64, device=eng.device, generator=g)
    m = torch.rand(D, **kw)
    L = torch.randn(D, D, **kw)
    cov_t = L @ L.T + 1e-3 * torch.eye(D, dtype=torch.float64, device=eng.device)
    P = torch.linalg.inv(cov_t)
    P = (0.5 * (P + P.T)).contiguous()
    mu0 = torch.randn(D, **kw)
    A = torch.randn(D, D, **kw)
    S0 = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device=eng.device)
    S0 = (0.5 * (S0 + S0.T)).contiguous()
    F0 = torch.linalg.cholesky(S0).T.contiguous()
    Z = torch.randn(B, D, **kw)
    X = (mu0[None, :] + Z @ F0).contiguous()
    G = eng.gaussian_score(X, m, P)
    return dict(m=m, P=P, mu0=mu0, S0=S0, X=X, G=G)


res = {"D": D, "device": torch.cuda.get_device_name(0), "rows": {}}
for B in BS:
    st = state(B, 1)
    mu, S = eng.empty(D), eng.empty(D, D)
    flag = eng.new_flag()
    row = {"B": B}
    eng.last_path()
    f_gsm = lambda: eng.gsm_update(st["X"], st["G"], st["mu0"], st["S0"], out=(mu, S))
    f_gsm()
    row["gsm_path"] = sorted(eng.last_path())
    row["U_gsm_dense_us"] = back_to_back(f_gsm, 10, 200)
    try:
        f_bam = lambda: eng.bam_update(st["X"], st["G"], st["mu0"], st["S0"], 1.0, 1e-6, out=(mu, S), flag=flag)
        f_bam()
        row["bam_path"] = sorted(eng.last_path())
        row["U_bam_dense_us"] = back_to_back(f_bam, 5, 60)
        row["bam_flag"] = eng.read_flag(flag)
    except Exception as e:                                    # noqa: BLE001
        row["U_bam_dense_us"] = None
        row["bam_error"] = f"{type(e).__name__}: {e}"
    tgt = gsmvi_amd.GaussianTarget(st["m"].cpu().numpy(), precision=st["P"].cpu().numpy())
    for kind in ("gsm", "bam"):
        try:
            if kind == "gsm":
                fitter = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
                run = lambda n: fitter.fit(1, niter=n, batch_size=B, verbose=False)
            else:
                fitter = gsmvi_amd.BaM(D, tgt.lp, tgt.lp_g)
                run = lambda n: fitter.fit(1, lambda i: 100.0 / (1 + i), niter=n, batch_size=B, verbose=False)
            run(4)
            torch.cuda.synchronize()
            n = 100
            t0 = time.perf_counter()
            run(n - 1)
            torch.cuda.synchronize()
            t1 = time.perf_counter() - t0
            t0 = time.perf_counter()
            run(3 * n - 1)
            torch.cuda.synchronize()
            t3 = time.perf_counter() - t0
            row[f"F_{kind}"] = {"method": fitter.method_used, "it_per_s": n / t1, "it_per_s_marginal": 2 * n / (t3 - t1),
                                "iteration_us_marginal": (t3 - t1) / (2 * n) * 1e6, "n_reverts": int(fitter.n_reverts)}
        except Exception as e:                                # noqa: BLE001
            row[f"F_{kind}"] = {"error": f"{type(e).__name__}: {e}"}
    res["rows"][str(B)] = row
    print(json.dumps(row), flush=True)
    del st, mu, S, tgt
    torch.cuda.empty_cache()
ref = res["rows"].get("128")
if ref:
    for b, row in res["rows"].items():
        row["per_sample_vs_B128"] = {
            k: (row[k] / row["B"]) / (ref[k] / 128.0) for k in ("U_gsm_dense_us", "U_bam_dense_us") if row.get(k) and ref.get(k)}
        for kind in ("gsm", "bam"):
            a, r0 = row.get(f"F_{kind}", {}), ref.get(f"F_{kind}", {})
            if "iteration_us_marginal" in a and "iteration_us_marginal" in r0:
                row["per_sample_vs_B128"][f"F_{kind}"] = (a["iteration_us_marginal"] / row["B"]) / (r0["iteration_us_marginal"] / 128.0)
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({b: r.get("per_sample_vs_B128") for b, r in res["rows"].items()}, indent=1))
