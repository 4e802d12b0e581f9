#!/usr/bin/env python3
"""Diagnostic: 30 factor-form BaM updates for `rocprofv3 --kernel-trace` (scripts/trace_timeline.py prints the last one's
launches with start offsets and queues).  usage: bamf_trace.py D B [bam_basis]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gsmvi_amd
import _inputs as orc
D, B = int(sys.argv[1]), int(sys.argv[2])
eng = gsmvi_amd.get_engine()
if len(sys.argv) > 3 and sys.argv[3].isdigit():
    eng.set_tuning("bam_basis", int(sys.argv[3]))
for kv in sys.argv[4:]:
    if "=" in kv:
        eng.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
if "cfg" in sys.argv:                              # the inputs of scripts/configs_bench.py (a raw L L^T + 1e-3 I target: large scores)
    g = torch.Generator(device=eng.device); g.manual_seed(101)
    kw = dict(dtype=torch.float64, device=eng.device, generator=g)
    m = torch.rand(D, **kw); L = torch.randn(D, D, **kw)
    P = torch.linalg.inv(L @ L.T + 1e-3 * torch.eye(D, dtype=torch.float64, device=eng.device)); P = (0.5 * (P + P.T)).contiguous()
    mu0 = torch.randn(D, **kw); A = torch.randn(D, D, **kw)
    S0 = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device=eng.device)
    F0 = torch.linalg.cholesky(0.5 * (S0 + S0.T)).T.contiguous()
    Z = torch.randn(B, D, **kw); X = (mu0[None, :] + Z @ F0).contiguous(); G = eng.gaussian_score(X, m, P)
else:
    st = orc.make_update_state(D, B, 1)
    X, G, mu0, Z = (eng.asarray(st[k]) for k in ("samples", "vs", "mu0", "Z"))
    F0 = eng.asarray(st["L"].T.copy())
out = (eng.empty(D), eng.empty(D, D)); flag = eng.new_flag()
for _ in range(30):
    eng.bam_factor_update(Z, X, G, mu0, F0, 1.0, out=out, flag=flag)
torch.cuda.synchronize()
import numpy as np
ts = []
for _ in range(60):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.bam_factor_update(Z, X, G, mu0, F0, 1.0, out=out, flag=flag); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print(f"eager median {np.median(ts):.1f} us", file=sys.stderr)
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100):
    eng.bam_factor_update(Z, X, G, mu0, F0, 1.0, out=out, flag=flag)
th = (time.perf_counter() - t0) / 100 * 1e6
torch.cuda.synchronize(); tt = (time.perf_counter() - t0) / 100 * 1e6
print(f"host enqueue {th:.1f} us per update; back-to-back {tt:.1f} us per update", file=sys.stderr)
