#!/usr/bin/env python3
"""A/B: factor-form fit rate by the split-K count of the (non-transposed) panel products (knob panel_kc; 0 = automatic).
In the fit the factor is cache-warm, unlike in scripts/ab_panel_kc.py's HBM-cold ring.  usage: fit_kc_ab.py [D B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
import _inputs as orc
D, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 32)
eng = gsmvi_amd.get_engine()
m, cov, P = orc.make_gaussian_target(D, 1)
tgt = gsmvi_amd.GaussianTarget(m, precision=P)
for kc in (0, 1, 2, 4, 0, 1):
    eng.set_tuning("panel_kc", kc)
    gsm = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
    gsm.fit(1, niter=50, batch_size=B, verbose=False)
    torch.cuda.synchronize()
    n, t0 = 2000, time.perf_counter()
    gsm.fit(1, niter=n - 1, batch_size=B, verbose=False)
    torch.cuda.synchronize()
    print(f"D={D} B={B} panel_kc={kc}: fit {n / (time.perf_counter() - t0):.0f} it/s")
eng.set_tuning("panel_kc", 0)
