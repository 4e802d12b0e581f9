#!/usr/bin/env python3
"""A/B of the factor-form fit rate at BASELINE config 5 (D=4096, B=64, cond-1e8 target) by knob: fork_min_D, wide."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
import _inputs as orc
D, B = 4096, 64
eng = gsmvi_amd.get_engine()
m, cov, P = orc.make_gaussian_target(D, 0, cond=1e8)
tgt = gsmvi_amd.GaussianTarget(m, precision=P)
for fork, wide in ((0, 0), (1, 0), (0, 1), (1, 1)):
    eng.set_tuning("fork_min_D", 3072 if fork else 0)
    eng.set_tuning("wide", wide)
    gsm = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
    gsm.fit(1, niter=5, batch_size=B, verbose=False, rng="device", method="factor")
    torch.cuda.synchronize()
    rates = []
    for n in (100, 300):
        t0 = time.perf_counter()
        gsm.fit(1, niter=n - 1, batch_size=B, verbose=False, rng="device", method="factor")
        torch.cuda.synchronize()
        rates.append((n, time.perf_counter() - t0))
    per_it = (rates[1][1] - rates[0][1]) / 200 * 1e6
    print(f"fork={fork} wide={wide}: {rates[0][0] / rates[0][1]:.0f} it/s over 100, {rates[1][0] / rates[1][1]:.0f} it/s over 300; marginal iteration {per_it:.0f} us, reverts {gsm.n_reverts}")
eng.set_tuning("fork_min_D", 3072); eng.set_tuning("wide", 1)
