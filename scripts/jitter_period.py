#!/usr/bin/env python3
"""Round-5 verdict, item 1: how far is the factor-form BaM fit from the reference's loop (bam.py:198: + jitter * I after every
update) when the owed jitter is absorbed every K accepted updates (F <- chol(F^T F + owed I)), and what does it cost?
For each shape and K: the factor fit's own samples are recorded and forced into the reference-faithful dense loop
(jitter = 1e-6); deviation = max over 20 checkpoints of max|cov_f - cov_d| / max|cov_d| (BASELINE.json's metric); rate =
iterations/s of an unmonitored fit with the built-in device score.  usage: jitter_period.py out.json [niter]"""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import gsmvi_amd                                                   # noqa: E402
from gsmvi_amd.targets import GaussianTarget, device_score         # noqa: E402

out = sys.argv[1]
niter = int(sys.argv[2]) if len(sys.argv) > 2 else 500
eng = gsmvi_amd.get_engine()
res = {"what": __doc__, "niter": niter, "jitter": 1e-6, "shapes": {}}
for D, B in ((1024, 128), (1024, 32)):
    g = torch.Generator(device=eng.device)
    g.manual_seed(5)
    kw = dict(dtype=torch.float64, device=eng.device, generator=g)
    m = torch.rand(D, **kw)
    L = torch.randn(D, D, **kw)
    cov_t = L @ L.T + 1e-3 * torch.eye(D, dtype=torch.float64, device=eng.device)    # examples/example_bam.py:20-23, seeded
    P = torch.linalg.inv(cov_t)
    tgt = GaussianTarget(m.cpu().numpy(), precision=(0.5 * (P + P.T)).cpu().numpy())
    sched = lambda c: 100.0 / c                                   # noqa: E731

    class Snap:
        checkpoint = 25
        device_native = True

        def __init__(self):
            self.store = {}

        def __call__(self, i, params, lp, key, nevals=1):
            self.store[i] = params[1].clone()

    def rate(**kwargs):
        n = 300 if B >= 128 else 600
        bam = gsmvi_amd.BaM(D, None, tgt.lp_g)
        bam.fit(7, gsmvi_amd.Regularizers().custom(sched), batch_size=B, niter=40, verbose=False, as_torch=True, **kwargs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bam.fit(7, gsmvi_amd.Regularizers().custom(sched), batch_size=B, niter=n, verbose=False, as_torch=True, **kwargs)
        torch.cuda.synchronize()
        return (n + 1) / (time.perf_counter() - t0)

    rows = {}
    for K in (0, 1, 2, 4, 8, 16):
        seen = []

        @device_score
        def lp_g(x):
            seen.append(x.clone())
            return tgt.lp_g(x)

        sf, sd = Snap(), Snap()
        bam = gsmvi_amd.BaM(D, None, lp_g)
        bam.fit(7, gsmvi_amd.Regularizers().custom(sched), batch_size=B, niter=niter, verbose=False, monitor=sf, as_torch=True,
                method="factor", jitter=1e-6, jitter_every=K)
        assert bam.method_used == "factor" and bam.n_reverts == 0
        bd = gsmvi_amd.BaM(D, None, tgt.lp_g)
        bd.fit(7, gsmvi_amd.Regularizers().custom(sched), batch_size=B, niter=niter, verbose=False, jitter=1e-6,
               forced_samples=seen, monitor=sd, as_torch=True, method="dense")
        assert bd.n_reverts == 0
        devs = {i: float((sf.store[i] - sd.store[i]).abs().max() / sd.store[i].abs().max()) for i in sorted(sf.store) if i > 0}
        rows[str(K)] = {"max_dev": max(devs.values()), "dev_at_end": devs[max(devs)], "n_absorbed": bam.n_absorbed,
                        "fit_it_per_s": rate(method="factor", jitter=1e-6, jitter_every=K)}
        print(D, B, "K =", K, rows[str(K)], flush=True)
    rows["dense"] = {"max_dev": 0.0, "fit_it_per_s": rate(method="dense", jitter=1e-6)}
    rows["factor_jitter0"] = {"fit_it_per_s": rate(method="factor", jitter=0.0)}
    print(D, B, "dense", rows["dense"], "factor j=0", rows["factor_jitter0"], flush=True)
    res["shapes"][f"{D}x{B}"] = rows
json.dump(res, open(out, "w"), indent=1)
