"""A/B in one process: fit iterations per second with the panel finishes folded into the product launches (seam_finish=1)
and as their own launches (seam_finish=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
from oracle import gsm_oracle as orc
eng = gsmvi_amd.get_engine()
for D, B in ((1024, 32), (256, 8), (4096, 64)):
    m, cov_t, P = orc.make_gaussian_target(D, 0)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    n = 400 if D <= 1024 else 80
    for method in ("factor", "dense"):
        res = {}
        for rep in range(2):
            for sf in (0, 2):
                eng.set_tuning("seam_finish", sf)
                g = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
                g.fit(1, niter=10, batch_size=B, verbose=False, rng="device", method=method)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                g.fit(1, niter=n - 1, batch_size=B, verbose=False, rng="device", method=method)
                torch.cuda.synchronize(); res.setdefault(sf, []).append(n / (time.perf_counter() - t0))
        print(f"D={D} B={B} {method}: finish kernels {max(res[0]):.0f} it/s, seam {max(res[2]):.0f} it/s")
eng.set_tuning("seam_finish", 1)
