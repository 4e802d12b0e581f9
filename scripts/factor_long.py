import sys
sys.path.insert(0, "/root/repo")
import numpy as np, gsmvi_amd
from oracle import gsm_oracle as orc
for D, B, n in ((10, 2, 5000), (16, 8, 5000), (64, 8, 3000)):
    m, cov_t, P = orc.make_gaussian_target(D, 3)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    for method in ("dense", "factor"):
        gsm = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
        mean, cov = gsm.fit(5, niter=n, batch_size=B, verbose=False, rng="device", method=method)
        print(f"D={D} B={B} niter={n} {method}: mean err {np.abs(mean-m).max():.2e} cov rel err {np.abs(cov-cov_t).max()/np.abs(cov_t).max():.2e} reverts {gsm.n_reverts} finite {np.isfinite(cov).all()}")
