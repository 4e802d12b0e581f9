"""Diagnostic: the factor form on DEGENERATE inputs (isotropic state on an isotropic target: every u_b - a_b z_b is
parallel to mu - m, so [Z; U] has rank B + 1 < 2B and its Gram matrix is singular)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsmvi_amd
for D, B in ((4, 2), (8, 2), (64, 8), (256, 32), (300, 64)):
    for method in ("dense", "factor"):
        calls = [0]
        def lp_g(x):
            calls[0] += 1
            return -2.0 * (x - 0.5)
        g = gsmvi_amd.GSM(D, None, lp_g)
        mean, cov = g.fit(1, niter=60, batch_size=B, verbose=False, method=method)
        print(D, B, method, "reverts", g.n_reverts, "err mean %.2e cov %.2e" % (np.abs(mean - 0.5).max(), np.abs(cov - 0.5 * np.eye(D)).max()))
