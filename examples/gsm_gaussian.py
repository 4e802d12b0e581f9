#!/usr/bin/env python3
"""GSM fit of a Gaussian target through the drop-in API (the workflow of the reference's
examples/example_gsm_numpy.py and examples/example_gsm.py, written against gsmvi_amd).

    python examples/gsm_gaussian.py [D] [batch] [niter] [dense|factor]
"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gsmvi_amd

D = int(sys.argv[1]) if len(sys.argv) > 1 else 5
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
niter = int(sys.argv[3]) if len(sys.argv) > 3 else 500
method = sys.argv[4] if len(sys.argv) > 4 else "dense"

rs = np.random.RandomState(0)
mean = rs.random_sample(D)
L = rs.normal(size=(D, D))
cov = L @ L.T + 1e-3 * np.eye(D)

# 1. the reference's calling convention: numpy score callback, numpy results
icov = np.linalg.inv(cov)
lp = lambda x: -0.5 * np.einsum("bi,ij,bj->", x - mean, icov, x - mean)
lp_g = lambda x: -(x - mean) @ icov
gsm = gsmvi_amd.GSM(D=D, lp=lp, lp_g=lp_g)
mean_fit, cov_fit = gsm.fit(key=99, niter=niter, batch_size=batch, verbose=False, method=method)
print("numpy callback :", "mean ok" if np.allclose(mean, mean_fit, atol=1e-4) else "mean differs",
      "| cov ok" if np.allclose(cov, cov_fit, rtol=1e-3, atol=1e-4) else "| cov differs")

# 2. everything on the device: built-in Gaussian score kernel, counter-based draw stream, KL monitor
tgt = gsmvi_amd.GaussianTarget(mean, cov)
ref = rs.multivariate_normal(mean, cov, 1000)
mon = gsmvi_amd.KLMonitor(batch_size_kl=32, checkpoint=max(1, niter // 10), ref_samples=ref)
mean_fit, cov_fit = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g).fit(99, niter=niter, batch_size=batch, verbose=False,
                                                          rng="device", method=method, monitor=mon)
print("device-native  :", "mean ok" if np.allclose(mean, mean_fit, atol=1e-4) else "mean differs",
      "| cov ok" if np.allclose(cov, cov_fit, rtol=1e-3, atol=1e-4) else "| cov differs",
      f"| reverse KL {mon.rkl[0]:.3g} -> {mon.rkl[-1]:.3g} over {mon.nevals[-1]} gradient evaluations")
