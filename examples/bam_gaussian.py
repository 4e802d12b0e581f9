#!/usr/bin/env python3
"""BaM fit of a Gaussian target through the drop-in API (the workflow of the reference's examples/example_bam.py,
written against gsmvi_amd): decaying regulariser 100 / (1 + i), low-rank update form.

    python examples/bam_gaussian.py [D] [batch] [niter] [method]

method = "dense" (default: the reference's loop, covariance kept, one Cholesky per iteration) or "factor" (state kept as a
square factor, no D x D factorisation per iteration; needs 2 * batch <= min(D, 128)).
"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gsmvi_amd

D = int(sys.argv[1]) if len(sys.argv) > 1 else 5
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
niter = int(sys.argv[3]) if len(sys.argv) > 3 else 100
method = sys.argv[4] if len(sys.argv) > 4 else "dense"

rs = np.random.RandomState(1)
mean = rs.random_sample(D)
L = rs.normal(size=(D, D))
cov = L @ L.T + 1e-3 * np.eye(D)
tgt = gsmvi_amd.GaussianTarget(mean, cov)

reg = gsmvi_amd.Regularizers()
bam = gsmvi_amd.BaM(D, tgt.lp, tgt.lp_g, use_lowrank=True)
mean_fit, cov_fit = bam.fit(99, regf=reg.custom(lambda i: 100.0 / (1 + i)), niter=niter, batch_size=batch,
                            verbose=False, method=method)
print(f"BaM ({method}):", "mean ok" if np.allclose(mean, mean_fit, atol=1e-3) else "mean differs",
      "| cov ok" if np.allclose(cov, cov_fit, rtol=1e-3, atol=1e-3) else "| cov differs",
      f"| {reg.counter} regulariser calls, {bam.n_reverts} reverts")
