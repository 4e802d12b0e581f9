import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
import _inputs as orc
D, B = int(sys.argv[1]), int(sys.argv[2])
m, cov_t, P = orc.make_gaussian_target(D, 0)
tgt = gsmvi_amd.GaussianTarget(m, precision=P)
method = sys.argv[3] if len(sys.argv) > 3 else "factor"
if method == "bamf":
    gsmvi_amd.BaM(D, tgt.lp, tgt.lp_g).fit(1, gsmvi_amd.Regularizers().constant(1.0), niter=40, batch_size=B, verbose=False,
                                          method="factor", jitter=0.0)
elif method == "bam":
    gsmvi_amd.BaM(D, tgt.lp, tgt.lp_g).fit(1, gsmvi_amd.Regularizers().constant(1.0), niter=40, batch_size=B, verbose=False,
                                          method="dense")
else:
    # graph=False: every launch of every iteration goes through the profiler's kernel trace
    gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g).fit(1, niter=40, batch_size=B, verbose=False, rng="device", method=method, graph=False)
torch.cuda.synchronize()
