import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
from oracle import gsm_oracle as orc
D, B = int(sys.argv[1]), int(sys.argv[2])
m, cov_t, P = orc.make_gaussian_target(D, 0)
tgt = gsmvi_amd.GaussianTarget(m, precision=P)
gsm = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
gsm.fit(1, niter=40, batch_size=B, verbose=False, rng="device", method=sys.argv[3] if len(sys.argv) > 3 else "factor")
torch.cuda.synchronize()
