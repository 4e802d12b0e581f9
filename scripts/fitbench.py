"""Fit iteration rates (dense and factor form) on a Gaussian target.  usage: fitbench.py D B [niter]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
D, B = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
g = torch.Generator(device="cuda"); g.manual_seed(0)
L = torch.randn(D, D, dtype=torch.float64, device="cuda", generator=g)
P = torch.linalg.inv(L @ L.T + 1e-3 * torch.eye(D, dtype=torch.float64, device="cuda"))
P = (0.5 * (P + P.T)).cpu().numpy(); m = np.random.RandomState(0).random_sample(D)
tgt = gsmvi_amd.GaussianTarget(m, precision=P)
for method in ("dense", "factor"):
    gsm = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
    gsm.fit(1, niter=5, batch_size=B, verbose=False, rng="device", method=method)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gsm.fit(1, niter=n - 1, batch_size=B, verbose=False, rng="device", method=method)
    torch.cuda.synchronize()
    print(f"D={D} B={B} {method}: {n / (time.perf_counter() - t0):.0f} it/s reverts {gsm.n_reverts}")
