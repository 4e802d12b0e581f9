"""Factor-form BaM update against the dense BaM update (jitter 0) on the same inputs: F^T F vs S, means."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import importlib
pkg = importlib.import_module("gsm-vi_amd")
from importlib import import_module
eng = import_module("gsm-vi_amd.engine").get_engine()

def run(D, B, reg, seed=0):
    rs = np.random.RandomState(seed)
    A = rs.standard_normal((D, D)) / np.sqrt(D)
    S0 = A @ A.T + 0.5 * np.eye(D)
    mu0 = rs.standard_normal(D)
    S0t = eng.asarray(S0)
    F0, _ = eng.potrf(S0t)
    Z = eng.asarray(rs.standard_normal((B, D)))
    X = eng.sample(Z, eng.asarray(mu0), F0)
    # target: Gaussian with a different covariance
    Bm = rs.standard_normal((D, D)) / np.sqrt(D)
    P = eng.asarray(np.linalg.inv(Bm @ Bm.T + 0.3 * np.eye(D)))
    m = eng.asarray(rs.standard_normal(D))
    G = -(X - m) @ P
    S0f = eng.gram(F0)
    mu_d, S_d, fl = eng.bam_update(X, G, eng.asarray(mu0), S0f, reg, 0.0)
    mu_f, F, flf = eng.bam_factor_update(Z, X, G, eng.asarray(mu0), F0, reg)
    S_f = eng.gram(F)
    torch.cuda.synchronize()
    es = float((S_f - S_d).abs().max() / S_d.abs().max())
    em = float((mu_f - mu_d).abs().max() / (1 + mu_d.abs().max()))
    print(f"D={D} B={B} reg={reg}: flags {int(fl.item())} {int(flf.item())}  rel err S {es:.2e}  mean {em:.2e}")

for D, B in ((64, 8), (256, 8), (1024, 32), (1024, 16), (512, 7), (300, 20), (1024, 64), (128, 1), (130, 33)):
    for reg in (1.0, 50.0):
        run(D, B, reg)
