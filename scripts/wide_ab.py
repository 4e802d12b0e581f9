#!/usr/bin/env python3
"""A/B: 64-row panel products on the 64 x 64-tile kernels (gsmvi_wide.hip, knob wide=1) against the 16-column strips
(wide=0): sampler, Gaussian score, dense update and factor update at B = 64.  usage: wide_ab.py [D]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
D = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = 64
eng = gsmvi_amd.get_engine()
g = torch.Generator(device="cuda"); g.manual_seed(0)
kw = dict(dtype=torch.float64, device="cuda", generator=g)
A = torch.randn(D, D, **kw); S0 = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device="cuda")
F0 = torch.linalg.cholesky(S0).T.contiguous(); mu0 = torch.randn(D, **kw); Z = torch.randn(B, D, **kw)
P = torch.linalg.inv(S0); P = (0.5 * (P + P.T)).contiguous(); m = torch.randn(D, **kw)
X = eng.sample(Z, mu0, F0); G = eng.gaussian_score(X, m, P)

def gtime(f, reps=4, nrep=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            f()
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nrep):
        gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * nrep) * 1e6

res = {}
kcs = [int(a) for a in sys.argv[2:]] or [0]
for wide, kc in [(0, 0)] + [(1, k) for k in kcs]:
    eng.set_tuning("wide", wide)
    eng.set_tuning("wide_kc", kc)
    Xo, Go = eng.empty(B, D), eng.empty(B, D)
    mu, Fo, So, flag = eng.empty(D), eng.empty(D, D), eng.empty(D, D), eng.new_flag()
    t_s = gtime(lambda: eng.sample(Z, mu0, F0, out=Xo))
    t_g = gtime(lambda: eng.gaussian_score(X, m, P, out=Go))
    t_f = gtime(lambda: eng.gsm_factor_update(Z, X, G, mu0, F0, out=(mu, Fo), flag=flag))
    t_d = gtime(lambda: eng.gsm_update(X, G, mu0, S0, out=(mu, So)))
    torch.cuda.synchronize()
    res[(wide, kc)] = [t.clone() for t in (Xo, Go, Fo, So)]
    print(f"D={D} B={B} wide={wide} kc={kc}: sample {t_s:.1f} us, score {t_g:.1f} us, factor update {t_f:.1f} us, dense update {t_d:.1f} us")
ref = res[(0, 0)]
for k, v in res.items():
    if k != (0, 0):
        print(k, "max rel diff vs narrow:", [f"{float((a - b).abs().max() / b.abs().max()):.1e}" for a, b in zip(v, ref)])
eng.set_tuning("wide", 1); eng.set_tuning("wide_kc", 0)
