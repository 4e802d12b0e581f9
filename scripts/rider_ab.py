#!/usr/bin/env python3
"""A/B: the 2B x 2B chain as a rider workgroup of the V Fm product launch (rider=1) against its own launch (rider=0):
factor-form update in a replayed graph and the factor-form fit rate.  usage: rider_ab.py [D B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
import _inputs as orc
D, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 32)
eng = gsmvi_amd.get_engine()
m, cov, P = orc.make_gaussian_target(D, 1)
tgt = gsmvi_amd.GaussianTarget(m, precision=P)
mu0 = eng.asarray(m)
F0, _ = eng.potrf(eng.asarray(cov))
Z = eng.normal(B, D, 5, 0)
X = eng.sample(Z, mu0, F0)
G = tgt.lp_g(X)
outs = {}
for rider, gmt in ((0, 4), (1, 4), (1, 2), (1, 1)):
    eng.set_tuning("rider", rider)
    eng.set_tuning("gram_mt", gmt)
    mu, Fo, flag = eng.empty(D), eng.empty(D, D), eng.new_flag()
    f = lambda: eng.gsm_factor_update(Z, X, G, mu0, F0, out=(mu, Fo), flag=flag)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    outs[rider] = (mu.clone(), Fo.clone(), int(flag.item()))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(8):
            f()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    t_up = (time.perf_counter() - t0) / 400 * 1e6
    gsm = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
    gsm.fit(1, niter=50, batch_size=B, verbose=False)
    torch.cuda.synchronize()
    n, t0 = 1500, time.perf_counter()
    gsm.fit(1, niter=n - 1, batch_size=B, verbose=False)
    torch.cuda.synchronize()
    print(f"D={D} B={B} rider={rider} gram_mt={gmt}: update {t_up:.1f} us (graph), fit {n / (time.perf_counter() - t0):.0f} it/s")
print("bit-identical results:", torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), outs[0][2], outs[1][2])
eng.set_tuning("rider", 1)
eng.set_tuning("gram_mt", 2)
