#!/usr/bin/env python3
"""A/B: the factor-form fit iteration (randn -> sample -> score -> factor update, ping-pong state) issued eagerly from Python
against the same launches captured once into a hipGraph and replayed.  usage: fit_graph_ab.py [D B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
D, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 32)
eng = gsmvi_amd.get_engine()
g = torch.Generator(device="cuda"); g.manual_seed(0)
L = torch.randn(D, D, dtype=torch.float64, device="cuda", generator=g)
P = torch.linalg.inv(L @ L.T + 1e-3 * torch.eye(D, dtype=torch.float64, device="cuda")); P = (0.5 * (P + P.T)).contiguous()
m = torch.rand(D, dtype=torch.float64, device="cuda", generator=g)
mu = [eng.zeros(D), eng.empty(D)]; F = [eng.eye(D), eng.empty(D, D)]
Z, X, G = eng.empty(B, D), eng.empty(B, D), eng.empty(B, D)
flag, nrev = eng.new_flag(), eng.new_flag()
def body(i, a):
    eng.normal(B, D, 7, i, out=Z)
    eng.sample(Z, mu[a], F[a], out=X)
    eng.gaussian_score(X, m, P, out=G)
    eng.gsm_factor_update(Z, X, G, mu[a], F[a], out=(mu[1 - a], F[1 - a]), flag=flag, n_reverts=nrev)
N = 200
for i in range(20): body(i, i & 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N): body(i, i & 1)
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / N * 1e6
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for i in range(20): body(i, i & 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N // 20): gr.replay()
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / N * 1e6
print(f"D={D} B={B}: eager {te:.1f} us/iteration ({1e6 / te:.0f} it/s), graph replay {tg:.1f} us/iteration ({1e6 / tg:.0f} it/s)")
