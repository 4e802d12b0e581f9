#!/usr/bin/env python3
"""A/B: V Fm product of the n = 128 factor update on the context's side stream (fork_min_D <= D) against one stream
(fork_min_D = 0).  usage: fork_ab.py [D B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
D, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 64)
eng = gsmvi_amd.get_engine()
g = torch.Generator(device="cuda"); g.manual_seed(0)
kw = dict(dtype=torch.float64, device="cuda", generator=g)
A = torch.randn(D, D, **kw); S0 = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device="cuda")
F0 = torch.linalg.cholesky(S0).T.contiguous(); mu0 = torch.randn(D, **kw); Z = torch.randn(B, D, **kw)
X = (mu0 + Z @ F0).contiguous(); G = -(X - 0.5)
res = {}
for fork in (0, 1, 0, 1):
    eng.set_tuning("fork_min_D", 64 if fork else 0)
    mu, Fo, flag = eng.empty(D), eng.empty(D, D), eng.new_flag()
    f = lambda: eng.gsm_factor_update(Z, X, G, mu0, F0, out=(mu, Fo), flag=flag)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    res[fork] = (mu.clone(), Fo.clone())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    t_e = e0.elapsed_time(e1) * 1e3 / 20
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(4):
            f()
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gr.replay()
    torch.cuda.synchronize()
    print(f"D={D} B={B} fork={fork}: factor update {t_e:.1f} us eager, {(time.perf_counter() - t0) / 40 * 1e6:.1f} us graph")
print("fork vs one stream, max rel diff of F:", float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max()),
      "(0 with wide=0: same kernels on both sides; the forked product carries no side job and takes the wide kernel)")
eng.set_tuning("fork_min_D", 3072)
