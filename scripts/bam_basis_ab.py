#!/usr/bin/env python3
"""A/B of the factor-form BaM update in the round-4 basis [Vw; Zw] (knob bam_basis=0) against the orthogonal basis [Vw; Zt]
(bam_basis=1, round 5): update time (eager median, replayed graph) and the marginal fit rate.  usage: bam_basis_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
import _inputs as orc
eng = gsmvi_amd.get_engine()
for D, B in ((256, 8), (1024, 32), (1024, 64), (1024, 128), (4096, 64)):
    st = orc.make_update_state(D, B, 1) if D <= 1024 else None
    if st is None:
        g = torch.Generator(device=eng.device); g.manual_seed(1)
        kw = dict(dtype=torch.float64, device=eng.device, generator=g)
        A = torch.randn(D, D, **kw); S0 = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device=eng.device)
        F0, _ = eng.potrf(S0); mu0 = torch.randn(D, **kw); Z = torch.randn(B, D, **kw); X = eng.sample(Z, mu0, F0)
        G = -(X - 0.3) * 0.5
    else:
        X, G, mu0, Z = (eng.asarray(st[k]) for k in ("samples", "vs", "mu0", "Z"))
        F0 = eng.asarray(st["L"].T.copy())
    out = (eng.empty(D), eng.empty(D, D)); flag = eng.new_flag()
    call = lambda: eng.bam_factor_update(Z, X, G, mu0, F0, 1.0, out=out, flag=flag)
    m, P = torch.rand(D, dtype=torch.float64, device=eng.device), None
    for knob in (1, 3, 0):                          # 1 = default; 3 = the chain factors Gamma11 itself (round-5 A/B); 0 = round-4 basis
        eng.set_tuning("bam_basis", knob)
        for _ in range(10): call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(100):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); call(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(4): call()
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 40 * 1e6
        assert eng.read_flag(flag) == 0
        print(f"D={D} B={B} bam_basis={knob}: update eager median {np.median(ts):7.1f} us  replayed {tg:7.1f} us", flush=True)
    eng.set_tuning("bam_basis", 1)
