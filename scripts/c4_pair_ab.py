#!/usr/bin/env python3
"""BASELINE config 4 in factor form (BaM, D=1024, B=128): the factor update with the paired launches of the two-level
2B x 2B chain (knob "chain_pair" = 1: Gamma11 beside k_bam_cholw, A'11 beside Gamma's second block) against one launch per
one-workgroup factorisation (0).  HIP events, median / min of 200 eager calls, and replayed from a hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
import _inputs as orc
cases = [(1024, 128), (1024, 96), (2048, 128)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
eng = gsmvi_amd.get_engine()
for D, B in cases:
    rng = np.random.default_rng(D + B)
    mu0 = rng.standard_normal(D)
    F0 = np.linalg.cholesky(np.eye(D) + 0.1 * np.cov(rng.standard_normal((D, 2 * D)))).T
    Z = rng.standard_normal((B, D)); X = mu0 + Z @ F0
    m, _, P = orc.make_gaussian_target(D, 11)
    G = orc.gaussian_score(X, m, P)
    dv = [eng.asarray(a) for a in (Z, X, G, mu0, F0)]
    out = (eng.empty(D), eng.empty(D, D)); flag = eng.new_flag()
    for kind in ("bam", "gsm"):
        call = (lambda: eng.bam_factor_update(*dv, 1.0, out=out, flag=flag)) if kind == "bam" else \
               (lambda: eng.gsm_factor_update(*dv, out=out, flag=flag))
        for pair in (1, 0):
            eng.set_tuning("chain_pair", pair)
            for _ in range(20): call()
            torch.cuda.synchronize()
            ts = []
            for _ in range(200):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); call(); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(10): call()
            for _ in range(3): g.replay()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): g.replay()
            torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 300 * 1e6
            assert eng.read_flag(flag) == 0
            print(f"{kind} factor update D={D} B={B} chain_pair={pair}: eager median {np.median(ts):7.1f} us  min {np.min(ts):7.1f} us"
                  f"   replayed {tg:7.1f} us", flush=True)
eng.set_tuning("chain_pair", 1)
