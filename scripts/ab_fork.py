"""A/B in one process: factor update / fit with U F forked onto the context's second stream (fork=1) or not (fork=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
from oracle import gsm_oracle as orc
eng = gsmvi_amd.get_engine()
for D, B in ((1024, 32), (256, 8), (4096, 64)):
    st = orc.make_update_state(D, B, 1) if D <= 1024 else None
    m, cov_t, P = orc.make_gaussian_target(D, 0)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    g0 = torch.Generator(device="cuda"); g0.manual_seed(0)
    A = torch.randn(D, D, dtype=torch.float64, device="cuda", generator=g0)
    F0 = torch.linalg.cholesky(A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device="cuda")).T.contiguous()
    mu0 = torch.randn(D, dtype=torch.float64, device="cuda", generator=g0)
    Z = torch.randn(B, D, dtype=torch.float64, device="cuda", generator=g0)
    X = (mu0 + Z @ F0).contiguous(); G = tgt.lp_g(X)
    out = (eng.empty(D), eng.empty(D, D)); fl = eng.new_flag()
    res = {}
    for rep in range(2):
        for fk in (0, 1):
            eng.set_tuning("fork", fk)
            eng.gsm_factor_update(Z, X, G, mu0, F0, out=out, flag=fl)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(10):
                    eng.gsm_factor_update(Z, X, G, mu0, F0, out=out, flag=fl)
            for _ in range(3): gr.replay()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): gr.replay()
            torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 200 * 1e6
            n = 300 if D <= 1024 else 60
            gs = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
            gs.fit(1, niter=10, batch_size=B, verbose=False, method="factor")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gs.fit(1, niter=n - 1, batch_size=B, verbose=False, method="factor")
            torch.cuda.synchronize(); fr = n / (time.perf_counter() - t0)
            res.setdefault(fk, []).append((tg, fr))
    for fk in (0, 1):
        print(f"D={D} B={B} fork={fk}: factor update in a graph {min(r[0] for r in res[fk]):.1f} us, fit {max(r[1] for r in res[fk]):.0f} it/s")
eng.set_tuning("fork", 1)
