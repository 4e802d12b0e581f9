"""Blocked Cholesky (gsmvi_potrf_f64): eager vs. replayed hipGraph.  (A two-stream look-ahead schedule was measured
with this script and rejected: 745 / 686 us eager / graph against 628 / 630 us at D=1024 -- DESIGN.md section 8.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gsmvi_amd
eng = gsmvi_amd.get_engine()
for D in (256, 1024, 4096):
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    A = torch.randn(D, D, dtype=torch.float64, device="cuda", generator=g)
    S = (A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device="cuda")).contiguous()
    R = eng.empty(D, D); f = eng.new_flag()
    for nola in (1,):
        n = 50 if D <= 1024 else 10
        for _ in range(3): eng.potrf(S, out=R, flag=f)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): eng.potrf(S, out=R, flag=f)
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n * 1e6
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            eng.potrf(S, out=R, flag=f)
        for _ in range(3): gr.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): gr.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / n * 1e6
        print(f"D={D} {'single-stream' if nola else 'look-ahead  '}: eager {te:.0f} us  graph {tg:.0f} us  flag {eng.read_flag(f)}")
