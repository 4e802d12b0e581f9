"""Per-rank cost of the row-block sharded update (what ONE of P ranks executes, without the all-gather),
against the fused single-GPU update.  usage: rowsbench.py D B"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gsmvi_amd
from gsmvi_amd.dist import row_bounds
from bench import make_instances
D, B = int(sys.argv[1]), int(sys.argv[2])
eng = gsmvi_amd.get_engine()
inst, m, P = make_instances(eng, D, B, 1)
it = inst[0]
X, G, mu0, S0 = it["X"], it["G"], it["mu0"], it["S0"]


def timed(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


mu, S = eng.empty(D), eng.empty(D, D)
print(f"D={D} B={B} fused update: {timed(lambda: eng.gsm_update(X, G, mu0, S0, out=(mu, S))):.1f} us")
SG = G @ S0
rec = eng.gsm_records(X, G, mu0, SG)
for Pn in (1, 2, 4, 8):
    lo, hi = row_bounds(D, Pn, 0)
    S0r = S0[lo:hi].contiguous(); Sr = eng.empty(hi - lo, D); SGc = eng.empty(B, hi - lo)
    t1 = timed(lambda: eng.gsm_rows_stage(G, S0r, out=SGc))
    t2 = timed(lambda: eng.gsm_records(X, G, mu0, SG, out=rec))
    t3 = timed(lambda: eng.gsm_apply_rows(rec, mu0, S0r, lo, out=(mu, Sr)))
    print(f"  P={Pn}: rows_stage {t1:.1f}  records {t2:.1f}  apply_rows {t3:.1f}  sum {t1 + t2 + t3:.1f} us")
