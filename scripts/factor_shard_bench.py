"""Per-rank cost of the batch-sharded factor update (what ONE of P ranks executes, without the all-gather),
against the fused single-GPU update.  usage: factor_shard_bench.py D B"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gsmvi_amd
D, B = int(sys.argv[1]), int(sys.argv[2])
eng = gsmvi_amd.get_engine()
g = torch.Generator(device="cuda"); g.manual_seed(0)
kw = dict(dtype=torch.float64, device="cuda", generator=g)
A = torch.randn(D, D, **kw); S0 = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device="cuda")
F0 = torch.linalg.cholesky(S0).T.contiguous(); mu0 = torch.randn(D, **kw); Z = torch.randn(B, D, **kw)
X = (mu0 + Z @ F0).contiguous(); G = (-(X - 0.5)).contiguous()


def timed(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


out = (eng.empty(D), eng.empty(D, D)); flag = eng.new_flag()
print(f"D={D} B={B} fused factor update: {timed(lambda: eng.gsm_factor_update(Z, X, G, mu0, F0, out=out, flag=flag)):.1f} us")
rec = eng.gsm_factor_local_stage(Z, X, G, mu0, F0)
t_apply = timed(lambda: eng.gsm_factor_apply(Z, rec, mu0, F0, out=out, flag=flag))
for P in (1, 2, 4, 8):
    per = B // P
    r = eng.empty(per, eng.record_len(D))
    t_loc = timed(lambda: eng.gsm_factor_local_stage(Z[:per], X[:per], G[:per], mu0, F0, out=r))
    print(f"  P={P}: local stage ({per} samples) {t_loc:.1f}  apply {t_apply:.1f}  sum {t_loc + t_apply:.1f} us  "
          f"(+ all-gather of {per * 3 * D * 8 / 1024:.0f} KiB per rank)")
