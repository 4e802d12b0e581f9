"""Fit rate and update time of the factor-form BaM against the dense form (D, B from argv; default 1024 32)."""
import sys
import time
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
import gsmvi_amd
from oracle import gsm_oracle as orc

D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
eng = gsmvi_amd.get_engine()
m, cov, P = orc.make_gaussian_target(D, 1)
tgt = gsmvi_amd.GaussianTarget(m, precision=P)
sched = lambda i: 100.0 / (1 + i)
for method, n in (("dense", 100), ("factor", 600)):
    bam = gsmvi_amd.BaM(D, tgt.lp, tgt.lp_g)
    bam.fit(1, sched, niter=3, batch_size=B, verbose=False, rng="device", method=method)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bam.fit(1, sched, niter=n - 1, batch_size=B, verbose=False, rng="device", method=method)
    torch.cuda.synchronize()
    print(f"BaM fit D={D} B={B} {method}: {n / (time.perf_counter() - t0):.0f} it/s, reverts {bam.n_reverts}")
mu0 = eng.asarray(m)
F0, _ = eng.potrf(eng.asarray(cov))
Z = eng.normal(B, D, 5, 0)
X = eng.sample(Z, mu0, F0)
G = tgt.lp_g(X)
mu, Fo, flag = eng.empty(D), eng.empty(D, D), eng.new_flag()
f = lambda: eng.bam_factor_update(Z, X, G, mu0, F0, 1.0, out=(mu, Fo), flag=flag)
for _ in range(5):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    f()
e1.record()
torch.cuda.synchronize()
print(f"factor-form BaM update eager: {e0.elapsed_time(e1) * 10:.1f} us")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(4):
        f()
g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
print(f"factor-form BaM update graph: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
