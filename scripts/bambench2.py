import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
from oracle import gsm_oracle as orc
eng = gsmvi_amd.get_engine()
D, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 128)
st = orc.make_update_state(D, B, 1)
X, G, mu0, S0 = (eng.asarray(st[k]) for k in ("samples", "vs", "mu0", "S0"))
out = (eng.empty(D), eng.empty(D, D)); flag = eng.new_flag()
for _ in range(3): eng.bam_update(X, G, mu0, S0, 1.0, 1e-6, out=out, flag=flag)
ts = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.bam_update(X, G, mu0, S0, 1.0, 1e-6, out=out, flag=flag)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.2f}" for t in ts))
