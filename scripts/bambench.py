import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
from oracle import gsm_oracle as orc
eng = gsmvi_amd.get_engine()
for D, B in ((1024, 32), (1024, 128)):
    st = orc.make_update_state(D, B, 1)
    X, G, mu0, S0 = (eng.asarray(st[k]) for k in ("samples", "vs", "mu0", "S0"))
    out = (eng.empty(D), eng.empty(D, D)); flag = eng.new_flag()
    for _ in range(3): eng.bam_update(X, G, mu0, S0, 1.0, 1e-6, out=out, flag=flag)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): eng.bam_update(X, G, mu0, S0, 1.0, 1e-6, out=out, flag=flag)
    torch.cuda.synchronize(); print(f"BaM update D={D} B={B}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
