import sys
import numpy as np
sys.path.insert(0, ".")
import gsmvi_amd
from oracle import gsm_oracle as orc
eng = gsmvi_amd.get_engine()
for B, scale, reg in ((16, 1e5, 1e3), (16, 1e4, 1e3), (16, 1e5, 1e2), (4, 1e3, 1e3), (32, 30.0, 100.0)):
    st = orc.make_update_state(160, B, seed=B)
    Gs = st["vs"] * scale
    X, G, mu0, S0 = (eng.asarray(a) for a in (st["samples"], Gs, st["mu0"], st["S0"]))
    for full in (0, 1):
        eng.set_tuning("bam_full", full)
        mu, S, f = eng.bam_update(X, G, mu0, S0, reg, 0.0)
        print(B, scale, reg, "full" if full else "fused", "flag", eng.read_flag(f), "finite", bool(np.isfinite(S.cpu().numpy()).all()))
    eng.set_tuning("bam_full", 0)
