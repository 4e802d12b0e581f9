import os as _os; _os.environ.setdefault("GSMVI_HIP_DEBUG_LIB", "1")   # gsmvi_debug_* are exported by libgsmvi_hip_debug.so only
import ctypes as C, sys, numpy as np
sys.path.insert(0, '.')
import gsmvi_amd
from oracle import gsm_oracle as orc
eng = gsmvi_amd.get_engine()
D, B = 8, 2
st = orc.make_update_state(D, B, D + B)
Fm = st["L"].T.copy(); Z = st["Z"]; X = st["samples"]; G = st["vs"]; mu0 = st["mu0"]; n = 2 * B
mu, F, flag = eng.gsm_factor_update(eng.asarray(Z), eng.asarray(X), eng.asarray(G), eng.asarray(mu0), eng.asarray(Fm))
def rd(region, off, cnt):
    buf = (C.c_double * cnt)()
    eng.lib.gsmvi_debug_read_workspace(eng._ctx, region, off, buf, cnt)
    return np.array(buf)
Rt = rd(1, 0, n * D).reshape(n, D); Tm = rd(1, n * D, n * D).reshape(n, D); Fs = rd(1, 2 * n * D, n * D).reshape(n, D)
Gam = rd(2, 0, n * n).reshape(n, n); Rg = rd(2, n * n, n * n).reshape(n, n); Ap = rd(2, 2 * n * n, n * n).reshape(n, n); T = rd(2, 3 * n * n, n * n).reshape(n, n)
W = G @ Fm.T
ww = (W * W).sum(1); zw = (Z * W).sum(1); rho = 0.5 * np.sqrt(1 + 4 * (ww + zw ** 2)) - 0.5; den = 1 + rho - zw
U = ((W + Z) + Z * ((ww + zw) / den)[:, None]) / (1 + rho)[:, None]
Rt_o = np.vstack([Z, U]); Tm_o = np.vstack([X - mu0, U @ Fm])
print("Z err", abs(Rt[:B] - Z).max(), "U err", abs(Rt[B:] - U).max())
print("Tm top err", abs(Tm[:B] - Tm_o[:B]).max(), "Tm bot err", abs(Tm[B:] - Tm_o[B:]).max())
Gam_o = Rt_o @ Rt_o.T
print("Gam err", abs(Gam - Gam_o).max(), "Rg err", abs(Rg - np.linalg.cholesky(Gam_o).T).max())
print("flag", eng.read_flag(flag))
print("U dev", Rt[B:][:, :4], "\nU ref", U[:, :4])
