import os as _os; _os.environ.setdefault("GSMVI_HIP_DEBUG_LIB", "1")   # gsmvi_debug_* are exported by libgsmvi_hip_debug.so only
import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch, gsmvi_amd
from oracle import gsm_oracle as orc
eng = gsmvi_amd.get_engine()
D, B = int(sys.argv[1]), int(sys.argv[2])
st = orc.make_update_state(D, B, 1)
X, G, mu0, S0 = (eng.asarray(st[k]) for k in ("samples", "vs", "mu0", "S0"))
mu, S, f = eng.bam_update(X, G, mu0, S0, 1.0, 1e-6)
n = B + 1
npk = n * (n + 1) // 2
off_scr = 5 * n * n + 3 * n + npk + 2
LL = 144 * 144
def rd(off, cnt):
    buf = (C.c_double * cnt)()
    eng.lib.gsmvi_debug_read_workspace(eng._ctx, 2, off, buf, cnt)
    return np.array(buf)
coef = rd(off_scr + 5 * LL, 48)
print("flag", eng.read_flag(f), "kstar", coef[40], "s", coef[41], "bad", coef[42], "c2", coef[:int(coef[40])])
Nd = rd(3 * n * n + 3 * n, n * n).reshape(n, n)
BB = rd(off_scr + 5 * LL + 64, n * n).reshape(n, n)
w, V = np.linalg.eigh(Nd + 0.25 * np.eye(n))
print("eig(A) min/max", w.min(), w.max())
BBref = Nd + 0.5 * np.eye(n) + (V * np.sqrt(w)) @ V.T
print("BB err", np.abs(BB - BBref).max() / np.abs(BBref).max(), "nan in BB", np.isnan(BB).any())
ks = int(coef[40])
Y = rd(off_scr + (2 * LL if ks & 1 else 0), LL).reshape(144, 144)[:n, :n]
print("Y err", np.abs(Y * np.sqrt(coef[41]) - (V * np.sqrt(w)) @ V.T).max())
print("mu nan", torch.isnan(mu).any().item(), "S nan", torch.isnan(S).any().item())
Ld = rd(2 * n * n, n * n).reshape(n, n)
print("L nan", np.isnan(Ld).any(), "L err", np.abs(Ld @ Ld.T - BBref).max())


