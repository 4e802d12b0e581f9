"""Diagnostic: intermediates of the n = 128 factor chain on linearly dependent rows."""
import os as _os; _os.environ.setdefault("GSMVI_HIP_DEBUG_LIB", "1")   # gsmvi_debug_* are exported by libgsmvi_hip_debug.so only
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, gsmvi_amd
eng = gsmvi_amd.get_engine()
D, B = 300, 64
n = 2 * B
rs = np.random.RandomState(D)
mu0, F0 = np.zeros(D), np.eye(D)
Z = rs.standard_normal((B, D)); X = mu0 + Z @ F0; G = -2.0 * (X - 0.5)
mu, F, flag = eng.gsm_factor_update(eng.asarray(Z), eng.asarray(X), eng.asarray(G), eng.asarray(mu0), eng.asarray(F0))
print("flag", eng.read_flag(flag))
def rd(k):
    buf = (C.c_double * (n * n))()
    eng.lib.gsmvi_debug_read_workspace(eng._ctx, 2, k * n * n, buf, n * n)
    return np.array(buf).reshape(n, n)
K, Rg, W, T, P = rd(0), rd(1), rd(2), rd(3), rd(4)
print("Rg diag zeros:", np.sum(np.diag(Rg) == 0), "min pos diag", np.diag(Rg)[np.diag(Rg) > 0].min(), "max |Rg|", np.abs(Rg).max())
print("T diag min", np.diag(T).min(), "max|T|", np.abs(T).max(), "nan", np.isnan(T).any())
print("max|W|", np.abs(W).max(), "max|P|", np.abs(P).max(), "max|K|", np.abs(K).max())
J = np.block([[np.zeros((B, B)), np.eye(B)], [np.eye(B), -np.eye(B)]]) / B
Ap = np.eye(n) + Rg @ J @ Rg.T
print("T^T T vs A':", np.abs(T.T @ T - Ap).max(), " eig min A'", np.linalg.eigvalsh(0.5 * (Ap + Ap.T)).min())
Rgt = Rg.copy(); dz = np.diag(Rg) == 0; Rgt[dz, dz] = 1.0
W_ref = np.linalg.inv(Rgt).T
print("W vs ref:", np.abs(W - W_ref).max() / np.abs(W_ref).max())
K_ref = W_ref.T @ (T - np.eye(n)) @ W_ref
print("K vs ref:", np.abs(K - K_ref).max() / max(np.abs(K_ref).max(), 1e-300), "max|K_ref|", np.abs(K_ref).max())
