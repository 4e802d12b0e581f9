#!/usr/bin/env python3
"""Diagnostic: phase times of k_gsmf_small (the 2B x 2B chain of the factor update) from s_memrealtime stamps."""
import os as _os; _os.environ.setdefault("GSMVI_HIP_DEBUG_LIB", "1")   # gsmvi_debug_* are exported by libgsmvi_hip_debug.so only
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
import _inputs as orc
D, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 32)
eng = gsmvi_amd.get_engine()
g = torch.Generator(device="cuda"); g.manual_seed(0)
kw = dict(dtype=torch.float64, device="cuda", generator=g)
A = torch.randn(D, D, **kw); S0 = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device="cuda")
F0 = torch.linalg.cholesky(S0).T.contiguous(); mu0 = torch.randn(D, **kw); Z = torch.randn(B, D, **kw)
X = (mu0 + Z @ F0).contiguous(); G = -(X - 0.5)
eng.set_tuning("cov_dbg", 128)
names = ["load", "chol([Gamma|I])->Rg,W", "A'", "chol(A')", "P=(T-I)W", "K=W^T P"]
for trial in range(3):
    for _ in range(20):
        eng.gsm_factor_update(Z, X, G, mu0, F0)
    buf = (C.c_double * 8)()                      # the stamps are the last 16 words of the small-matrix workspace (ws_sizes)
    R = 2 * eng._max_B + 8
    ldb = max(R // 2 + 16, 144)
    n_small = 8 * R + 7 * R * R + 4096 + 5 * ldb * ldb + 64 + ldb * ldb + 64 + 8 * R * R + 16
    eng.lib.gsmvi_debug_read_workspace(eng._ctx, 2, n_small - 16, buf, 8)
    st = np.frombuffer(buf, dtype=np.int64)[:7]
    print("  ".join(f"{n} {d / 100.0:.2f}us" for n, d in zip(names, np.diff(st))), f" total {(st[6] - st[0]) / 100.0:.2f}us")
