#!/usr/bin/env python3
"""Diagnostic: phase times of k_bam_small48 (BaM's small chain, n <= 48, one workgroup) from s_memrealtime stamps."""
import os as _os; _os.environ.setdefault("GSMVI_HIP_DEBUG_LIB", "1")   # gsmvi_debug_* are exported by libgsmvi_hip_debug.so only
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
import _inputs as orc
D, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 32)
reg = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
eng = gsmvi_amd.get_engine()
m, cov, P = orc.make_gaussian_target(D, 1)
tgt = gsmvi_amd.GaussianTarget(m, precision=P)
mu0 = eng.asarray(m)
F0, _ = eng.potrf(eng.asarray(cov))
Z = eng.normal(B, D, 5, 0)
X = eng.sample(Z, mu0, F0)
G = tgt.lp_g(X)
eng.set_tuning("cov_dbg", 256)
names = ["slab sum", "N, trace, Y0", "Newton-Schulz", "BB", "chol64_blk", "outputs"]
for trial in range(3):
    for _ in range(10):
        eng.bam_factor_update(Z, X, G, mu0, F0, reg)
    torch.cuda.synchronize()
    buf = (C.c_double * 16)()
    R = 2 * eng._max_B + 8
    ldb = max(R // 2 + 16, 144)
    n_small = 8 * R + 7 * R * R + 4096 + 5 * ldb * ldb + 64 + ldb * ldb + 64 + 8 * R * R + 16
    eng.lib.gsmvi_debug_read_workspace(eng._ctx, 2, n_small - 16, buf, 16)
    st = np.frombuffer(buf, dtype=np.int64)
    print("  ".join(f"{n} {d / 100.0:.2f}us" for n, d in zip(names, np.diff(st[:7]))), f" total {(st[6] - st[0]) / 100.0:.2f}us, k* = {st[7]}, SIMD of waves 0..7: {list(st[8:16])}")
