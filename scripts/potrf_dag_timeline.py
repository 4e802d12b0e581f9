#!/usr/bin/env python3
"""Diagnostic: where an iteration of k_potrf_dag's CHAIN workgroup spends its time (s_memrealtime stamps, 100 MHz; knob "timeline" = 2).
Phases per iteration c: wait for the two tiles' flags | sc1 loads + staging | solve and update products, E staged | drain + publish of
the solved block | chol64_blk ([T | I] -> [R | W]) | W copy, write-through stores, drain, publish.  The factor's own block and the mirror block
are stored behind the NEXT iteration's loads (phase 2).
usage: potrf_dag_timeline.py [D]"""
import os as _os; _os.environ.setdefault("GSMVI_HIP_DEBUG_LIB", "1")   # gsmvi_debug_* are exported by libgsmvi_hip_debug.so only
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
eng = gsmvi_amd.get_engine()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator(device="cuda"); g.manual_seed(0)
A = torch.randn(D, D, dtype=torch.float64, device="cuda", generator=g)
S = (A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device="cuda")).contiguous()
R = eng.empty(D, D); f = eng.new_flag()
eng.potrf(S, out=R, flag=f)
eng.set_tuning("timeline", 2)
for _ in range(5):
    eng.potrf(S, out=R, flag=f)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (4 * 4096))()
eng.lib.gsmvi_debug_read_stamps(eng._ctx, buf, 4 * 4096)
n = min(64, (D + 63) // 64)
st16 = np.array(buf, dtype=np.uint64).reshape(4, 4096)[3][:16 * n].reshape(-1, 16).astype(np.int64)
st = st16[:, :8]
names = ["wait flags", "sc1 loads, own stores, W publish, staging", "products + E", "drain + publish X", "chol64_blk", "W copy + stores (published next iteration)", "(end)"]
d = np.diff(st, axis=1) / 100.0
print(f"k_potrf_dag chain, D = {D}: {n} iterations, {(st[-1, 7] - st[0, 0]) / 100.0:.1f} us from the first stamp to the last "
      f"({(st[-1, 7] - st[0, 0]) / 100.0 / n:.2f} us per iteration)")
print("median us per phase over iterations 2 .. n-2:")
for k, nm in enumerate(names):
    print(f"  {nm:28s} {np.median(d[2:-1, k]):6.2f}   (max {d[2:-1, k].max():6.2f})")
print(f"  {'sum of medians':28s} {np.median(d[2:-1], axis=0).sum():6.2f}")
sub = np.stack([st16[:, 8] - st16[:, 2], st16[:, 9] - st16[:, 8], st16[:, 10] - st16[:, 9], st16[:, 3] - st16[:, 10]], axis=1) / 100.0
print("inside 'products + E' (thread 0's wave): " + ", ".join(f"{nm} {np.median(sub[2:-1, k]):.2f}" for k, nm in
      enumerate(["solve product", "X^T to LDS + X stores + barrier", "update product", "barrier + E staged"])))
