#!/usr/bin/env python3
"""Diagnostic: per-step timeline of the diagonal-tile workgroup of k_potrf_step8 (s_memrealtime stamps, 100 MHz)."""
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
eng.set_tuning("timeline", 1)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    eng.potrf(S, out=R, flag=f)
for _ in range(5): gr.replay()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (4 * 4096))()
eng.lib.gsmvi_debug_read_stamps(eng._ctx, buf, 4 * 4096)
st = np.array(buf, dtype=np.uint64).reshape(4, 4096)[3][:8 * (D // 64)].reshape(-1, 8).astype(np.int64)
t0 = st[0, 0]
print("step: start  staged  X_I  chol_start  chol_end  stored   (us since step 0 start); next-start gap")
for k in range(st.shape[0]):
    r = (st[k, :6] - t0) / 100.0
    nxt = (st[k + 1, 0] - st[k, 5]) / 100.0 if k + 1 < st.shape[0] else float("nan")
    print(f"{k:3d}: " + " ".join(f"{v:8.2f}" for v in r) + f"   | step {((st[k+1,0] if k+1<st.shape[0] else st[k,5]) - st[k,0]) / 100.0:6.2f}  gap {nxt:5.2f}")
d = np.diff(st[1:-1, :6], axis=1) / 100.0
print("median phase (staged, X_I, ->chol_start, chol, store):", np.round(np.median(d, axis=0), 2))
