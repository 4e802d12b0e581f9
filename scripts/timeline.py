#!/usr/bin/env python3
"""Diagnostic: per-phase timeline of k_gsm_cov_sym from in-kernel s_memrealtime stamps (100 MHz)."""
import os as _os; _os.environ.setdefault("GSMVI_HIP_DEBUG_LIB", "1")   # gsmvi_debug_* are exported by libgsmvi_hip_debug.so only
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
from bench import make_instances
eng = gsmvi_amd.get_engine()
D, B = 1024, 32
inst, m, P = make_instances(eng, D, B, 2)
it = inst[0]
rec = eng.gsm_local_stage(it["X"], it["G"], it["mu0"], it["S0"])
eng.set_tuning("cov_dbg", 16)
nwg = sum((D // 32 - ti + 1) // 2 for ti in range(D // 32))
for trial in range(3):
    torch.cuda.synchronize()
    for _ in range(300):                       # steady state: clocks up, stamps of the LAST launch survive
        eng.gsm_apply(rec, it["mu0"], it["S0"], out=(it["mu"], it["S"]))
    buf = (C.c_ulonglong * (nwg * 8))()
    eng.lib.gsmvi_debug_read_stamps(eng._ctx, buf, nwg * 8)
    st = np.array(buf, dtype=np.uint64).reshape(nwg, 8)[:, :6].astype(np.int64)
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0
    names = ["start", "loads_done", "lds_staged", "mfma_done", "stores_issued", "stores_drained"]
    print(f"trial {trial}: kernel span {us[:, 5].max():.2f} us over {nwg} WGs")
    for k, n in enumerate(names):
        print(f"  {n:15s} min {us[:, k].min():6.2f}  median {np.median(us[:, k]):6.2f}  max {us[:, k].max():6.2f}")
    d = np.diff(us, axis=1)
    print("  per-WG phase durations (median):", {names[k + 1]: round(float(np.median(d[:, k])), 2) for k in range(5)})

# which workgroups are slow?
nt = D // 32
rows = []
for ti in range(nt):
    rows += [ti] * ((nt - ti + 1) // 2)
rows = np.array(rows)
order = np.argsort(-us[:, 1])
print("slowest loads_done:", [(int(i), int(rows[i]), round(float(us[i, 0]), 2), round(float(us[i, 1]), 2)) for i in order[:12]])
print("fastest loads_done:", [(int(i), int(rows[i]), round(float(us[i, 0]), 2), round(float(us[i, 1]), 2)) for i in order[-6:]])
ld = us[:, 1] - us[:, 0]
print("corr(load duration, start) =", round(float(np.corrcoef(ld, us[:, 0])[0, 1]), 2), " corr(load duration, blockIdx) =", round(float(np.corrcoef(ld, np.arange(nwg))[0, 1]), 2))
for lo in range(0, nwg, 34):
    print(f"  blocks {lo:3d}-{min(lo+33,nwg-1):3d}: start {us[lo:lo+34,0].mean():.2f}  load {ld[lo:lo+34].mean():.2f}  end {us[lo:lo+34,5].mean():.2f}")
