#!/usr/bin/env python3
"""Diagnostic: timeline of ONE dense GSM update (all kernels) inside a replayed hipGraph, from in-kernel s_memrealtime stamps
(100 MHz, one clock for the whole chip).  usage: timeline2.py [cold|warm] [name=value tuning ...]"""
import os as _os; _os.environ.setdefault("GSMVI_HIP_DEBUG_LIB", "1")   # gsmvi_debug_* are exported by libgsmvi_hip_debug.so only
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
from bench import make_instances
mode = sys.argv[1] if len(sys.argv) > 1 else "cold"
eng = gsmvi_amd.get_engine()
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    eng.set_tuning(k, int(v))
D, B = 1024, 32
n_inst = 21 if mode == "cold" else 1
inst, m, P = make_instances(eng, D, B, n_inst)
eng.gsm_update(inst[0]["X"], inst[0]["G"], inst[0]["mu0"], inst[0]["S0"], out=(inst[0]["mu"], inst[0]["S"]))
eng.set_tuning("timeline", 1)
def step(k):
    it = inst[k % n_inst]
    eng.gsm_update(it["X"], it["G"], it["mu0"], it["S0"], out=(it["mu"], it["S"]))
for k in range(21): step(k)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for k in range(21): step(k)
NW = 4 * 4096
names = {0: "panel", 1: "scalars", 2: "cov", 3: "k3"}
for trial in range(4):
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * NW)()
    eng.lib.gsmvi_debug_read_stamps(eng._ctx, buf, NW)
    st = np.array(buf, dtype=np.uint64).reshape(4, 512, 8).astype(np.int64)
    t0 = st[st > 0].min()
    print(f"--- trial {trial} ({mode}) ---")
    prev_end = None
    for k in range(4):
        a = st[k]
        live = a[:, 0] > 0
        if not live.any():
            continue
        a = a[live]
        us = np.where(a > 0, (a - t0) / 100.0, np.nan)
        cols = [c for c in range(8) if np.isfinite(us[:, c]).any()]
        s = f"{names[k]:8s} {live.sum():3d} WGs: "
        for c in cols:
            s += f"[{c}] {np.nanmin(us[:, c]):6.2f}/{np.nanmedian(us[:, c]):6.2f}/{np.nanmax(us[:, c]):6.2f}  "
        print(s)
        first, last = np.nanmin(us[:, cols[0]]), np.nanmax(us[:, cols[-1]])
        if prev_end is not None:
            print(f"         gap from previous kernel's last stamp to this kernel's first start: {first - prev_end:.2f} us")
        print(f"         span {last - first:.2f} us")
        prev_end = last

# ---- who is in the tail of the covariance kernel (last trial)? ----
a = st[2]
live = a[:, 0] > 0
us = (a[live] - t0) / 100.0
end = us[:, 5]
order = np.argsort(-end)
print("cov kernel: slowest workgroups (blockIdx, start, loads, staged, mfma, mirror, drained):")
for i in order[:14]:
    print("   ", int(i), " ".join(f"{v:6.2f}" for v in us[i, :6]))
print("cov kernel: fastest:")
for i in order[-4:]:
    print("   ", int(i), " ".join(f"{v:6.2f}" for v in us[i, :6]))
ph = np.diff(us[:, :6], axis=1)
print("phase medians (loads, stage, mfma, mirror, drain):", np.round(np.median(ph, axis=0), 2), " p95:", np.round(np.percentile(ph, 95, axis=0), 2))
print("blocks >= 256 (single-tile): end median", np.median(end[256:]) if len(end) > 256 else None, " blocks < 256: end median", np.median(end[:256]), "p95", np.percentile(end[:256], 95))
a = st[0]; live = a[:, 0] > 0; usp = (a[live] - t0) / 100.0
php = np.diff(usp[:, :4], axis=1)
print("panel phase medians (staged, mfma, end):", np.round(np.median(php, axis=0), 2), " p95:", np.round(np.percentile(php, 95, axis=0), 2))
