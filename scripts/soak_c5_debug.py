#!/usr/bin/env python3
"""Diagnostic soak of the factor update at D=4096, B=64 (or argv D B): every call's intermediates (W slabs / V Fm slabs, Rt, Tm,
Fs, Gram slabs, Rg, T, W, P, K'', coefficients) are compared ON THE DEVICE with the first call's (in-place views of the context
workspace through gsmvi_debug_workspace_ptr); on a mismatch every differing stage is reported in pipeline order.
usage: soak_c5_debug.py [seconds] [D B] [knob=value ...]"""
import os as _os; _os.environ.setdefault("GSMVI_HIP_DEBUG_LIB", "1")   # gsmvi_debug_* are exported by libgsmvi_hip_debug.so only
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
args = [a for a in sys.argv[1:] if "=" not in a]
budget = float(args[0]) if args else 60.0
D, B = (int(args[1]), int(args[2])) if len(args) > 2 else (4096, 64)
eng = gsmvi_amd.get_engine()
n = 2 * B
rs = np.random.RandomState(D + B)
F0 = eng.asarray(rs.standard_normal((D, D)) / np.sqrt(D) + 0.6 * np.eye(D))
mu0 = eng.asarray(rs.standard_normal(D))
Z = eng.asarray(rs.standard_normal((B, D)))
X = eng.sample(Z, mu0, F0)
G = -(X - 0.3)
mu, F, flag = eng.empty(D), eng.empty(D, D), eng.new_flag()
eng.gsm_factor_update(Z, X, G, mu0, F0, out=(mu, F), flag=flag)      # (sizes the context)
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        eng.set_tuning(k, int(v))
eng.gsm_factor_update(Z, X, G, mu0, F0, out=(mu, F), flag=flag)      # the reference call, knobs applied
R = 2 * eng._max_B + 8
ldb = max(R // 2 + 16, 144)
n_small = 8 * R + 7 * R * R + 4096 + 5 * ldb * ldb + 64 + ldb * ldb + 64 + 8 * R * R + 16

class View:                                   # a float64 device array at a raw address, through __cuda_array_interface__
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

def region_ptr(region):
    p = C.c_void_p()
    assert eng.lib.gsmvi_debug_workspace_ptr(eng._ctx, region, C.byref(p)) == 0
    return p.value

base = [region_ptr(r) for r in range(3)]
stages = [("Rt", 1, 0, n * D), ("Tm top (X - mu)", 1, n * D, B * D), ("gram_slabs", 2, n_small - (8 * R * R + 16), 8 * n * n),
          ("coef", 2, 5 * n * n, 2 * n), ("Rg", 2, n * n, n * n), ("W(Pm)", 2, 4 * n * n, n * n), ("T", 2, 3 * n * n, n * n),
          ("P(Ap)", 2, 2 * n * n, n * n), ("K''(Gam)", 2, 0, n * n), ("pp (V Fm slabs)", 0, 0, 8 * B * D),
          ("Tm bottom (V Fm)", 1, n * D + B * D, B * D), ("Fs", 1, 2 * n * D, n * D)]
views = [torch.as_tensor(View(base[r] + 8 * off, cnt), device="cuda") for _, r, off, cnt in stages] + [F.view(-1), mu]
names = [s[0] for s in stages] + ["F", "mu"]
torch.cuda.synchronize()
ref = [v.clone() for v in views]
t0, calls, bad = time.perf_counter(), 0, 0
while time.perf_counter() - t0 < budget:
    eng.gsm_factor_update(Z, X, G, mu0, F0, out=(mu, F), flag=flag)
    if os.environ.get("SOAK_CLONE"):             # the allocation / idle pattern of scripts/soak_round3.py
        torch.cuda.synchronize()
        keep = (mu.clone(), F.clone())
    same = torch.stack([torch.equal(v, r) if False else (v == r).all() for v, r in zip(views, ref)])   # one sync for all
    ok = bool(same.all().item())
    if not ok:
        bad += 1
        rep = []
        for nm, v, r, s_ in zip(names, views, ref, same.tolist()):
            if not s_:
                d = (v - r).abs()
                rep.append((nm, int((d > 0).sum()), float(d.max())))
        if bad <= 20:
            print("MISMATCH at call", calls, rep, flush=True)
    calls += 1
print(f"{calls} calls in {time.perf_counter() - t0:.0f} s, {bad} mismatching")
