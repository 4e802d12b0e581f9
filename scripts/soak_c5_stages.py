#!/usr/bin/env python3
"""The soak that reproduces the rare second outcome at D=4096, B=64 (synchronise, clone, compare with the first result, as
scripts/soak_round3.py does), with the intermediates of the context workspace compared against the first call's snapshot
ONLY when the result deviates (they survive until the next call): says which stage took the other outcome.
usage: soak_c5_stages.py [seconds] [knob=value ...]"""
import os as _os; _os.environ.setdefault("GSMVI_HIP_DEBUG_LIB", "1")   # gsmvi_debug_* are exported by libgsmvi_hip_debug.so only
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
args = [a for a in sys.argv[1:] if "=" not in a]
budget = float(args[0]) if args else 60.0
D, B = 4096, 64
eng = gsmvi_amd.get_engine()
n = 2 * B
rs = np.random.RandomState(D + B)
F0 = eng.asarray(rs.standard_normal((D, D)) / np.sqrt(D) + 0.6 * np.eye(D))
mu0 = eng.asarray(rs.standard_normal(D))
Z = eng.asarray(rs.standard_normal((B, D)))
X = eng.sample(Z, mu0, F0)
G = -(X - 0.3)
mu, F, flag = eng.empty(D), eng.empty(D, D), eng.new_flag()
eng.gsm_factor_update(Z, X, G, mu0, F0, out=(mu, F), flag=flag)
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        eng.set_tuning(k, int(v))
eng.gsm_factor_update(Z, X, G, mu0, F0, out=(mu, F), flag=flag)
torch.cuda.synchronize()
R = 2 * eng._max_B + 8
ldb = max(R // 2 + 16, 144)
n_small = 8 * R + 7 * R * R + 4096 + 5 * ldb * ldb + 64 + ldb * ldb + 64 + 8 * R * R + 16

class View:
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
views = [torch.as_tensor(View(base[r] + 8 * off, cnt), device="cuda") for _, r, off, cnt in stages]
snap = [v.clone() for v in views]
ref = (mu.clone(), F.clone())
t0, calls, bad = time.perf_counter(), 0, 0
while time.perf_counter() - t0 < budget:
    eng.gsm_factor_update(Z, X, G, mu0, F0, out=(mu, F), flag=flag)
    torch.cuda.synchronize()
    h = (mu.clone(), F.clone(), int(flag.item()))
    if not (h[2] == 0 and torch.equal(h[0], ref[0]) and torch.equal(h[1], ref[1])):
        bad += 1
        rep = []
        for (nm, *_), v, r in zip(stages, views, snap):
            d = (v - r).abs()
            if bool((d > 0).any()):
                rep.append((nm, int((d > 0).sum()), float(d.max())))
        dF = (h[1] - ref[1]).abs()
        print("MISMATCH at call", calls, "flag", h[2], "F:", int((dF > 0).sum()), float(dF.max()), "stages that differ:", rep, flush=True)
    calls += 1
print(f"{calls} calls in {time.perf_counter() - t0:.0f} s, {bad} deviating")
