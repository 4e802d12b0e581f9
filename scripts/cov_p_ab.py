#!/usr/bin/env python3
"""A/B of the persistent covariance kernel (k_gsm_cov_sym_p) against the one-item-per-workgroup kernel (knob cov_dbg = 512)
at large D, dispatch-event timing on three HBM-cold instances; also checks the two are bit-identical."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsmvi_amd
from bench import make_instances
eng = gsmvi_amd.get_engine()
for D, B in ((4096, 32), (4096, 64), (4096, 16), (3072, 32), (2048, 32), (8192, 32)):
    li, _, _ = make_instances(eng, D, B, 3, seed0=7)
    for tag, dbg in (("persistent", 0), ("one item per workgroup", 512)):
        eng.set_tuning("cov_dbg", dbg)
        eng.set_profiling(True)
        tl = []
        for kk in range(15):
            it = li[kk % 3]
            eng.gsm_update(it["X"], it["G"], it["mu0"], it["S0"], out=(it["mu"], it["S"]))
            if kk >= 3: tl.append(eng.get_profile()["cov_update"])
        eng.set_profiling(False)
        torch.cuda.synchronize()
        outs = [li[0]["S"].clone(), li[0]["mu"].clone()]
        if dbg == 0: ref = outs
        else: print("   bit-identical to the persistent kernel:", torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1]))
        t = float(np.mean(tl)) * 1e3
        alg = 16.0 * D * D + 16.0 * B * D
        nt = D // 32
        moved = (nt * (nt + 1) // 2) * 32 * 32 * 8.0 + 8.0 * D * D + 16.0 * B * D
        print(f"D={D} B={B} {tag:24s}: {t:7.1f} us  algorithmic {alg / t / 1e3:7.1f} GB/s ({alg / t / 1e3 / 8000:.3f} of 8 TB/s)  moved {moved / t / 1e3:7.1f} GB/s", flush=True)
    eng.set_tuning("cov_dbg", 0)
    del li
    torch.cuda.empty_cache()
