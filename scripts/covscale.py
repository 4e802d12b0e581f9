#!/usr/bin/env python3
"""Covariance-update kernel (gsm_apply) and whole update vs D at fixed B, HBM-cold ring inside a replayed
hipGraph: algorithmic GB/s = (16 D^2 + 16 B D) / time (apply) and (24 D^2 + 72 B D) / time (full)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gsmvi_amd
from bench import make_instances
eng = gsmvi_amd.get_engine()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for kv in sys.argv[2:]:
    k, v = kv.split("="); eng.set_tuning(k, int(v))
for D in (1024, 2048, 4096, 8192):
    n_inst = max(2, 320 * 2**20 // (16 * D * D) + 1)
    inst, m, P = make_instances(eng, D, B, n_inst)
    recs = [eng.gsm_local_stage(it["X"], it["G"], it["mu0"], it["S0"]) for it in inst]
    def f_apply(k): eng.gsm_apply(recs[k], inst[k]["mu0"], inst[k]["S0"], out=(inst[k]["mu"], inst[k]["S"]))
    def f_full(k): eng.gsm_update(inst[k]["X"], inst[k]["G"], inst[k]["mu0"], inst[k]["S0"], out=(inst[k]["mu"], inst[k]["S"]))
    out = []
    for name, f, nbytes in (("apply", f_apply, 16.0 * D * D + 16.0 * B * D), ("full", f_full, 24.0 * D * D + 72.0 * B * D)):
        reps = max(n_inst, 42 // n_inst * n_inst)
        for k in range(reps): f(k % n_inst)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(reps): f(k % n_inst)
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        nrep = 20 if D <= 2048 else 5
        for _ in range(nrep): g.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / (nrep * reps) * 1e6
        out.append(f"{name} {us:.2f} us = {nbytes / us / 1e3:.0f} GB/s")
    print(f"D={D} B={B} inst={n_inst}: " + "   ".join(out), flush=True)
    del inst, recs
    torch.cuda.empty_cache()
