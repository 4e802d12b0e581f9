#!/usr/bin/env python3
"""Per-stage time of the GSM update inside a replayed hipGraph (no profiler): local stage (panel +
scalars), apply (covariance update), full update; cold ring vs one cache-resident instance."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsmvi_amd
from bench import make_instances

ap = argparse.ArgumentParser()
ap.add_argument("--D", type=int, default=1024)
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--tune", action="append", default=[])
ap.add_argument("--ldpad", type=int, default=0)
args = ap.parse_args()
eng = gsmvi_amd.get_engine()
for kv in args.tune:
    k, v = kv.split("="); eng.set_tuning(k, int(v))
D, B = args.D, args.B
for n_inst in (1, max(2, 320 * 2**20 // (16 * D * D) + 1)):
    inst, m, P = make_instances(eng, D, B, n_inst, ldpad=args.ldpad)
    recs = [eng.empty(B, eng.record_len(D)) for _ in range(n_inst)]
    for k, it in enumerate(inst):
        eng.gsm_local_stage(it["X"], it["G"], it["mu0"], it["S0"], out=recs[k])
    def f_local(k): eng.gsm_local_stage(inst[k]["X"], inst[k]["G"], inst[k]["mu0"], inst[k]["S0"], out=recs[k])
    def f_apply(k): eng.gsm_apply(recs[k], inst[k]["mu0"], inst[k]["S0"], out=(inst[k]["mu"], inst[k]["S"]))
    def f_full(k): eng.gsm_update(inst[k]["X"], inst[k]["G"], inst[k]["mu0"], inst[k]["S0"], out=(inst[k]["mu"], inst[k]["S"]))
    def f_score(k): eng.gaussian_score(inst[k]["X"], m, P, out=inst[k]["G"])
    res = {}
    for name, f in (("local(panel+scalars)", f_local), ("apply(cov)", f_apply), ("full", f_full), ("score(panel+finish)", f_score)):
        reps = max(n_inst, 42 // n_inst * n_inst)
        for k in range(reps): f(k % n_inst)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(reps): f(k % n_inst)
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / (20 * reps) * 1e6
    print(f"D={D} B={B} instances={n_inst}: " + "  ".join(f"{k}={v:.2f}us" for k, v in res.items()))
