#!/usr/bin/env python3
"""Soak: the round-3 launch structures (chain as a rider workgroup, side-stream fork, one-launch BaM chain, wide panels) called
back to back for a while, eagerly and from replayed graphs, every result compared bit for bit with the first one.
usage: soak_round3.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
from oracle import gsm_oracle as orc
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
eng = gsmvi_amd.get_engine()
for kv in sys.argv[2:]:                     # knob=value ...
    k, v = kv.split("=")
    eng.set_tuning(k, int(v))
only = os.environ.get("SOAK_ONLY")          # e.g. "4096,64"
kinds = os.environ.get("SOAK_KINDS", "gsm,bam").split(",")
mode = os.environ.get("SOAK_MODE", "both")  # eager | graph | both
bad = []
cases = []
for D, B in ((1024, 32), (256, 8), (4096, 64), (1024, 64)):
    if only and only != f"{D},{B}":
        continue
    rs = np.random.RandomState(D + B)
    F0 = eng.asarray(rs.standard_normal((D, D)) / np.sqrt(D) + 0.6 * np.eye(D))
    mu0 = eng.asarray(rs.standard_normal(D))
    Z = eng.asarray(rs.standard_normal((B, D)))
    X = eng.sample(Z, mu0, F0)
    G = -(X - 0.3)
    cases.append((D, B, Z, X, G, mu0, F0))
ref, graphs, outs = {}, {}, {}
n_calls, t0 = 0, time.perf_counter()
rnd = 0
while time.perf_counter() - t0 < budget:
    for ci, (D, B, Z, X, G, mu0, F0) in enumerate(cases):
        for kind in kinds:
            if kind == "bam" and 2 * B > 128:
                continue
            key = (ci, kind)
            if key not in outs:
                outs[key] = (eng.empty(D), eng.empty(D, D), eng.new_flag())
            mu, F, flag = outs[key]
            call = (lambda: eng.gsm_factor_update(Z, X, G, mu0, F0, out=(mu, F), flag=flag)) if kind == "gsm" else \
                   (lambda: eng.bam_factor_update(Z, X, G, mu0, F0, 1.0, out=(mu, F), flag=flag))
            used_graph = False
            if (mode != "graph" and (rnd % 2 == 0 or mode == "eager")) or key not in graphs:
                call()
                if key not in graphs and rnd > 0:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        call(); call()
                    graphs[key] = g
            else:
                mu.zero_(); F.zero_()
                graphs[key].replay()
                used_graph = True
            torch.cuda.synchronize()
            h = (mu.clone(), F.clone(), int(flag.item()))
            if key not in ref:
                ref[key] = h
            else:
                if not (h[2] == 0 and torch.equal(h[0], ref[key][0]) and torch.equal(h[1], ref[key][1])):
                    dF = (h[1] - ref[key][1]).abs()
                    bad.append((D, B, kind, rnd, "graph" if used_graph else "eager", h[2],
                                float(dF.max()), int((dF > 0).sum()), float((h[0] - ref[key][0]).abs().max())))
                    if len(bad) <= 5:
                        nz = (dF > 0).nonzero()
                        print("MISMATCH", bad[-1], "rows", int(nz[:, 0].min()), int(nz[:, 0].max()), "cols", int(nz[:, 1].min()),
                              int(nz[:, 1].max()), flush=True)
            n_calls += 1
    rnd += 1
print(f"mismatches: {len(bad)}")
print(f"soak done: {n_calls} calls in {time.perf_counter() - t0:.0f} s over {len(ref)} (case, kind) pairs, {rnd} rounds, all bit-identical")
