#!/usr/bin/env python3
"""Soak of gsmvi_potrf_f64's persistent launch (k_potrf_dag): the same matrix factored again and again must give the same bits
(every hand-off of the task graph is a place where a missing ordering would show up as a changed tile, a wrong pivot or the
abort code D + 1) -- sizes on and off the 64-grid, each also compared with the launch-per-step form once.
usage: potrf_soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
eng = gsmvi_amd.get_engine()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
sizes = [130, 200, 256, 1000, 1024, 1091, 2048, 3000, 4096]
cases = []
for D in sizes:
    g = torch.Generator(device="cuda"); g.manual_seed(D)
    A = torch.randn(D, D + 8, dtype=torch.float64, device="cuda", generator=g)
    S = (A @ A.T / D + 0.05 * torch.eye(D, dtype=torch.float64, device="cuda")).contiguous()
    eng.set_tuning("potrf_dag", 0)
    Rs, f = eng.potrf(S); Rs = Rs.clone(); assert eng.read_flag(f) == 0
    eng.set_tuning("potrf_dag", 1)
    R0, f = eng.potrf(S); R0 = R0.clone(); assert eng.read_flag(f) == 0
    assert float((R0 - Rs).abs().max()) <= 1e-13 * float(Rs.abs().max()), D
    cases.append((D, S, R0, eng.empty(D, D), eng.new_flag()))
t0, calls, bad, rounds = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    for D, S, R0, R, flag in cases:
        for _ in range(3):
            eng.potrf(S, out=R, flag=flag)
            calls += 1
            if eng.read_flag(flag) != 0 or not torch.equal(R, R0):
                bad += 1
                print(f"MISMATCH D={D} flag={eng.read_flag(flag)} max diff {float((R - R0).abs().max()):.3e}")
    rounds += 1
print(f"potrf soak: {calls} factorisations in {time.time() - t0:.0f} s over D = {sizes}, {rounds} rounds, mismatches: {bad}")
sys.exit(1 if bad else 0)
