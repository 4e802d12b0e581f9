#!/usr/bin/env python3
"""BASELINE config 4 (BaM update, D=1024, B=128, reg=1): time of one dense update (HIP events, median / min over the trials,
eager and replayed from a hipGraph).  (The A/B against the round-3 kernels this chain replaced, and against a one-launch
Newton-Schulz step, is recorded in profiles/r04/c4_chain_ab.txt; those kernels were deleted afterwards.)  usage: c4_update_bench.py [D B] [prof]   -- `prof`: 60 plain updates only (for rocprofv3 --kernel-trace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
import _inputs as orc
args = [a for a in sys.argv[1:] if a != "prof"]
D, B = (int(args[0]), int(args[1])) if len(args) > 1 else (1024, 128)
eng = gsmvi_amd.get_engine()
st = orc.make_update_state(D, B, 1)
X, G, mu0, S0 = (eng.asarray(st[k]) for k in ("samples", "vs", "mu0", "S0"))
out = (eng.empty(D), eng.empty(D, D)); flag = eng.new_flag()
call = lambda: eng.bam_update(X, G, mu0, S0, 1.0, 1e-6, out=out, flag=flag)
if "prof" in sys.argv:
    for _ in range(60): call()
    torch.cuda.synchronize()
    sys.exit(0)

def measure(tag):
    for _ in range(20): call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): call()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 300 * 1e6
    assert eng.read_flag(flag) == 0
    print(f"{tag:28s} D={D} B={B}: eager median {np.median(ts):7.1f} us  min {np.min(ts):7.1f} us   replayed {tg:7.1f} us", flush=True)

measure("dense BaM update")
# round 5 A/B: the low-rank update with 64-row staging passes (4 passes, half the barriers) against the default 32-row passes
eng.set_tuning("lowrank_kp", 64)
measure("  lowrank_kp=64 (4 passes)")
eng.set_tuning("lowrank_kp", 0)
