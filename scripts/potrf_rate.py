#!/usr/bin/env python3
"""Time of gsmvi_potrf_f64 by matrix size (dispatch-to-dispatch, 10 calls).  usage: potrf_rate.py [D ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gsmvi_amd
eng = gsmvi_amd.get_engine()
split_m = int(os.environ.get("SPLIT_M", "0"))
eng.set_tuning("potrf_split_m", split_m)      # 0 = the library's default
dag = int(os.environ.get("POTRF_DAG", "1"))
eng.set_tuning("potrf_dag", dag)               # 1 = one persistent launch with look-ahead (round 6), 0 = one launch per block step
for D in [int(a) for a in sys.argv[1:]] or [256, 1024, 2048, 4096]:
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    A = torch.randn(D, D, dtype=torch.float64, device="cuda", generator=g)
    S = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device="cuda")
    R, flag = eng.empty(D, D), eng.new_flag()
    for _ in range(3):
        eng.potrf(S, out=R, flag=flag)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        eng.potrf(S, out=R, flag=flag)
    e1.record(); torch.cuda.synchronize()
    print(f"potrf D={D} dag={dag} split_m={split_m}: {e0.elapsed_time(e1) / n * 1e3:.1f} us, flag {int(flag.item())}")
