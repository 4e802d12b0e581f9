#!/usr/bin/env python3
"""Diagnostic: 30 factor-form BaM updates for `rocprofv3 --kernel-trace` (scripts/trace_timeline.py prints the last one's
launches with start offsets and queues).  usage: bamf_trace.py D B [bam_basis]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gsmvi_amd
import _inputs as orc
D, B = int(sys.argv[1]), int(sys.argv[2])
eng = gsmvi_amd.get_engine()
if len(sys.argv) > 3:
    eng.set_tuning("bam_basis", int(sys.argv[3]))
st = orc.make_update_state(D, B, 1)
X, G, mu0, Z = (eng.asarray(st[k]) for k in ("samples", "vs", "mu0", "Z"))
F0 = eng.asarray(st["L"].T.copy())
out = (eng.empty(D), eng.empty(D, D)); flag = eng.new_flag()
for _ in range(30):
    eng.bam_factor_update(Z, X, G, mu0, F0, 1.0, out=out, flag=flag)
torch.cuda.synchronize()
