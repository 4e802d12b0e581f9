#!/usr/bin/env python3
"""A/B: one panel product (the sampler X = mu + Z F, D x D matrix streamed once) by split-K count, in a replayed hipGraph of
NREP dependent calls (HBM-cold ring of matrices).  kc > 1 needs the k_panel_finish launch behind the product.
usage: ab_panel_kc.py [D B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gsmvi_amd
D, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 32)
eng = gsmvi_amd.get_engine()
g = torch.Generator(device="cuda"); g.manual_seed(0)
kw = dict(dtype=torch.float64, device="cuda", generator=g)
NR = 24
Fs = [torch.randn(D, D, **kw) for _ in range(NR)]
Z = torch.randn(B, D, **kw); mu = torch.randn(D, **kw); X = torch.empty(B, D, dtype=torch.float64, device="cuda")
for kc in (0, 1, 2, 4):
    eng.set_tuning("panel_kc", kc)
    for F in Fs: eng.sample(Z, mu, F, out=X)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for F in Fs: eng.sample(Z, mu, F, out=X)
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / NR)
    ts.sort()
    print(f"D={D} B={B} panel_kc={kc} (0 = auto): {ts[len(ts) // 2]:.2f} us per sampler call (median), min {ts[0]:.2f}")
eng.set_tuning("panel_kc", 0)
