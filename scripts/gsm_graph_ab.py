#!/usr/bin/env python3
"""GSM.fit (default method) with graph=False / graph=True: marginal iteration between a 400- and a 1200-iteration fit.
usage: gsm_graph_ab.py [D B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gsmvi_amd
import _inputs as orc
shapes = [(256, 8), (512, 16), (1024, 32), (2048, 32), (1024, 64)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for kv in sys.argv[3:]:                                  # name=value tuning knobs (diagnostics)
    gsmvi_amd.get_engine().set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
for D, B in shapes:
    m, _, P = orc.make_gaussian_target(D, 0)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    for graph in (False, True):
        gsm = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
        gsm.fit(1, niter=47, batch_size=B, verbose=False, rng="device", graph=False)
        ts = {}
        for n in (400, 1200, 400, 1200):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gsm.fit(1, niter=n - 1, batch_size=B, verbose=False, rng="device", graph=graph)
            torch.cuda.synchronize(); ts[n] = min(ts.get(n, 1e9), time.perf_counter() - t0)
        marg = 800 / (ts[1200] - ts[400])
        print(f"D={D} B={B} graph={graph} ({gsm.method_used}): 400 it {ts[400] * 1e3:7.1f} ms ({400 / ts[400]:7.0f} it/s)  1200 it "
              f"{ts[1200] * 1e3:7.1f} ms ({1200 / ts[1200]:7.0f} it/s)  marginal {marg:7.0f} it/s = {1e6 / marg:6.1f} us  replays {gsm.graph_replays}", flush=True)
