#!/usr/bin/env python3
"""BaM.fit (factor form) with and without hipGraph replay of its iteration blocks: whole-fit time at two lengths, the marginal
iteration rate between them and the one-off capture cost.  usage: bam_graph_ab.py [D B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gsmvi_amd
import _inputs as orc
shapes = [(256, 8), (1024, 32), (1024, 64), (1024, 128)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for kv in sys.argv[3:]:                                  # name=value tuning knobs (diagnostics)
    gsmvi_amd.get_engine().set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
sched = lambda i: 100.0 / (1 + i)
for D, B in shapes:
    m, _, P = orc.make_gaussian_target(D, 0)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    for graph in (False, True):
        bam = gsmvi_amd.BaM(D, tgt.lp, tgt.lp_g)
        bam.fit(1, sched, niter=47, batch_size=B, verbose=False, rng="device", method="factor", graph=False)
        ts = {}
        for n in (400, 1200, 400, 1200):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            bam.fit(1, sched, niter=n - 1, batch_size=B, verbose=False, rng="device", method="factor", graph=graph)
            torch.cuda.synchronize(); ts[n] = min(ts.get(n, 1e9), time.perf_counter() - t0)
        marg = 800 / (ts[1200] - ts[400])
        print(f"D={D} B={B} graph={graph}: 400 it {ts[400] * 1e3:7.1f} ms ({400 / ts[400]:7.0f} it/s)  1200 it {ts[1200] * 1e3:7.1f} ms "
              f"({1200 / ts[1200]:7.0f} it/s)  marginal {marg:7.0f} it/s = {1e6 / marg:6.1f} us  replays {bam.graph_replays}", flush=True)
