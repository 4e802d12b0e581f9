#!/usr/bin/env python3
"""Round-5 verdict, item 5: the panel product at D >= 2048 (rows <= 32) with the next chunk's loads in flight (k_panel_fast_p, knob
panel_w4_min_D = 2048, the default) against k_panel_fast (knob = 0): the built-in score G = -(X - m) P (product + finish launch)
over a ring of HBM-cold matrices, back to back, and the bit-level agreement of the two.  (The knob's name is that of the first
attempt -- 64-column strips, four tiles per staged chunk -- which measured 47.1 against 47.1 us at (4096, 32) and was removed.)
usage: panel_w4_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gsmvi_amd  # noqa: E402

eng = gsmvi_amd.get_engine()


def b2b(fn, warm=10, n=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for D in (1024, 2048, 4096, 8192):
    for B in (16, 32):
        g = torch.Generator(device=eng.device)
        g.manual_seed(D + B)
        kw = dict(dtype=torch.float64, device=eng.device, generator=g)
        # a ring of matrices larger than the caches: HBM-cold products, as in a fit at this size
        nring = max(2, (600 * 2 ** 20) // (8 * D * D))
        Ps = [torch.randn(D, D, **kw) for _ in range(nring)]
        X, m = torch.randn(B, D, **kw), torch.rand(D, **kw)
        G = eng.empty(B, D)
        res, outs = {}, {}
        for knob in (2048, 0):
            eng.set_tuning("panel_w4_min_D", knob)
            k = [0]

            def f():
                eng.gaussian_score(X, m, Ps[k[0] % nring], out=G)
                k[0] += 1
            res[knob] = b2b(f, 2 * nring, 6 * nring)
            outs[knob] = eng.gaussian_score(X, m, Ps[0]).clone()
        eng.set_tuning("panel_w4_min_D", 2048)
        dmax = float((outs[2048] - outs[0]).abs().max() / outs[0].abs().max())
        gb = 8.0 * D * D / 1e3
        print(f"D={D} B={B}: score (product + finish) prefetching {res[2048]:.1f} us ({gb / res[2048]:.0f} GB/s of M), "
              f"k_panel_fast {res[0]:.1f} us ({gb / res[0]:.0f} GB/s); max rel difference {dmax:.1e}", flush=True)
        del Ps
        torch.cuda.empty_cache()
