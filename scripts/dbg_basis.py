import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np, gsmvi_amd
from gsmvi_amd.targets import GaussianTarget
eng = gsmvi_amd.get_engine()
for D, B, niter, ck in ((1024, 128, 500, 50), (1024, 32, 1500, 250), (256, 16, 1500, 250)):
    g = torch.Generator(device=eng.device); g.manual_seed(5)
    kw = dict(dtype=torch.float64, device=eng.device, generator=g)
    m = torch.rand(D, **kw); L = torch.randn(D, D, **kw)
    cov_t = L @ L.T + 1e-3 * torch.eye(D, dtype=torch.float64, device=eng.device)
    P = torch.linalg.inv(cov_t)
    tgt = GaussianTarget(m.cpu().numpy(), precision=(0.5 * (P + P.T)).cpu().numpy())
    class Snap:
        checkpoint = ck; device_native = True
        def __init__(self, name): self.name = name
        def __call__(self, i, params, lp, key, nevals=1):
            print(D, B, self.name, i, "rel err cov vs target %.3e  mean %.3e" % (float((params[1]-cov_t).abs().max()/cov_t.abs().max()), float((params[0]-m).abs().max()/m.abs().max())), flush=True)
    sched = lambda c: 100.0 / c
    for name, knob in (("factor_old", 0), ("factor_newbasis", 1)):
        eng.set_tuning("bam_basis", knob)
        bam = gsmvi_amd.BaM(D, None, tgt.lp_g)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bam.fit(7, gsmvi_amd.Regularizers().custom(sched), batch_size=B, niter=niter, verbose=False, monitor=Snap(name), as_torch=True, method="factor")
        torch.cuda.synchronize()
        print(name, "reverts", bam.n_reverts, "it/s %.0f" % (niter / (time.perf_counter() - t0)))
    eng.set_tuning("bam_basis", 0)
