import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, gsmvi_amd
import _inputs as orc
for D, B, n in ((64, 8, 4000), (200, 40, 1500), (300, 100, 600)):
    m, cov_t, P = orc.make_gaussian_target(D, 11)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    reg = gsmvi_amd.Regularizers()
    bam = gsmvi_amd.BaM(D, tgt.lp, tgt.lp_g, use_lowrank=True)
    t0 = time.perf_counter()
    mean, cov = bam.fit(3, reg.custom(lambda i: 100.0 / (1 + i)), niter=n, batch_size=B, verbose=False, rng="device")
    dt = time.perf_counter() - t0
    print(f"BaM D={D} B={B} niter={n}: {n/dt:.0f} it/s  mean err {np.abs(mean-m).max():.2e}  cov rel err {np.abs(cov-cov_t).max()/np.abs(cov_t).max():.2e}  reverts {bam.n_reverts}")
