"""SURVEY 8(f) rows on the GPU: the device KL monitor (f2; gsmvi/monitors.py:83-125), lbfgs_init and the ADVI harness
(f4; gsmvi/initializers.py:5-17, gsmvi/advi.py:49-112) driven by the device-resident Gaussian target, and the two
helper entry points behind them (gsmvi_gram_f64, gsmvi_whiten_rows_f64)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _kl_gauss(m0, S0, m1, S1):
    D = m0.shape[0]
    iS1 = np.linalg.inv(S1)
    d = m1 - m0
    return 0.5 * (np.trace(iS1 @ S0) + d @ iS1 @ d - D + np.linalg.slogdet(S1)[1] - np.linalg.slogdet(S0)[1])


@pytest.mark.parametrize("D", [1, 5, 64, 100, 257, 1024])
def test_gram_and_whiten_rows_against_numpy(D):
    import gsmvi_amd
    eng = gsmvi_amd.get_engine()
    rs = np.random.RandomState(D)
    F = rs.standard_normal((D, D))
    Cd = eng.gram(eng.asarray(F)).cpu().numpy()
    assert rel_err(Cd, F.T @ F) < 1e-13 and np.array_equal(Cd, Cd.T)
    S = F.T @ F / D + 0.5 * np.eye(D)
    R = np.linalg.cholesky(S).T
    mu = rs.standard_normal(D)
    X = rs.standard_normal((7, D)) * 3.0
    Z, ld = eng.whiten_rows(eng.asarray(X), eng.asarray(mu), eng.asarray(R))
    Zo = np.linalg.solve(R.T, (X - mu).T).T
    assert rel_err(Z.cpu().numpy(), Zo) < 1e-10
    assert abs(float(ld.cpu().numpy()[0]) - np.log(np.diag(R)).sum()) < 1e-10 * max(1.0, D)
    Z0, _ = eng.whiten_rows(eng.asarray(X), None, eng.asarray(R))
    assert rel_err(Z0.cpu().numpy(), np.linalg.solve(R.T, X.T).T) < 1e-10


def test_device_kl_monitor_against_closed_form_gaussian_kl():
    """As tests/test_monitor.py does for the host monitor: reverse / forward KL of two Gaussians from samples."""
    import gsmvi_amd
    from oracle import gsm_oracle as orc
    D = 4
    m, cov_t, P = orc.make_gaussian_target(D, 1)
    mq, Sq = m + 0.3, cov_t * 1.5 + 0.1 * np.eye(D)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    norm = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(cov_t)[1]

    def lp(x):                                   # normalised log density, SUMMED over the rows (device tensor in/out)
        return tgt.lp(x) + norm * x.shape[0]

    ref = np.random.RandomState(3).multivariate_normal(m, cov_t, size=100000)
    mon = gsmvi_amd.DeviceKLMonitor(batch_size_kl=40000, checkpoint=1, ref_samples=ref)
    assert mon.device_native
    eng = gsmvi_amd.get_engine()
    mon(0, [eng.asarray(mq), eng.asarray(Sq)], lp, 11, nevals=5)
    assert abs(mon.rkl[0] - _kl_gauss(mq, Sq, m, cov_t)) < 3e-2
    assert abs(mon.fkl[0] - _kl_gauss(m, cov_t, mq, Sq)) < 3e-2
    assert mon.nevals == [5]
    mon(1, [eng.asarray(mq), eng.asarray(-np.eye(D))], lp, 11, nevals=3)       # non-PD covariance -> NaN, no exception
    assert np.isnan(mon.rkl[1]) and np.isnan(mon.fkl[1]) and mon.nevals == [5, 8]


@pytest.mark.parametrize("method", ["factor", "dense"])
def test_device_monitor_inside_the_fit_loop(method):
    """GSM.fit hands the device monitor device tensors (no D x D host copy); KL -> 0 at convergence (K3)."""
    import torch
    import gsmvi_amd
    from oracle import gsm_oracle as orc
    D = 8
    m, cov_t, P = orc.make_gaussian_target(D, 5)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    norm = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(cov_t)[1]
    seen = []

    def lp(x):
        seen.append(isinstance(x, torch.Tensor) and x.is_cuda)
        return tgt.lp(x) + norm * x.shape[0]

    ref = np.random.RandomState(1).multivariate_normal(m, cov_t, size=4096)
    mon = gsmvi_amd.DeviceKLMonitor(batch_size_kl=256, checkpoint=100, ref_samples=ref)
    gsm = gsmvi_amd.GSM(D, lp, tgt.lp_g)
    mean, cov = gsm.fit(3, niter=600, batch_size=4, verbose=False, monitor=mon, rng="device", method=method)
    assert gsm.method_used == method and all(seen) and len(mon.rkl) == 8       # i = 0,100..600 + the final call
    assert mon.rkl[0] > 0.5 and abs(mon.rkl[-1]) < 1e-8 and abs(mon.fkl[-1]) < 1e-8
    assert mon.nevals[0] == 1 and mon.nevals[1] == 1 + 400
    assert rel_err(mean, m) < 1e-9 and rel_err(cov, cov_t) < 1e-9


def test_default_fit_method_is_the_factor_form_where_it_applies():
    import gsmvi_amd
    from oracle import gsm_oracle as orc
    D = 32
    m, cov_t, P = orc.make_gaussian_target(D, 9)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    g = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
    a = g.fit(3, niter=20, batch_size=4, verbose=False)
    assert g.method_used == "factor"
    b = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g).fit(3, niter=20, batch_size=4, verbose=False, method="factor")
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    g.fit(3, niter=2, batch_size=20, verbose=False)                  # 2B > D: dense
    assert g.method_used == "dense"
    g.fit(3, niter=2, batch_size=4, verbose=False, sampler="svd")    # reference sampler: dense
    assert g.method_used == "dense"
    g2 = gsmvi_amd.GSM(256, None, lambda x: -x)
    g2.fit(1, niter=1, batch_size=80, verbose=False)                 # 2B > 128: dense
    assert g2.method_used == "dense"


def test_lbfgs_init_with_the_device_target():
    """gsmvi/initializers.py:5-17 with lp / lp_g evaluated on the GPU (GaussianTarget), then handed to GSM.fit as
    examples/example_initializers.py:44-52 does."""
    import gsmvi_amd
    from oracle import gsm_oracle as orc
    D = 12
    m, cov_t, P = orc.make_gaussian_target(D, 4)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    mu0, cov0, res = gsmvi_amd.lbfgs_init(np.zeros(D), tgt.lp, tgt.lp_g)
    assert res.success and np.allclose(mu0, m, atol=5e-3)        # scipy's stopping rule on a cond-1e4 target
    cov0 = np.asarray(cov0)
    cov0 = 0.5 * (cov0 + cov0.T)
    assert np.all(np.linalg.eigvalsh(cov0) > 0)
    mean, cov = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g).fit(0, mean=mu0, cov=cov0, niter=1500, batch_size=4, verbose=False)
    assert rel_err(mean, m) < 1e-6 and rel_err(cov, cov_t) < 1e-6


def test_advi_with_the_device_target_and_device_monitor():
    """gsmvi/advi.py:49-112 on cuda: lp is GaussianTarget.lp (device tensors, autograd through it)."""
    import torch
    import gsmvi_amd
    from oracle import gsm_oracle as orc
    D = 3
    m, cov_t, P = orc.make_gaussian_target(D, 0)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    mon = gsmvi_amd.DeviceKLMonitor(batch_size_kl=64, checkpoint=500)
    adv = gsmvi_amd.ADVI(D, tgt.lp, device="cuda")
    mean, c, losses = adv.fit(0, lambda p: torch.optim.Adam(p, lr=5e-2), batch_size=16, niter=1500, nprint=0,
                              monitor=mon)
    assert len(losses) == 1501 and np.mean(losses[-100:]) < np.mean(losses[:100])
    assert np.allclose(mean, m, atol=0.15) and np.linalg.norm(c - cov_t) / np.linalg.norm(cov_t) < 0.3
    assert len(mon.rkl) == 5 and np.isfinite(mon.rkl).all() and mon.rkl[-1] < mon.rkl[0]


def test_blocked_cholesky_with_late_replica_waves(tmp_path):
    """chol64_blk keeps replicas of the 16 x 16 diagonal block in the other column sets' panel waves (AUG >= 1), loaded at an
    unordered time; since round 3 the block is written back only behind the panel barrier (a once-in-3e5
    deviation found by soaking).  The standalone harness (which since round 4 also TESTS the inverse factor it always printed)
    is built six ways:
      plain                                                                                  -> must pass
      replica waves held back ~7 us (CHOLB_TEST_REPLICA_DELAY; a scalar branch since round 4)   -> must pass
      CHOLB_TEST_FORCE_ORDER: the replicas load their copy only after BOTH waves of set 0 have finished the panel (an LDS
        counter): the latest schedule the hardware may produce, forced                       -> must pass
      either late schedule with CHOLB_TEST_OLD_WRITEBACK (the pre-fix in-place write-back)    -> must FAIL (W = R^-T wrong at O(1))
      CHOLB_TEST_CORRUPT_REPLICA (a replica that does not hold the block)                    -> must FAIL (the harness notices)
    i.e. the mechanism the fix removes breaks the result whenever the interleaving occurs, and the shipped header is immune."""
    import os
    import shutil
    import subprocess
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    src = os.path.join(ROOT, "scripts", "chol64b_test.hip")
    variants = [("plain", [], True), ("delay", ["-DCHOLB_TEST_REPLICA_DELAY=2"], True),
                ("forced", ["-DCHOLB_TEST_FORCE_ORDER"], True),
                ("forced_oldwb", ["-DCHOLB_TEST_FORCE_ORDER", "-DCHOLB_TEST_OLD_WRITEBACK"], False),
                ("delay_oldwb", ["-DCHOLB_TEST_REPLICA_DELAY=2", "-DCHOLB_TEST_OLD_WRITEBACK"], False),
                ("corrupt", ["-DCHOLB_TEST_CORRUPT_REPLICA"], False)]
    for tag, defs, must_pass in variants:
        exe = str(tmp_path / f"chol64b_test_{tag}")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "gsm-vi_amd", "csrc"),
               src, "-o", exe] + defs
        b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert b.returncode == 0, b.stderr[-2000:]
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        if must_pass:
            assert r.returncode == 0 and "ALL OK" in r.stdout, (tag, r.stdout[-3000:])
        else:
            assert r.returncode != 0 and "FAILED" in r.stdout, (tag, "expected the harness to fail", r.stdout[-3000:])
