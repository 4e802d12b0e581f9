"""Shape tests DERIVED from the dispatch code (round-5 verdict, item 3).  tests/dispatch_thresholds.py lists every shape threshold
the C ABI branches on (mirrored from the sources: tests/test_thresholds_mirror.py); here every update kind whose dispatch
contains a threshold runs at threshold - 2 .. + 2 against the oracle, each case with a FRESH, EXACTLY SIZED context (the two bugs
of round 5 -- B >> D slabs, unsummed slice tails below D = 256 -- were a workspace sized for other shapes and a launch geometry
nobody had run), plus the default fits against the dense loops on forced samples at the thresholds of the method rule."""
import numpy as np
import pytest

from conftest import rel_err
import dispatch_thresholds as dt

pytestmark = pytest.mark.gpu

_TARGETS = {}


def _target(D):
    from oracle import gsm_oracle as orc
    if D not in _TARGETS:
        if D <= 320:
            m, _, P = orc.make_gaussian_target(D, 1000 + D)
        else:                                           # (large D: a diagonal-plus-low-rank precision keeps the host side O(D^2))
            rs = np.random.RandomState(1000 + D)
            m = rs.rand(D)
            U = rs.standard_normal((D, 4)) / np.sqrt(D)
            P = np.diag(0.5 + rs.rand(D)) + U @ U.T
        _TARGETS.clear()
        _TARGETS[D] = (m, P)
    return _TARGETS[D]


def _state(D, B, seed):
    """mu0, a dense non-triangular factor F0, whitened draws Z, samples X = mu0 + Z F0 and the scores of a Gaussian target"""
    rs = np.random.RandomState(seed)
    F0 = rs.standard_normal((D, D)) / np.sqrt(D) + 0.7 * np.eye(D)
    mu0 = rs.standard_normal(D)
    Z = rs.standard_normal((B, D))
    X = mu0 + Z @ F0
    m, P = _target(D)
    G = -(X - m) @ P
    return mu0, F0, Z, X, G


@pytest.mark.parametrize("kind,D,B,why", dt.cases(), ids=[f"{k}-{D}-{B}" for k, D, B, _ in dt.cases()])
def test_update_kinds_around_every_dispatch_threshold(kind, D, B, why):
    import torch
    import gsmvi_amd
    from gsmvi_amd.engine import HipEngine
    from oracle import gsm_oracle as orc
    from oracle import bam_oracle as borc
    eng = HipEngine(torch.cuda.current_device(), max_D=D + (D & 1), max_B=B)        # fresh and exactly sized (odd D runs padded to D + 1)
    try:
        if kind == "potrf":
            rs = np.random.RandomState(D)
            A = rs.standard_normal((D, D))
            S = A @ A.T / D + 0.5 * np.eye(D)
            R, flag = eng.potrf(eng.asarray(S))
            Rn = R.cpu().numpy()
            assert eng.read_flag(flag) == 0 and rel_err(Rn.T @ Rn, S) < 1e-13 and np.array_equal(Rn, np.triu(Rn)), why
            return
        mu0, F0, Z, X, G = _state(D, B, 7 * D + B)
        S0 = F0.T @ F0
        reg = 1.5
        if kind == "gsm":
            mu_o, S_o = orc.gsm_update_batched(X, G, mu0, S0)
            mu, S = gsmvi_amd.gsm_update(X, G, mu0, S0, engine=eng)
            assert rel_err(mu, mu_o) < 1e-10 and rel_err(S, S_o) < 1e-11, why
            assert np.array_equal(S, S.T), why
        elif kind == "bam" and B > HipEngine.bam_max_batch:
            with pytest.raises(ValueError):                          # beyond the largest chain: refused by the driver, up front
                gsmvi_amd.bam_update(X, G, mu0, S0, reg, engine=eng)
        elif kind == "bam":
            mu_o, S_o = borc.bam_lowrank_update_exact(X, G, mu0, S0, reg)
            mu, S = gsmvi_amd.bam_update(X, G, mu0, S0, reg, engine=eng)
            assert rel_err(mu, mu_o) < 1e-7 and rel_err(S, 0.5 * (S_o + S_o.T)) < 1e-8, why
        elif kind in ("gsmf", "bamf"):
            dv = [eng.asarray(a) for a in (Z, X, G, mu0, F0)]
            if 2 * B > min(D, HipEngine.factor_max_rows):
                with pytest.raises(gsmvi_amd.GsmviError) as ei:          # beyond the largest chain: refused before anything runs
                    (eng.gsm_factor_update(*dv) if kind == "gsmf" else eng.bam_factor_update(*dv, reg))
                assert ei.value.status == 5, why
                return
            if kind == "gsmf":
                mu_o, S_o = orc.gsm_update_batched(X, G, mu0, S0)
                mu, F, flag = eng.gsm_factor_update(*dv)
                tol_s, tol_m = 1e-9, 1e-9
            else:
                mu_o, S_o = borc.bam_lowrank_update_exact(X, G, mu0, S0, reg)
                S_o = 0.5 * (S_o + S_o.T)
                mu, F, flag = eng.bam_factor_update(*dv, reg)
                tol_s, tol_m = 1e-7, 1e-7
            assert eng.read_flag(flag) == 0, why
            Fn = F.cpu().numpy()
            # (2B within a few rows of D: the basis of the 2B x 2B chain is nearly singular by construction)
            if 2 * B >= D - 4:
                tol_s, tol_m = 1e-5, 1e-6
            assert rel_err(Fn.T @ Fn, S_o) < tol_s and rel_err(mu.cpu().numpy(), mu_o) < tol_m, (why, rel_err(Fn.T @ Fn, S_o))
    finally:
        eng.close()


def _fit_cases():
    """(cls, D, B): the method rule's thresholds -- GSM auto: 2B <= min(D, 128), <= 256 from D = 1024; BaM auto (jitter 0): 2B <=
    min(D, 256) -- and the chain thresholds 2B = 64, 128, at -2 / 0 / +2 rows"""
    out = []
    for B in (31, 32, 33, 63, 64, 65):
        out.append(("gsm", 264, B))
        out.append(("bam", 264, B))
    for B in (127, 128, 129):
        out.append(("bam", 264, B))
    for D in (126, 128, 130):
        out.append(("gsm", D, 64))                       # 2B = 128 against D
        out.append(("bam", D, 64))
    return out


@pytest.mark.parametrize("cls,D,B", _fit_cases())
def test_default_fits_follow_the_dense_loops_around_the_method_thresholds(cls, D, B):
    """The default fit (method="auto"; BaM with jitter = 0, where auto may take the factor form) records its samples; forced into
    the dense loop both walk the same trajectory -- whichever method the rule picked on either side of its thresholds."""
    import torch
    import gsmvi_amd
    from gsmvi_amd.engine import HipEngine
    from gsmvi_amd.targets import GaussianTarget, device_score
    eng = HipEngine(torch.cuda.current_device(), max_D=D, max_B=B)
    try:
        m, P = _target(D)
        tgt = GaussianTarget(m, precision=P, engine=eng)
        seen = []

        @device_score
        def lp_g(x):
            seen.append(x.clone())
            return tgt.lp_g(x)

        niter = 8
        sched = lambda i: 100.0 / (1 + i)                            # noqa: E731
        if cls == "bam":
            f = gsmvi_amd.BaM(D, None, lp_g, engine=eng)
            mean_f, cov_f = f.fit(7, sched, batch_size=B, niter=niter, verbose=False, jitter=0.0)
            forced = [x.cpu().numpy() for x in seen]
            mean_d, cov_d = gsmvi_amd.BaM(D, None, tgt.lp_g, engine=eng).fit(7, sched, batch_size=B, niter=niter, verbose=False,
                                                                            jitter=0.0, forced_samples=forced, method="dense")
            want = "factor" if 2 * B <= min(D, 256) else "dense"
        else:
            f = gsmvi_amd.GSM(D, None, lp_g, engine=eng)
            mean_f, cov_f = f.fit(7, batch_size=B, niter=niter, verbose=False)
            forced = [x.cpu().numpy() for x in seen]
            mean_d, cov_d = gsmvi_amd.GSM(D, None, tgt.lp_g, engine=eng).fit(7, batch_size=B, niter=niter, verbose=False,
                                                                            forced_samples=forced, method="dense")
            want = "factor" if 2 * B <= min(D, 256 if D >= 1024 else 128) else "dense"
        assert f.method_used == want and f.n_reverts == 0, (f.method_used, f.n_reverts)
        tol = 1e-5 if 2 * B >= D - 4 else 1e-7
        assert rel_err(mean_f, mean_d) < tol and rel_err(cov_f, cov_d) < tol, (rel_err(mean_f, mean_d), rel_err(cov_f, cov_d))
    finally:
        eng.close()
