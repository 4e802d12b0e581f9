"""Pins oracle/gsm_oracle.py against the golden vectors generated from the reference
(tests/golden/make_golden.py imported gsmvi.gsm_numpy).  CPU only."""
import numpy as np
import pytest

from oracle import gsm_oracle as orc
from conftest import rel_err

TOL = 1e-12   # fp64 restatement vs fp64 reference; BASELINE's bar is 1e-5


def _cases(golden):
    g = golden("g1_update.npz")
    return g, [str(c) for c in g["cases"]]


def test_g1_faithful_equals_reference(golden):
    g, cases = _cases(golden)
    for c in cases:
        mu, S = orc.gsm_update_faithful(g[f"{c}/samples"], g[f"{c}/vs"], g[f"{c}/mu0"], g[f"{c}/S0"])
        assert rel_err(mu, g[f"{c}/mu"]) < TOL, c
        assert rel_err(S, g[f"{c}/S"]) < TOL, c


def test_g1_batched_equals_reference(golden):
    g, cases = _cases(golden)
    for c in cases:
        mu, S = orc.gsm_update_batched(g[f"{c}/samples"], g[f"{c}/vs"], g[f"{c}/mu0"], g[f"{c}/S0"])
        assert rel_err(mu, g[f"{c}/mu"]) < TOL, c
        assert rel_err(S, g[f"{c}/S"]) < TOL, c
        assert np.array_equal(S, S.T) or rel_err(S, S.T) < 1e-15


def test_g1_per_sample_terms(golden):
    g, cases = _cases(golden)
    n = 0
    for c in cases:
        if f"{c}/dmu_b" not in g.files:
            continue
        t = orc.gsm_per_sample_terms(g[f"{c}/samples"], g[f"{c}/vs"], g[f"{c}/mu0"], g[f"{c}/S0"])
        assert rel_err(t["dmu"], g[f"{c}/dmu_b"]) < TOL, c
        assert rel_err(t["rho"], g[f"{c}/rho_b"]) < TOL, c
        n += 1
    assert n >= 10


def test_factor_form_equals_reference(golden):
    """Sigma = L L^T, x = mu + L z: the whitened update reproduces the reference's (mu, Sigma)."""
    for D, B, seed in [(5, 2, 0), (10, 2, 1), (64, 8, 2)]:
        st = orc.make_update_state(D, B, seed)
        mu_ref, S_ref = orc.gsm_update_faithful(st["samples"], st["vs"], st["mu0"], st["S0"])
        mu, F, ok = orc.gsm_factor_update(st["Z"], st["vs"], st["mu0"], st["L"])
        assert ok
        assert rel_err(mu, mu_ref) < 1e-11
        assert rel_err(F @ F.T, S_ref) < 1e-11


@pytest.mark.parametrize("D", [5, 10])
def test_g2_teacher_forced_trajectory(golden, D):
    """Replays the reference's recorded samples through the oracle fit loop; every (mean_i, cov_i)
    must match, including the niter+1 iteration count and the final state (G3)."""
    g = golden(f"g2_traj_D{D}.npz")
    m, P = g["target_m"], g["target_P"]
    rec = []
    mean, cov = orc.gsm_fit(D, None, lambda x: orc.gaussian_score(x, m, P), key=int(g["key"]),
                            niter=int(g["niter"]), batch_size=2, forced_samples=g["samples"], record=rec)
    assert len(rec) == int(g["niter"]) + 1 == g["samples"].shape[0]
    means, covs = g["means"], g["covs"]            # state BEFORE iteration i; last row = final
    for i, (_, vs, mu_i, cov_i, good) in enumerate(rec):
        assert rel_err(vs, g["vs"][i]) < 1e-9
        assert rel_err(mu_i, means[i + 1]) < 1e-9, i
        assert rel_err(cov_i, covs[i + 1]) < 1e-9, i
    assert rel_err(mean, g["mean_fit"]) < 1e-9 and rel_err(cov, g["cov_fit"]) < 1e-9
    # G3 / K3: converged to the target
    assert rel_err(mean, m) < 1e-10 and rel_err(cov, g["target_cov"]) < 1e-10


@pytest.mark.parametrize("D", [5, 10])
def test_g2_free_running_same_seed(golden, D):
    """Same key, own sampler restatement (legacy MT19937 + SVD): on the same LAPACK build the
    sample stream is bit-identical to the reference's, so the free-running fit lands on the same
    converged endpoint (K3)."""
    g = golden(f"g2_traj_D{D}.npz")
    m, P = g["target_m"], g["target_P"]
    mean, cov = orc.gsm_fit(D, None, lambda x: orc.gaussian_score(x, m, P), key=99, niter=500, batch_size=2)
    assert rel_err(mean, m) < 1e-9 and rel_err(cov, g["target_cov"]) < 1e-9


def test_g4_revert(golden):
    g = golden("g4_revert.npz")
    mu, S = orc.gsm_update_faithful(g["samples"], g["vs"], g["mu0"], g["S0"])
    assert not orc.cov_is_good(S)
    assert not bool(g["is_good"]) and not bool(g["nan_is_good"])
    assert not orc.cov_is_good(np.full((3, 3), np.nan))
    # the fit keeps BOTH mean and cov on a bad update (gsm_numpy.py:121-125)
    D = g["mu0"].shape[0]
    mean, cov = orc.gsm_fit(D, None, lambda x: g["vs"], key=0, mean=g["mu0"], cov=g["S0"], niter=0,
                            batch_size=2, forced_samples=g["samples"][None])
    assert np.array_equal(mean, g["mu0"]) and np.array_equal(cov, g["S0"])


def test_g5_monitor_cadence(golden):
    g = golden("g5_monitor.npz")
    m, P = g["target_m"], g["target_P"]

    class Mon:
        checkpoint = 3

        def __init__(self):
            self.calls = []

        def __call__(self, i, mc, lp, key, nevals=0):
            assert isinstance(mc, list) and mc[0].ndim == 1 and mc[1].ndim == 2
            self.calls.append((i, nevals))

    mon = Mon()
    n = [0]

    def lp_g(x):
        n[0] += 1
        return orc.gaussian_score(x, m, P)

    orc.gsm_fit(4, None, lp_g, key=5, niter=10, batch_size=2, monitor=mon)
    assert mon.calls == [tuple(r) for r in g["calls"].tolist()]
    assert n[0] == int(g["n_lp_g"]) == 11


def test_g6_sampler_stream(golden):
    g = golden("g6_sampler.npz")
    tags = sorted({k.split("/")[0] for k in g.files})
    for t in tags:
        seed = int(t.split("_s")[1])
        B = int(t.split("_B")[1].split("_")[0])
        rs = np.random.RandomState(seed)
        x = orc.svd_sampler(rs, g[f"{t}/mean"], g[f"{t}/cov"], B)
        x2 = orc.svd_sampler(rs, g[f"{t}/mean"], g[f"{t}/cov"], B)
        # LAPACK-build dependent in principle (SVD basis); identical build here => tight
        assert rel_err(x, g[f"{t}/x"]) < 1e-10 and rel_err(x2, g[f"{t}/x2"]) < 1e-10


def test_kat_k1_score_match():
    """K1: B=1, after the update -Sigma'^-1 (x - mu') = g exactly."""
    st = orc.make_update_state(12, 1, 3)
    mu, S = orc.gsm_update_batched(st["samples"], st["vs"], st["mu0"], st["S0"])
    g = -np.linalg.solve(S, st["samples"][0] - mu)
    assert rel_err(g, st["vs"][0]) < 1e-9


def test_kat_k2_fixed_point():
    """K2: at the Gaussian target the update is zero."""
    m, cov_t, P = orc.make_gaussian_target(9, 11)
    rs = np.random.RandomState(0)
    X = m + rs.standard_normal((4, 9)) @ np.linalg.cholesky(cov_t).T
    mu, S = orc.gsm_update_batched(X, orc.gaussian_score(X, m, P), m, cov_t)
    assert rel_err(mu, m) < 1e-9 and rel_err(S, cov_t) < 1e-9
