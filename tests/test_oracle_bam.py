"""Known-answer tests anchoring the BaM restatement (oracle/bam_oracle.py; PARITY UNPINNED against
gsmvi/bam.py because jax is absent) to the pinned GSM oracle.  SURVEY section 4, K4-K6.  CPU only."""
import numpy as np
import pytest

from oracle import gsm_oracle as orc
from oracle import bam_oracle as borc
from conftest import rel_err


def test_k4_fixed_point_full_and_lowrank():
    m, cov_t, P = orc.make_gaussian_target(8, 21)
    rs = np.random.RandomState(1)
    for B, reg in [(3, 0.3), (5, 10.0)]:
        X = m + rs.standard_normal((B, 8)) @ np.linalg.cholesky(cov_t).T
        G = orc.gaussian_score(X, m, P)
        for f, tol in [(borc.bam_update_full, 1e-7), (borc.bam_lowrank_update_exact, 1e-9),
                       (borc.bam_lowrank_update_svds, 1e-8)]:
            mu, S = f(X, G, m, cov_t, reg)
            assert rel_err(mu, m) < tol and rel_err(S, cov_t) < tol, (f.__name__, B, reg)


def test_k5_bam_b1_large_reg_is_gsm():
    st = orc.make_update_state(10, 1, 5)
    mu_g, S_g = orc.gsm_update_faithful(st["samples"], st["vs"], st["mu0"], st["S0"])
    mu, S = borc.bam_lowrank_update_exact(st["samples"], st["vs"], st["mu0"], st["S0"], 1e7)
    assert rel_err(mu, mu_g) < 1e-5 and rel_err(S, S_g) < 1e-5
    mu2, S2 = borc.bam_lowrank_update_exact(st["samples"], st["vs"], st["mu0"], st["S0"], 1e5)
    assert rel_err(mu2, mu_g) > rel_err(mu, mu_g)      # O(1/reg) gap shrinks with reg


@pytest.mark.parametrize("D,B,reg", [(6, 2, 0.01), (16, 4, 1.0), (64, 8, 100.0)])
def test_k6_lowrank_equals_full(D, B, reg):
    st = orc.make_update_state(D, B, 2)
    mu_f, S_f = borc.bam_update_full(st["samples"], st["vs"], st["mu0"], st["S0"], reg)
    S_f = 0.5 * (S_f + S_f.T)
    for f in (borc.bam_lowrank_update_exact, borc.bam_lowrank_update_svds):
        mu, S = f(st["samples"], st["vs"], st["mu0"], st["S0"], reg)
        assert rel_err(mu, mu_f) < 1e-5 and rel_err(S, S_f) < 1e-5, f.__name__


def test_exact_factor_is_exact():
    st = orc.make_update_state(12, 5, 0)
    _, _, U, _ = borc.bam_stats(st["samples"], st["vs"], st["mu0"], st["S0"], 3.0)
    Q = borc.exact_Q(st["vs"], 3.0)
    assert Q.shape == (12, 6) and rel_err(Q @ Q.T, U) < 1e-13


def test_exact_factor_allows_B_ge_D(golden):
    g = golden("r1_bam.npz")
    assert "restatement-derived" in str(g["label"])
    for c in [str(x) for x in g["cases"]]:
        args = (g[f"{c}/samples"], g[f"{c}/vs"], g[f"{c}/mu0"], g[f"{c}/S0"], float(g[f"{c}/reg"]))
        mu, S = borc.bam_lowrank_update_exact(*args)
        assert rel_err(mu, g[f"{c}/mu_lowrank"]) < 1e-12 and rel_err(S, g[f"{c}/S_lowrank"]) < 1e-12
        assert rel_err(S, 0.5 * (g[f"{c}/S_full"] + g[f"{c}/S_full"].T)) < 1e-5


def test_regularizers_count_calls_not_iterations():
    r = borc.Regularizers()
    lin = r.linear(100.0)
    assert [lin(0), lin(0), lin(7)] == [100.0, 50.0, 100.0 / 3]
    r.reset()
    cus = r.custom(lambda i: 100 / (1 + i))
    assert cus(99) == 50.0 and cus(99) == 100 / 3
    con = r.constant(2.5)
    assert con(1) == 2.5 and r.counter == 3


def test_bam_fit_converges_on_gaussian():
    D = 5
    m, cov_t, P = orc.make_gaussian_target(D, 17)
    reg = borc.Regularizers()
    mean, cov = borc.bam_fit(D, None, lambda x: orc.gaussian_score(x, m, P), key=99,
                             regf=reg.custom(lambda i: 100 / (1 + i)), niter=100, batch_size=2, jitter=1e-6)
    assert np.allclose(mean, m, atol=1e-3) and np.allclose(cov, cov_t, atol=1e-3, rtol=1e-3)
