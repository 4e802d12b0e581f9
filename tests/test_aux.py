"""Initialiser and ADVI comparison harness: host-side, off the hot path."""
import numpy as np
import pytest
import torch

import gsmvi_amd
from oracle import gsm_oracle as orc


def _target(D, seed):
    m, cov, P = orc.make_gaussian_target(D, seed)
    lp = lambda x: -0.5 * float((m - x) @ P @ (m - x))
    lp_g = lambda x: P @ (m - x)
    return m, cov, P, lp, lp_g


def test_lbfgs_init_gaussian_with_and_without_score():
    D = 6
    m, cov, P, lp, lp_g = _target(D, 3)
    for g in (lp_g, None):
        mu, c, res = gsmvi_amd.lbfgs_init(np.zeros(D), lp, g)
        assert res.success
        assert np.allclose(mu, m, atol=1e-4)
        c = np.asarray(c)
        assert c.shape == (D, D) and np.allclose(c, c.T, atol=1e-8)
        assert np.all(np.linalg.eigvalsh(0.5 * (c + c.T)) > 0)        # usable as an initial covariance


def test_lbfgs_init_accepts_one_element_arrays_and_torch_outputs():
    D = 4
    m, cov, P, lp, lp_g = _target(D, 1)
    lp_arr = lambda x: np.array([lp(x)])
    lp_g_t = lambda x: torch.as_tensor(lp_g(x))
    mu, c, res = gsmvi_amd.lbfgs_init(np.ones(D), lp_arr, lp_g_t, maxiter=200, maxfun=400)
    assert np.allclose(mu, m, atol=1e-4)


def test_advi_scales_roundtrip_and_elbo_value():
    D = 5
    rs = np.random.RandomState(0)
    A = rs.normal(size=(D, D)); cov = A @ A.T + np.eye(D)
    adv = gsmvi_amd.ADVI(D, lambda x: -0.5 * (x * x).sum(-1), device="cpu")
    L = np.linalg.cholesky(cov)
    assert np.allclose(adv.scales_to_cov(L[np.tril_indices(D)]), cov, atol=1e-12)
    # for q = p = N(0, I) the ELBO estimate is exactly B * D/2 * log(2 pi) whatever the draws
    loc = torch.zeros(D, dtype=torch.float64)
    scales = torch.tensor(np.eye(D)[np.tril_indices(D)], dtype=torch.float64)
    g = torch.Generator(); g.manual_seed(0)
    v = adv.neg_elbo([loc, scales], g, 7)
    assert abs(float(v) + 7 * 0.5 * D * np.log(2 * np.pi)) < 1e-10


def test_advi_fit_moves_to_the_target_and_calls_the_monitor():
    D = 3
    m, cov, P, _, _ = _target(D, 0)
    mt, Pt = torch.as_tensor(m), torch.as_tensor(P)
    lp = lambda x: -0.5 * torch.einsum("bi,ij,bj->b", x - mt, Pt, x - mt)
    calls = []

    class Mon:
        checkpoint = 100
        def __call__(self, i, state, lp, key, nevals=1):
            calls.append((i, nevals))
    adv = gsmvi_amd.ADVI(D, lp, device="cpu")
    mean, c, losses = adv.fit(0, lambda p: torch.optim.Adam(p, lr=5e-2), batch_size=16, niter=1500, nprint=0,
                              monitor=Mon())
    assert len(losses) == 1501
    assert np.allclose(mean, m, atol=0.15)
    assert np.linalg.norm(c - cov) / np.linalg.norm(cov) < 0.3
    assert calls[0] == (0, 1) and calls[1] == (100, 1600) and calls[-1][0] == 1500
    assert np.mean(losses[-100:]) < np.mean(losses[:100])
