"""Synthetic inputs of the measurement scripts: a seeded Gaussian target and one update's state.

The scripts under scripts/ measure the HIP path; they do not check it, so they take their inputs from here and never
import oracle/ (test infrastructure).  scripts/configs_bench.py is the one exception: its CPU-baseline columns time the
oracle beside the GPU, like bench.py's cpu_baseline leg."""
import numpy as np


def make_gaussian_target(D, seed, cond=None):
    """(m, cov, precision) of the example's target (example_gsm_numpy.py:11-14): m ~ U(0,1)^D, cov = L L^T + 1e-3 I;
    ``cond`` rescales the spectrum log-uniformly to that condition number (BASELINE config 5)."""
    rs = np.random.RandomState(seed)
    m = rs.random_sample(D)
    L = rs.normal(size=(D, D))
    cov = L @ L.T + 1e-3 * np.eye(D)
    if cond is not None:
        w, Q = np.linalg.eigh(cov)
        w = np.logspace(0.0, np.log10(cond), D) * max(w.min(), 1e-300)
        cov = (Q * w) @ Q.T
        cov = 0.5 * (cov + cov.T)
    return m, cov, np.linalg.inv(cov)


def gaussian_score(X, m, P):
    return -(X - m) @ P.T


def make_update_state(D, B, seed, target_seed=None):
    """mu0 ~ N(0, I), S0 = A A^T / D + 0.1 I, samples mu0 + z chol(S0)^T, scores of the seeded target at the samples."""
    rs = np.random.RandomState(1000 + seed)
    m, cov_t, P = make_gaussian_target(D, seed if target_seed is None else target_seed)
    mu0 = rs.standard_normal(D)
    A = rs.standard_normal((D, D))
    S0 = A @ A.T / D + 0.1 * np.eye(D)
    S0 = 0.5 * (S0 + S0.T)
    Lc = np.linalg.cholesky(S0)
    Z = rs.standard_normal((B, D))
    X = mu0 + Z @ Lc.T
    return dict(samples=X, vs=gaussian_score(X, m, P), mu0=mu0, S0=S0, Z=Z, L=Lc, m=m, P=P, cov_t=cov_t)
