"""CPU oracle for the BaM (batch-and-match) hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

PARITY UNPINNED against the reference implementation: ``gsmvi/bam.py`` needs jax/jaxlib, which
are not installed in the build image (and the reference holds no BaM test or golden vector), so
this restatement could not be executed side by side with the reference.  Its arithmetic lives in
unpinned third-party modules (``pyproject.toml:24-29``: jax, jaxlib, scipy, no versions):
``jax.scipy.linalg.sqrtm`` / ``scipy.linalg.sqrtm`` (bam.py:19-28), ``jnp.linalg.solve``
(bam.py:65,110) and ``scipy.sparse.linalg.svds`` (bam.py:10-13).  They are restated here with
scipy 1.15.3 / numpy 2.2.6.  The restatement is anchored instead on mathematical known-answer
tests that tie it to the pinned GSM oracle (tests/test_oracle_bam.py): fixed point at the target
(K4), BaM(B=1, reg->inf) == gsm_update (K5), low-rank == full (K6).  Fixtures derived from it are
labelled "restatement-derived, not reference-derived".
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
import scipy.sparse.linalg as spla

from .gsm_oracle import cov_is_good, svd_sampler

__all__ = ["bam_stats", "bam_update_full", "bam_lowrank_update_svds", "bam_lowrank_update_exact",
           "exact_Q", "Regularizers", "bam_fit"]


def bam_stats(samples, vs, mu0, S0, reg):
    """Batch statistics and the U, V matrices of bam.py:50-60 (identical in bam.py:91-101)."""
    B = samples.shape[0]
    xbar = samples.mean(axis=0)                               # :50
    xd = samples - xbar                                       # :52
    C = xd.T @ xd / B                                         # :53  mean_b outer(xd_b, xd_b)
    gbar = vs.mean(axis=0)                                    # :55
    gd = vs - gbar                                            # :56
    Gm = gd.T @ gd / B                                        # :57
    r1 = reg / (1 + reg)
    U = reg * Gm + r1 * np.outer(gbar, gbar)                  # :59
    dm = mu0 - xbar
    V = S0 + reg * C + r1 * np.outer(dm, dm)                  # :60
    return xbar, gbar, U, V


def _real_sqrtm(M):
    """get_sqrt of bam.py:19-28: principal matrix square root, real part."""
    return np.real(sla.sqrtm(M))


def _bam_mean(mu0, S, gbar, xbar, reg):
    return mu0 / (1 + reg) + reg / (1 + reg) * (S @ gbar + xbar)        # bam.py:67,112


def bam_update_full(samples, vs, mu0, S0, reg):
    """bam.py:31-69: S = 2 solve(I + sqrtm(I + 4 U V)^T, V^T)."""
    assert samples.ndim == 2 and vs.ndim == 2                 # :47-48
    D = samples.shape[1]
    xbar, gbar, U, V = bam_stats(samples, vs, mu0, S0, reg)
    I = np.identity(D)
    mat = I + 4 * U @ V                                       # :63
    S = 2 * np.linalg.solve(I + _real_sqrtm(mat).T, V.T)      # :65
    return _bam_mean(mu0, S, gbar, xbar, reg), S              # :67


def _lowrank_tail(Q, V, mu0, gbar, xbar, reg):
    """bam.py:105-112 for a given D x K factor Q of U."""
    K = Q.shape[1]
    I = np.identity(K)
    VT = V.T                                                  # :106
    A = VT @ Q                                                # :107
    BB = 0.5 * I + _real_sqrtm(A.T @ Q + 0.25 * I)            # :108
    BB = BB @ BB                                              # :109
    CC = np.linalg.solve(BB, A.T)                             # :110
    S = VT - A @ CC                                           # :111
    return _bam_mean(mu0, S, gbar, xbar, reg), S              # :112


def bam_lowrank_update_svds(samples, vs, mu0, S0, reg):
    """bam.py:72-114 with Q from ARPACK svds(U, k=B) as in compute_Q_host (bam.py:10-13).
    Requires B < D (ARPACK restriction)."""
    assert samples.ndim == 2 and vs.ndim == 2                 # :88-89
    B = samples.shape[0]
    xbar, gbar, U, V = bam_stats(samples, vs, mu0, S0, reg)
    UU, DD, _ = spla.svds(U, k=B)                             # :12
    Q = UU * np.sqrt(DD)                                      # :13
    return _lowrank_tail(Q, V, mu0, gbar, xbar, reg)


def exact_Q(vs, reg):
    """Exact D x (B+1) factor of U = reg G + reg/(1+reg) gbar gbar^T:
    columns sqrt(reg/B) (g_b - gbar) and sqrt(reg/(1+reg)) gbar.  U = Q Q^T exactly."""
    B = vs.shape[0]
    gbar = vs.mean(axis=0)
    return np.concatenate([np.sqrt(reg / B) * (vs - gbar).T,
                           np.sqrt(reg / (1 + reg)) * gbar[:, None]], axis=1)


def bam_lowrank_update_exact(samples, vs, mu0, S0, reg):
    """bam.py:72-114 with the svds factor replaced by the exact factor; no B < D restriction.
    This is the formulation the HIP kernels implement."""
    xbar, gbar, U, V = bam_stats(samples, vs, mu0, S0, reg)
    return _lowrank_tail(exact_Q(vs, reg), V, mu0, gbar, xbar, reg)


class Regularizers:
    """bam.py:237-274: schedules count CALLS (the ``iteration`` argument is ignored), so retries
    advance the schedule."""

    def __init__(self):
        self.counter = 0

    def reset(self):
        self.counter = 0

    def constant(self, reg0):
        def f(iteration):
            self.counter += 1
            return reg0
        return f

    def linear(self, reg0):
        def f(iteration):
            self.counter += 1
            return reg0 / self.counter
        return f

    def custom(self, func):
        def f(iteration):
            self.counter += 1
            return func(self.counter)
        return f


def bam_fit(D, lp, lp_g, key, regf, mean=None, cov=None, batch_size=2, niter=5000, nprint=10,
            verbose=False, monitor=None, retries=10, jitter=1e-6, update=bam_lowrank_update_exact,
            forced_samples=None, record=None):
    """Restatement of BaM.fit (bam.py:140-216).  The JAX threefry key split + per-iteration numpy
    re-seed (bam.py:191-192) is NOT reproduced (jax absent): one private RandomState(key) stream is
    used instead.  Everything else follows the reference: reg = regf(i) per attempt (:196), jitter
    and symmetrisation (:198-199), retry on any exception (:201-206), Cholesky accept/revert
    (:208-212), monitor cadence (:182-185,:214-215)."""
    mean = np.zeros(D) if mean is None else np.asarray(mean, dtype=np.float64)
    cov = np.identity(D) if cov is None else np.asarray(cov, dtype=np.float64)
    nevals = 1
    rs = np.random.RandomState(key)
    if nprint > niter:
        nprint = niter                                        # :177
    i = 0
    for i in range(niter + 1):
        if verbose and nprint > 0 and i % (niter // nprint) == 0:
            print(f"Iteration {i} of {niter}")
        if monitor is not None and i % monitor.checkpoint == 0:
            monitor(i, [mean, cov], lp, key, nevals=nevals)
            nevals = 0
        j = 0
        while True:
            try:
                if forced_samples is not None:
                    samples = forced_samples[i]
                else:
                    samples = svd_sampler(rs, mean, cov, batch_size)
                vs = lp_g(samples)
                nevals += batch_size
                reg = regf(i)
                mean_new, cov_new = update(samples, vs, mean, cov, reg)
                cov_new = cov_new + np.eye(D) * jitter
                cov_new = (cov_new + cov_new.T) / 2.0
                break
            except Exception as e:                            # noqa: BLE001 (reference behaviour)
                if j < retries:
                    j += 1
                    if verbose:
                        print(f"Failed with exception {e}\nTrying again {j} of {retries}")
                else:
                    raise
        good = cov_is_good(cov_new)
        if good:
            mean, cov = mean_new, cov_new
        elif verbose:
            print("Bad update for covariance matrix. Revert")
        if record is not None:
            record.append((samples, vs, reg, mean, cov, good))
    if monitor is not None:
        monitor(i, [mean, cov], lp, key, nevals=nevals)
    return mean, cov
