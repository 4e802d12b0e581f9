"""CPU oracle for the GSM hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``gsm-vi_amd/``) never imports anything under
``oracle/`` and fails loudly when the HIP library is missing.

This is a numpy restatement of the reference algorithm (float64, the reference's dtype).
Every function cites the reference lines it follows (paths relative to /root/reference).
Parity status: PINNED -- ``tests/golden/make_golden.py`` imports ``gsmvi.gsm_numpy`` from the
reference in the build container and freezes its outputs as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against those vectors.

Two formulations are provided on purpose:

* ``*_faithful``: the reference's own operation sequence, including its B x D x D temporaries
  (this is what ``bench.py`` times as the CPU baseline, kind "port");
* ``gsm_update_batched``: the algebraically equal O(B D^2) BLAS-3 form, the
  formulation the HIP kernels implement; also the "best-effort CPU" number.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "gsm_single_faithful", "gsm_update_faithful", "gsm_per_sample_terms", "gsm_update_batched",
    "gsm_factor_terms", "gsm_factor_update", "cov_is_good", "svd_sampler", "gaussian_score", "make_gaussian_target",
    "make_update_state", "gsm_fit", "philox4x32_10", "philox_randn",
]


# --------------------------------------------------------------------------------------
# a1: one (sample, score) pair          reference: gsmvi/gsm_numpy.py:4-24
# --------------------------------------------------------------------------------------
def gsm_single_faithful(x, g, mu0, S0):
    """Per-sample GSM increments (dmu, dS), same operation order as gsm_numpy.py:7-23."""
    d = mu0 - x                                             # (mu0 - sample), :9
    Sg = S0 @ g                                             # :7
    gSg = g @ Sg                                            # :8
    mv = d @ g                                              # :9
    rho = 0.5 * np.sqrt(1 + 4 * (gSg + mv ** 2)) - 0.5      # :10
    eps0 = Sg - mu0 + x                                     # :11
    den = 1 + rho + mv                                      # :15
    proj = np.eye(x.shape[0]) - np.outer(d, g) / den        # :14,:16 (D x D temporaries)
    dmu = 1 / (1 + rho) * (proj @ eps0)                     # :17
    mu = mu0 + dmu                                          # :18
    dS = np.outer(d, d) - np.outer(mu - x, mu - x)          # :21-23
    return dmu, dS


# --------------------------------------------------------------------------------------
# a2: batch update                      reference: gsmvi/gsm_numpy.py:27-55
# --------------------------------------------------------------------------------------
def gsm_update_faithful(samples, vs, mu0, S0):
    """mean over the batch of per-sample increments; allocates (B,D,D) like gsm_numpy.py:47."""
    assert samples.ndim == 2 and vs.ndim == 2               # :43-44
    B, D = samples.shape
    dmus = np.zeros((B, D))
    dSs = np.zeros((B, D, D))                               # :47
    for b in range(B):                                      # :48-49
        dmus[b], dSs[b] = gsm_single_faithful(samples[b], vs[b], mu0, S0)
    return mu0 + dmus.mean(axis=0), S0 + dSs.mean(axis=0)   # :50-53


def gsm_per_sample_terms(samples, vs, mu0, S0):
    """The per-sample scalars and factor rows of SURVEY A.1 (derived from gsm_numpy.py:7-23).

    Returns dict with SG (B,D), gSg, mv, rho, den (B,), dmu (B,D) = per-sample mean increment,
    dvec = mu0 - x (B,D) and evec = dvec + dmu (B,D); then dS_b = dvec dvec^T - evec evec^T.
    """
    X = np.asarray(samples, dtype=np.float64)
    G = np.asarray(vs, dtype=np.float64)
    dvec = mu0[None, :] - X
    SG = G @ S0.T                                           # row b = S0 @ g_b   (:7)
    gSg = np.einsum("bi,bi->b", G, SG)                      # :8
    mv = np.einsum("bi,bi->b", dvec, G)                     # :9
    rho = 0.5 * np.sqrt(1 + 4 * (gSg + mv ** 2)) - 0.5      # :10
    den = 1 + rho + mv                                      # :15
    eps = SG - dvec                                         # :11
    # (I - d g^T/den) eps = eps - d (g.eps)/den with g.eps = gSg - mv    (:14-17)
    dmu = (eps - dvec * ((gSg - mv) / den)[:, None]) / (1 + rho)[:, None]
    evec = dvec + dmu                                       # mu - x, :18,:22
    return dict(SG=SG, gSg=gSg, mv=mv, rho=rho, den=den, dmu=dmu, dvec=dvec, evec=evec)


def gsm_update_batched(samples, vs, mu0, S0):
    """O(B D^2) form of gsm_numpy.py:27-55: two skinny GEMMs, no (B,D,D) temporary."""
    assert samples.ndim == 2 and vs.ndim == 2
    B = samples.shape[0]
    t = gsm_per_sample_terms(samples, vs, mu0, S0)
    mu = mu0 + t["dmu"].mean(axis=0)
    S = S0 + (t["dvec"].T @ t["dvec"] - t["evec"].T @ t["evec"]) / B
    return mu, S


# --------------------------------------------------------------------------------------
# factor ("Cholesky-factor") form of the same update: Sigma = F F^T, x = mu + F z
# derived from gsm_numpy.py:4-55; there is no reference implementation.
# --------------------------------------------------------------------------------------
def gsm_factor_terms(Z, G, F):
    """Whitened per-sample quantities. Z (B,D) whitened draws, G (B,D) scores at x=mu+F z."""
    W = G @ F                                               # w_b = F^T g_b
    ww = np.einsum("bi,bi->b", W, W)                        # g S g
    zw = np.einsum("bi,bi->b", Z, W)                        # = -mv
    rho = 0.5 * np.sqrt(1 + 4 * (ww + zw ** 2)) - 0.5
    den = 1 + rho - zw
    U = ((W + Z) + Z * ((ww + zw) / den)[:, None]) / (1 + rho)[:, None]   # dmu_b = F u_b
    Y = U - Z                                               # e_b = F y_b ; d_b = -F z_b
    return dict(W=W, ww=ww, zw=zw, rho=rho, den=den, U=U, Y=Y)


def gsm_factor_update(Z, G, mu0, F):
    """Returns (mu, F', ok) with F' F'^T = Sigma' of gsm_update for samples mu0 + Z F^T.

    Sigma' = F M F^T with M = I + (Z^T Z - Y^T Y)/B (D x D, identity plus rank <= 2B).  Any F' = F C
    with C C^T = M is a valid factor (the device kernel uses a different C than this oracle), so
    parity is asserted on (mu, F' F'^T), never on F' itself.  ok = M positive definite, which is
    equivalent to the reference's Cholesky test on Sigma' (gsm_numpy.py:132-146) for nonsingular F.
    """
    B, D = Z.shape
    t = gsm_factor_terms(Z, G, F)
    M = np.eye(D) + (Z.T @ Z - t["Y"].T @ t["Y"]) / B
    mu = mu0 + F @ t["U"].mean(axis=0)
    try:
        C = np.linalg.cholesky(0.5 * (M + M.T))
        ok = not bool(np.isnan(C).any())
    except np.linalg.LinAlgError:
        return mu, F.copy(), False
    return mu, F @ C, ok


# --------------------------------------------------------------------------------------
# a5: accept / revert test              reference: gsmvi/gsm_numpy.py:132-146
# --------------------------------------------------------------------------------------
def cov_is_good(cov):
    """True iff np.linalg.cholesky succeeds and has no NaN (the reference's NaN branch raises a
    NameError that its bare ``except`` swallows, i.e. NaN => False; gsm_numpy.py:139-146)."""
    try:
        L = np.linalg.cholesky(cov)
    except Exception:
        return False
    return not bool(np.isnan(L).any())


# --------------------------------------------------------------------------------------
# a3: sampler                           reference: gsmvi/gsm_numpy.py:105,116
# --------------------------------------------------------------------------------------
def svd_sampler(rs, mean, cov, size):
    """Restates legacy ``RandomState.multivariate_normal(mean, cov, size)`` (check_valid='warn'
    without the warning): z ~ N(0,I) (size, D) from the MT19937 stream, (u,s,vt) = svd(cov),
    x = mean + z @ (sqrt(s)[:,None] * vt).  Bit-exact with numpy on the same LAPACK build."""
    D = mean.shape[0]
    z = rs.standard_normal((size, D)).reshape(-1, D)
    _, s, vt = np.linalg.svd(cov)
    return mean + z @ (np.sqrt(s)[:, None] * vt)


# --------------------------------------------------------------------------------------
# a4/a7: the synthetic Gaussian target  reference: examples/example_gsm_numpy.py:8-31
# --------------------------------------------------------------------------------------
def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon, Moraes, Dror & Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11;
    the Random123 library).  ``counter``: (n, 4) uint32, ``key``: (2,) uint32 -> (n, 4) uint32.  Not part of
    the reference (its z-stream is numpy's MT19937, gsm_numpy.py:105; the JAX twins use threefry,
    gsm.py:117): this restates the device generator of csrc/gsmvi_rng.hip and is pinned to the Random123
    known-answer vectors in tests/test_oracle_golden.py."""
    c = np.array(counter, dtype=np.uint64).reshape(-1, 4).copy()
    k0, k1 = (np.uint64(int(key[0])), np.uint64(int(key[1])))
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    sh = np.uint64(32)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        c = np.stack([(p1 >> sh) ^ c[:, 1] ^ k0, p1 & mask, (p0 >> sh) ^ c[:, 3] ^ k1, p0 & mask], axis=1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c.astype(np.uint32)


def philox_randn(seed, call, n, return_raw=False):
    """n standard normals of the device stream (csrc/gsmvi_rng.hip): element pair p of ``call`` is the Philox
    block with counter (p lo, p hi, call lo, call hi) and key (seed lo, seed hi); two 53-bit uniforms
    ((a >> 5) 2^26 + (b >> 6) + 1/2) 2^-53 and Box-Muller give z[2p], z[2p+1]."""
    seed, call, n = int(seed) & (2 ** 64 - 1), int(call), int(n)
    p = np.arange((n + 1) // 2, dtype=np.uint64)
    lo = np.uint64(0xFFFFFFFF)
    ctr = np.stack([p & lo, p >> np.uint64(32), np.full_like(p, call & 0xFFFFFFFF), np.full_like(p, call >> 32)], axis=1)
    w = philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32)).astype(np.uint64)
    u1 = (((w[:, 0] >> np.uint64(5)) << np.uint64(26)) | (w[:, 1] >> np.uint64(6))).astype(np.float64)
    u2 = (((w[:, 2] >> np.uint64(5)) << np.uint64(26)) | (w[:, 3] >> np.uint64(6))).astype(np.float64)
    u1 = (u1 + 0.5) / 9007199254740992.0
    u2 = (u2 + 0.5) / 9007199254740992.0
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.stack([r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)], axis=1).reshape(-1)[:n]
    return (z, w.astype(np.uint32)) if return_raw else z


def make_gaussian_target(D, seed, cond=None):
    """Seeded version of example_gsm_numpy.py:11-14: m = U(0,1)^D, Sig_t = L L^T + 1e-3 I.
    With ``cond`` the spectrum is rescaled log-uniformly to that condition number (config 5)."""
    rs = np.random.RandomState(seed)
    m = rs.random_sample(D)
    L = rs.normal(size=D * D).reshape(D, D)
    cov = L @ L.T + 1e-3 * np.eye(D)
    if cond is not None:
        w, Q = np.linalg.eigh(cov)
        w = np.logspace(0.0, np.log10(cond), D) * (w.min() if w.min() > 0 else 1.0)
        cov = (Q * w[None, :]) @ Q.T
        cov = 0.5 * (cov + cov.T)
    return m, cov, np.linalg.inv(cov)


def gaussian_score(X, m, P):
    """Score of N(m, P^-1) at rows of X: g_b = -P (x_b - m)   (example_gsm_numpy.py:24-29)."""
    return -(X - m[None, :]) @ P.T


def gaussian_logp(X, m, P):
    """Sum over rows of -1/2 (m-x)^T P (m-x)   (example_gsm_numpy.py:17-22)."""
    R = m[None, :] - X
    return float(-0.5 * np.einsum("bi,ij,bj->", R, P, R))


def make_update_state(D, B, seed, target_seed=None):
    """Synthetic inputs for one update, SURVEY section 8(d): mu0 ~ N(0,I), S0 = A A^T/D + 0.1 I,
    samples = mu0 + z chol(S0)^T, scores from the seeded Gaussian target."""
    rs = np.random.RandomState(1000 + seed)
    m, cov_t, P = make_gaussian_target(D, seed if target_seed is None else target_seed)
    mu0 = rs.standard_normal(D)
    A = rs.standard_normal((D, D))
    S0 = A @ A.T / D + 0.1 * np.eye(D)
    S0 = 0.5 * (S0 + S0.T)
    Lc = np.linalg.cholesky(S0)
    Z = rs.standard_normal((B, D))
    X = mu0[None, :] + Z @ Lc.T
    G = gaussian_score(X, m, P)
    return dict(samples=X, vs=G, mu0=mu0, S0=S0, Z=Z, L=Lc, m=m, P=P, cov_t=cov_t)


# --------------------------------------------------------------------------------------
# a6: fit driver                        reference: gsmvi/gsm_numpy.py:77-129
# --------------------------------------------------------------------------------------
def gsm_fit(D, lp, lp_g, key, mean=None, cov=None, batch_size=2, niter=5000, nprint=10,
            verbose=False, monitor=None, update=gsm_update_faithful, forced_samples=None,
            record=None):
    """Restatement of GSM.fit with a private RandomState(key) (same stream as np.random.seed(key),
    gsm_numpy.py:105).  niter+1 iterations (:106); monitor cadence and nevals bookkeeping (:103,
    :110-113,:119,:127-128); revert both mean and cov on a bad covariance (:121-125).
    ``forced_samples`` (niter+1,B,D) replaces the sampler (teacher forcing); ``record`` (a list)
    receives (samples, vs, mean, cov, accepted) per iteration."""
    mean = np.zeros(D) if mean is None else np.asarray(mean, dtype=np.float64)
    cov = np.identity(D) if cov is None else np.asarray(cov, dtype=np.float64)
    nevals = 1
    rs = np.random.RandomState(key)
    nprint = max(1, min(nprint, niter)) if niter > 0 else 1   # BaM's guard (bam.py:177)
    i = 0
    for i in range(niter + 1):
        if verbose and niter > 0 and i % (niter // nprint) == 0:
            print(f"Iteration {i} of {niter}")
        if monitor is not None and i % monitor.checkpoint == 0:
            monitor(i, [mean, cov], lp, key, nevals=nevals)
            nevals = 0
        if forced_samples is not None:
            samples = forced_samples[i]
        else:
            samples = svd_sampler(rs, mean, cov, batch_size)
        vs = lp_g(samples)
        mean_new, cov_new = update(samples, vs, mean, cov)
        nevals += batch_size
        good = cov_is_good(cov_new)
        if good:
            mean, cov = mean_new, cov_new
        elif verbose:
            print("Bad update for covariance matrix. Revert")
        if record is not None:
            record.append((samples, vs, mean, cov, good))
    if monitor is not None:
        monitor(i, [mean, cov], lp, key, nevals=nevals)
    return mean, cov
