"""The smallest shapes (the reference's own example is D = 5, B = 2: examples/example_gsm_numpy.py:38; nothing in
gsm_numpy.py:27-55 or bam.py:31-69 bounds D or B from below): D = 1 .. 17 with B = 1 .. 5 through the Python drop-ins
(odd D via the inert coordinate, B = 1 where BaM's centred statistics vanish), against the oracle; and whole fits of both
classes at D <= 5 with their default methods."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _o():
    from oracle import gsm_oracle as orc
    from oracle import bam_oracle as borc
    return orc, borc


@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 7, 16, 17])
def test_one_shot_updates_at_the_smallest_shapes(D):
    import gsmvi_amd
    orc, borc = _o()
    for B in (1, 2, 3, 5):
        rs = np.random.RandomState(D * 10 + B)
        A = rs.standard_normal((D, D))
        S0 = A @ A.T / D + 0.5 * np.eye(D)
        mu0 = rs.standard_normal(D)
        X = mu0 + rs.standard_normal((B, D)) @ np.linalg.cholesky(S0).T
        G = orc.gaussian_score(X, rs.standard_normal(D), 2.0 * np.eye(D))
        mu, S = gsmvi_amd.gsm_update(X, G, mu0, S0)
        mo, So = orc.gsm_update_faithful(X, G, mu0, S0)
        assert max(np.abs(mu - mo).max(), np.abs(S - So).max()) < 1e-12, (D, B)
        mb, Sb = gsmvi_amd.bam_update(X, G, mu0, S0, 1.5)
        mbo, Sbo = borc.bam_update_full(X, G, mu0, S0, 1.5)
        assert max(np.abs(mb - mbo).max(), np.abs(Sb - Sbo).max()) < 1e-10, (D, B)


@pytest.mark.parametrize("D,B", [(1, 1), (2, 1), (2, 2), (3, 2), (5, 2), (5, 4)])
def test_fits_at_the_smallest_shapes(D, B):
    """GSM reaches the Gaussian target to rounding; BaM's dense loop (chosen where 2B > D) stops at its jitter floor
    (bam.py:198: + 1e-6 I per iteration), its factor form (2B <= D) goes to 1e-10."""
    import gsmvi_amd
    orc, _ = _o()
    m, cov, P = orc.make_gaussian_target(D, 1)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    g = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
    mg, cg = g.fit(3, niter=300, batch_size=B, verbose=False)
    assert np.abs(mg - m).max() < 1e-12 and np.abs(cg - cov).max() < 1e-11 * max(1.0, np.abs(cov).max()), g.method_used
    for jit in (1e-6, 0.0):                      # round 6: the reference's default jitter -> its own (dense) loop; jitter 0 -> factor form
        b = gsmvi_amd.BaM(D, tgt.lp, tgt.lp_g)
        mb, cb = b.fit(3, lambda i: 10.0 / (1 + i), niter=300, batch_size=B, verbose=False, jitter=jit)
        assert b.method_used == ("factor" if (2 * B <= D and jit == 0.0) else "dense")
        tol = 1e-9 if b.method_used == "factor" else 1e-4
        assert np.abs(mb - m).max() < tol and np.abs(cb - cov).max() < tol * max(1.0, np.abs(cov).max())
