"""Host logic of the fit drivers (gsm-vi_amd/gsm.py, bam.py) exercised on CPU with the test-only
oracle-backed engine: monitor cadence, nevals, niter+1, revert, RNG stream, retries, schedules."""
import numpy as np
import pytest

import gsmvi_amd
from gsmvi_amd.gsm import GSM, gsm_update
from gsmvi_amd.bam import BaM, Regularizers, bam_update, bam_lowrank_update
from oracle import gsm_oracle as orc
from engines import OracleEngine
from conftest import rel_err


def test_update_functions_assert_on_shapes():
    eng = OracleEngine()
    with pytest.raises(AssertionError):
        gsm_update(np.zeros(4), np.zeros((2, 4)), np.zeros(4), np.eye(4), engine=eng)
    with pytest.raises(AssertionError):
        bam_update(np.zeros((2, 4)), np.zeros(4), np.zeros(4), np.eye(4), 1.0, engine=eng)
    with pytest.raises(AssertionError):
        bam_lowrank_update(np.zeros(4), np.zeros((2, 4)), np.zeros(4), np.eye(4), 1.0, engine=eng)


def test_update_is_pure():
    eng = OracleEngine()
    st = orc.make_update_state(6, 3, 0)
    keep = {k: st[k].copy() for k in ("samples", "vs", "mu0", "S0")}
    mu, S = gsm_update(st["samples"], st["vs"], st["mu0"], st["S0"], engine=eng)
    for k in keep:
        assert np.array_equal(keep[k], st[k])
    assert mu is not st["mu0"] and S is not st["S0"]


@pytest.mark.parametrize("D", [5, 10])
def test_fit_teacher_forced_matches_reference_trajectory(golden, D):
    g = golden(f"g2_traj_D{D}.npz")
    m, P = g["target_m"], g["target_P"]
    states = []

    class Mon:
        checkpoint = 1

        def __call__(self, i, mc, lp, key, nevals=0):
            states.append((i, mc[0].copy(), mc[1].copy(), nevals))

    gsm = GSM(D, None, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine())
    mean, cov = gsm.fit(99, niter=500, batch_size=2, verbose=False, monitor=Mon(), forced_samples=g["samples"])
    assert len(states) == 502                         # 501 checkpoints + final call (G5-like)
    for k, (i, mu_i, cov_i, _) in enumerate(states):
        assert rel_err(mu_i, g["means"][k]) < 1e-9 and rel_err(cov_i, g["covs"][k]) < 1e-9, k
    assert rel_err(mean, g["mean_fit"]) < 1e-9 and rel_err(cov, g["cov_fit"]) < 1e-9


def test_fit_svd_sampler_reproduces_reference_samples(golden):
    """sampler='svd' + rng='numpy' is the reference's sample stream: same samples at iteration 0 and
    the same converged endpoint (G3)."""
    g = golden("g2_traj_D5.npz")
    m, P = g["target_m"], g["target_P"]
    seen = []

    def lp_g(x):
        seen.append(x.copy())
        return orc.gaussian_score(x, m, P)

    mean, cov = GSM(5, None, lp_g, engine=OracleEngine()).fit(99, niter=500, verbose=False, sampler="svd")
    assert rel_err(seen[0], g["samples"][0]) < 1e-12
    assert rel_err(seen[7], g["samples"][7]) < 1e-7
    assert rel_err(mean, m) < 1e-9 and rel_err(cov, g["target_cov"]) < 1e-9


def test_fit_cholesky_sampler_converges(golden):
    g = golden("g2_traj_D10.npz")
    m, P = g["target_m"], g["target_P"]
    mean, cov = GSM(10, None, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine()).fit(
        99, niter=500, verbose=False)
    assert rel_err(mean, m) < 1e-9 and rel_err(cov, g["target_cov"]) < 1e-9


def test_monitor_cadence_and_nevals(golden):
    g = golden("g5_monitor.npz")
    m, P = g["target_m"], g["target_P"]
    calls, n = [], [0]

    class Mon:
        checkpoint = 3

        def __call__(self, i, mc, lp, key, nevals=0):
            assert isinstance(mc, list) and len(mc) == 2 and mc[0].shape == (4,) and mc[1].shape == (4, 4)
            assert lp == "LP" and key == 5
            calls.append((i, nevals))

    def lp_g(x):
        n[0] += 1
        return orc.gaussian_score(x, m, P)

    GSM(4, "LP", lp_g, engine=OracleEngine()).fit(5, niter=10, batch_size=2, verbose=False, monitor=Mon())
    assert calls == [tuple(r) for r in g["calls"].tolist()]
    assert n[0] == 11                                  # niter + 1 updates


def test_revert_keeps_mean_and_cov(golden, capsys):
    g = golden("g4_revert.npz")
    D = g["mu0"].shape[0]
    gsm = GSM(D, None, lambda x: g["vs"], engine=OracleEngine())
    mean, cov = gsm.fit(0, mean=g["mu0"], cov=g["S0"], niter=0, batch_size=2, verbose=True,
                        forced_samples=g["samples"][None])
    assert np.array_equal(mean, g["mu0"]) and np.array_equal(cov, g["S0"])
    assert gsm.n_reverts == 1
    assert "Bad update for covariance matrix. Revert" in capsys.readouterr().out


def test_nprint_guard_and_user_arrays_untouched():
    m, cov_t, P = orc.make_gaussian_target(3, 1)
    mean0, cov0 = np.ones(3), 2 * np.eye(3)
    a, b = mean0.copy(), cov0.copy()
    GSM(3, None, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine()).fit(
        1, mean=mean0, cov=cov0, niter=3, nprint=10, verbose=False)     # reference: ZeroDivisionError
    assert np.array_equal(a, mean0) and np.array_equal(b, cov0)


def test_initial_cov_must_be_pd():
    with pytest.raises(ValueError):
        GSM(2, None, lambda x: x, engine=OracleEngine()).fit(0, cov=-np.eye(2), niter=1, verbose=False)


def test_bam_fit_converges_and_schedule_counts_calls():
    D = 5
    m, cov_t, P = orc.make_gaussian_target(D, 17)
    reg = Regularizers()
    bam = BaM(D, None, lambda x: orc.gaussian_score(x, m, P), use_lowrank=True, engine=OracleEngine())
    mean, cov = bam.fit(99, regf=reg.custom(lambda i: 100 / (1 + i)), niter=100, batch_size=2, verbose=False, method="dense")
    assert reg.counter == 101 and bam.method_used == "dense"
    assert np.allclose(mean, m, atol=1e-3) and np.allclose(cov, cov_t, atol=1e-3, rtol=1e-3)


def test_bam_default_method_rule():
    """BaM.fit(method="auto"), round 6: with the reference's default jitter (bam.py:140: 1e-6, added to the covariance after every
    update, :198) the default is the reference's own loop ("dense"); the factor form is the default where it exists AND the
    call asks for no jitter, and opt-in otherwise (method="factor" or jitter_every=K: the owed shift is absorbed every K
    accepted updates)."""
    D = 6
    m, cov_t, P = orc.make_gaussian_target(D, 17)
    lp_g = lambda x: orc.gaussian_score(x, m, P)      # noqa: E731
    reg = Regularizers()

    def used(**kw):
        bam = BaM(D, None, lp_g, engine=OracleEngine())
        bam.fit(3, regf=reg.constant(5.0), niter=2, verbose=False, **kw)
        return bam.method_used

    assert used(batch_size=3) == "dense"                          # default jitter: the reference's loop, shift included
    assert used(batch_size=3, jitter=0.0) == "factor"             # 2B <= D and nothing to shift
    assert used(batch_size=3, jitter_every=4) == "factor"         # opt-in: the shift is absorbed every 4 accepted updates
    assert used(batch_size=3, jitter=1e-3) == "dense"
    assert used(batch_size=4, jitter=0.0) == "dense"              # 2B > D: the factor form does not exist
    assert used(batch_size=2, jitter=0.0, sampler="svd") == "dense"   # the reference's legacy sampler needs the covariance
    assert used(batch_size=2, jitter=0.0, forced_samples=[np.zeros((2, D))] * 3) == "dense"
    assert used(batch_size=3, jitter=0.0, method="dense") == "dense"


def test_bam_factor_fit_absorbs_the_jitter_it_owes():
    """bam.py:198 adds jitter * I to the covariance after every update.  The factor-form fit carries the owed shift and
    re-factorises F^T F + owed I every ``jitter_every`` accepted updates.  With jitter_every = 1 that IS the reference's loop:
    the factor fit's own samples forced into the dense loop give the same (mean, cov) at every checkpoint.  With a longer
    period the covariance the monitor sees still carries the full shift (what is owed is added to F^T F), and the number of
    absorptions is the number of completed periods."""
    D, B, niter, jit = 8, 3, 40, 1e-3                  # (a LARGE jitter, so that dropping it would show at 1e-3)
    m, cov_t, P = orc.make_gaussian_target(D, 29)
    lp_g = lambda x: orc.gaussian_score(x, m, P)      # noqa: E731

    class Mon:
        checkpoint = 5

        def __init__(self):
            self.s = []

        def __call__(self, i, params, lp, key, nevals=1):
            self.s.append((params[0].copy(), params[1].copy()))

    seen = []

    def rec(x):
        seen.append(np.array(x, copy=True))
        return lp_g(x)

    mf, md = Mon(), Mon()
    bam = BaM(D, None, rec, engine=OracleEngine())
    bam.fit(5, regf=Regularizers().custom(lambda i: 20 / (1 + i)), niter=niter, batch_size=B, verbose=False, method="factor",
            jitter=jit, jitter_every=1, monitor=mf)
    assert bam.method_used == "factor" and bam.n_absorbed == niter + 1 and bam.jitter_every_used == 1
    BaM(D, None, lp_g, engine=OracleEngine()).fit(5, regf=Regularizers().custom(lambda i: 20 / (1 + i)), niter=niter,
                                                  batch_size=B, verbose=False, method="dense", jitter=jit,
                                                  forced_samples=seen, monitor=md)
    assert len(mf.s) == len(md.s) == niter // 5 + 2
    for (m1, c1), (m2, c2) in zip(mf.s, md.s):
        assert rel_err(m1, m2) < 1e-9 and rel_err(c1, c2) < 1e-9
    # period 4: 41 updates = 10 absorptions + 1 owed; the returned covariance carries the owed shift
    bam4 = BaM(D, None, lp_g, engine=OracleEngine())
    _, cov4 = bam4.fit(5, regf=Regularizers().constant(5.0), niter=niter, batch_size=B, verbose=False, method="factor",
                       jitter=jit, jitter_every=4)
    bam0 = BaM(D, None, lp_g, engine=OracleEngine())
    _, cov0 = bam0.fit(5, regf=Regularizers().constant(5.0), niter=niter, batch_size=B, verbose=False, method="factor",
                       jitter=jit, jitter_every=0)
    assert bam4.n_absorbed == 10 and bam0.n_absorbed == 0 and bam0.jitter_every_used == 0
    assert np.all(np.linalg.eigvalsh(cov4) > 0)


def test_bam_factor_fit_runs_the_same_loop():
    """method="factor": state (mean, F), accept/revert decided by the update, same schedule / monitor / retry semantics."""
    D = 6
    m, cov_t, P = orc.make_gaussian_target(D, 23)
    reg = Regularizers()
    seen = []

    class Mon:
        checkpoint = 25

        def __call__(self, i, params, lp, key, nevals=1):
            seen.append((i, nevals, np.allclose(params[1], params[1].T)))

    bam = BaM(D, None, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine())
    mean, cov = bam.fit(5, regf=reg.custom(lambda i: 100 / (1 + i)), niter=100, batch_size=3, verbose=False,
                        method="factor", monitor=Mon())
    assert bam.method_used == "factor" and reg.counter == 101 and bam.n_reverts == 0
    assert [s[0] for s in seen] == [0, 25, 50, 75, 100, 100] and all(s[2] for s in seen)
    assert seen[1][1] == 25 * 3                        # nevals between checkpoints = iterations x batch
    assert np.allclose(mean, m, atol=1e-3) and np.allclose(cov, cov_t, atol=1e-3, rtol=1e-3)
    with pytest.raises(AssertionError):                # 2B <= D
        bam.fit(5, regf=reg.constant(1.0), niter=1, batch_size=4, verbose=False, method="factor")
    with pytest.raises(AssertionError):
        bam.fit(5, regf=reg.constant(1.0), niter=1, batch_size=2, verbose=False, method="factor", sampler="svd")


def test_bam_retries_then_reraises(capsys):
    D = 3
    calls = [0]

    def flaky(x):
        calls[0] += 1
        if calls[0] <= 2:
            raise FloatingPointError("bad sample")
        return -x

    reg = Regularizers()
    BaM(D, None, flaky, engine=OracleEngine()).fit(1, regf=reg.constant(1.0), niter=1, verbose=False, retries=3)
    out = capsys.readouterr().out
    assert "Trying again 2 of 3" in out and calls[0] == 4
    assert reg.counter == 2                           # regf is reached only by successful attempts

    def always(x):
        raise FloatingPointError("nope")

    with pytest.raises(FloatingPointError):
        BaM(D, None, always, engine=OracleEngine()).fit(1, regf=reg.constant(1.0), niter=1, verbose=False,
                                                        retries=2)


def test_regularizers_match_reference_semantics():
    r = Regularizers()
    lin = r.linear(100.0)
    assert [lin(0), lin(0), lin(5)] == [100.0, 50.0, 100.0 / 3]
    r.reset()
    assert r.custom(lambda i: 100 / (1 + i))(42) == 50.0


def test_fit_factor_method_converges_and_keeps_cadence(golden):
    """method='factor' (state = square factor, PD test on a 2B x 2B matrix): same driver logic -- converges
    to the target like the dense method (K3) and keeps the monitor / nevals cadence (G5)."""
    g = golden("g2_traj_D10.npz")
    m, P = g["target_m"], g["target_P"]
    calls = []

    class Mon:
        checkpoint = 100

        def __call__(self, i, mc, lp, key, nevals=0):
            assert mc[1].shape == (10, 10)
            calls.append((i, nevals))

    gsm = GSM(10, None, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine())
    mean, cov = gsm.fit(99, niter=500, batch_size=2, verbose=False, monitor=Mon(), method="factor")
    assert rel_err(mean, m) < 1e-9 and rel_err(cov, g["target_cov"]) < 1e-9 and gsm.n_reverts == 0
    assert calls == [(0, 1), (100, 200), (200, 200), (300, 200), (400, 200), (500, 200), (500, 2)]
    with pytest.raises(AssertionError):
        GSM(4, None, lambda x: x, engine=OracleEngine()).fit(0, niter=1, batch_size=4, verbose=False, method="factor")


def test_host_score_pool_never_hands_out_an_array_somebody_kept():
    """Round-5 advice (engine.py host_score): the pinned samples array given to a host lp_g is reused only when nobody outside
    the pool holds it -- a recording callable, or a retained VIEW (x[:, :k], whose .base is the array), keeps it out of reuse."""
    from gsmvi_amd.engine import _unheld_entry
    pool = [(object(), np.zeros((4, 6))) for _ in range(3)]
    assert _unheld_entry(pool) is pool[0]                     # nobody holds anything: the first entry is free
    kept = pool[0][1]                                         # a recording wrapper keeps its argument
    assert _unheld_entry(pool) is pool[1]
    view = pool[1][1][:, :2]                                  # ... another one keeps only a view of it
    assert _unheld_entry(pool) is pool[2]
    also = pool[2][1]
    assert _unheld_entry(pool) is None                        # all held: host_score allocates a fresh array
    del kept
    assert _unheld_entry(pool) is pool[0]
    del view, also
    assert _unheld_entry(pool) is pool[0]
