"""BaM fit driver, update functions and regularizer schedules with the reference's signatures
(gsmvi/bam.py).  The update runs in HIP kernels through the engine; see csrc/gsmvi_bam.hip."""
import numpy as np
import torch

from .engine import get_engine
from .gsm import _legacy_mvn, _is_torch, _host_draw


def bam_lowrank_update(samples, vs, mu0, S0, reg, engine=None, jitter=0.0):
    """Drop-in for ``bam_lowrank_update(samples, vs, mu0, S0, reg)`` (gsmvi/bam.py:72-114).

    The D x B ARPACK factor of U (bam.py:10-13,104) is replaced by an exact rank-B factor of U (Helmert recombination of the centred score rows), so
    there is no B < D restriction; the returned S is symmetrised (the reference symmetrises in
    ``fit``, bam.py:199)."""
    assert len(samples.shape) == 2
    assert len(vs.shape) == 2
    eng = engine if engine is not None else get_engine()
    want_torch = _is_torch(samples)
    Xd, Gd, m0, S0d = eng.asarray(samples), eng.asarray(vs), eng.asarray(mu0), eng.asarray(S0)
    D = int(m0.shape[0])
    if D % 2 == 1 and getattr(eng, "name", "") == "hip":
        # odd D: the (D + 1)-dimensional problem with an inert last coordinate runs on the tuned kernels (_oddpad.py)
        from . import _oddpad
        mu, S, _ = eng.bam_update(_oddpad.pad_rows(eng, Xd, D), _oddpad.pad_rows(eng, Gd, D), _oddpad.pad_vec(eng, m0, D),
                                  _oddpad.pad_mat(eng, S0d, D), float(reg), float(jitter))
        mu, S = mu[:D].contiguous(), S[:D, :D].contiguous()
    else:
        mu, S, _ = eng.bam_update(Xd, Gd, m0, S0d, float(reg), float(jitter))
    return (mu, S) if want_torch else (eng.to_numpy(mu), eng.to_numpy(S))


def bam_update(samples, vs, mu0, S0, reg, engine=None, jitter=0.0):
    """Drop-in for ``bam_update(samples, vs, mu0, S0, reg)`` (gsmvi/bam.py:31-69).  The full-rank
    formula (D x D sqrtm + solve) equals the low-rank one to round-off, so both names
    run the same kernels."""
    return bam_lowrank_update(samples, vs, mu0, S0, reg, engine=engine, jitter=jitter)


class Regularizers:
    """Regularizer schedules for BaM (gsmvi/bam.py:237-274).  As in the reference the schedules
    count CALLS and ignore the ``iteration`` argument, so retries advance them."""

    def __init__(self):
        self.counter = 0

    def reset(self):
        self.counter = 0

    def constant(self, reg0):
        def reg_iter(iteration):
            self.counter += 1
            return reg0
        return reg_iter

    def linear(self, reg0):
        def reg_iter(iteration):
            self.counter += 1
            return reg0 / self.counter
        return reg_iter

    def custom(self, func):
        def reg_iter(iteration):
            self.counter += 1
            return func(self.counter)
        return reg_iter


class BaM:
    """Wrapper class for using BaM updates to fit a distribution (gsmvi/bam.py:117-137).
    ``use_lowrank`` and ``jit_compile`` are accepted for signature compatibility; both update
    forms run the same HIP kernels and there is nothing to jit."""

    def __init__(self, D, lp, lp_g, use_lowrank=False, jit_compile=True, engine=None):
        self.D = D
        self.lp = lp
        self.lp_g = lp_g
        self.use_lowrank = use_lowrank
        self.jit_compile = jit_compile
        self._engine = engine

    def fit(self, key, regf, mean=None, cov=None, batch_size=2, niter=5000, nprint=10, verbose=True,
            check_goodness=True, monitor=None, retries=10, jitter=1e-6, *, sampler="cholesky", rng="auto",
            as_torch=False, forced_samples=None, shard=False, group=None, check_update_flag=False, method="auto",
            root_potrf=False, graph=None, jitter_every=None, _zero_cols_from=None):
        """gsmvi/bam.py:140-216.  Kept: niter+1 iterations (:178); nprint clamp (:177); reg = regf(i)
        per attempt (:196); jitter on the diagonal and symmetrisation (:198-199, done in-kernel);
        retry on any exception up to ``retries`` then re-raise (:189-206); Cholesky accept/revert of
        both mean and cov (:208-212); monitor cadence (:182-185,:214-215).
        Deviation: the JAX threefry key split + per-iteration numpy re-seed (:191-192) is replaced by
        one private RandomState(key) stream (JAX is not a dependency).
        ``shard=True`` (BASELINE config 4: B=128 sharded 16 per GPU; one process per GPU, torch.distributed
        initialised): every rank draws the same Z, samples and scores only its batch_size/world rows, the
        (x_b, g_b) rows are all-gathered (dist.sharded_bam_update) and every replica runs the identical update.
        Retries when sharded are COLLECTIVE: after the score call every rank contributes a fail bit to one
        all-reduce (MAX), so either all ranks retry -- redrawing and advancing the regulariser together -- or
        none does; a rank never re-enters a collective its peers have left (one host synchronisation per
        iteration, small beside the Cholesky of the accept test).
        Deviation (documented): the reference retries on ANY exception inside sample/score/update (:189-206),
        which on its JAX path includes numerical failures surfacing as exceptions from the host callbacks.  The
        device update never raises: a numerical failure poisons the outputs with NaN, the Cholesky accept test
        rejects them and the iteration is a revert (counted in ``n_reverts``), not a retry.
        ``check_update_flag=True`` restores the retry: the update's device flag is read every iteration (one host
        synchronisation) and a non-zero flag raises FloatingPointError into the retry loop.
        ``root_potrf`` (sharded dense fit only; default False): True lets rank 0 run the accept test's Cholesky and broadcast
        the factor and its flag (dist.root_potrf) instead of every rank factoring its identical replica -- saves the
        redundant D^3 work, costs a serial factorisation + an 8 D^2-byte broadcast per iteration; opt-in until measured on
        more than one GPU (see GSM.fit).
        ``method="factor"`` with ``shard=True``: the (x_b, g_b) rows are all-gathered as in the dense form and every replica
        runs the identical factor-form update (dist.sharded_bam_factor_update); retries are collective in the same way.
        ``graph=True``: factor-form fits replay blocks of 16 iterations as one hipGraph (the regulariser table on the device:
        engine.bam_reg_source); bit-identical to the eager loop and, measured, not faster on an idle host -- off by default.
        ``method="auto"`` (round 6): "dense" -- the reference's own loop: update, + jitter * I, symmetrise, Cholesky accept test
        (:189-212) -- whenever ``jitter`` > 0, i.e. at the reference's default arguments (1e-10 from the restated reference loop on the
        same samples: tests/test_gpu_bam.py::test_default_fit_is_the_reference_loop); "factor" for ``jitter`` = 0 where that form
        exists (below), or when ``jitter_every`` is given.  Round 5 took the factor form up to jitter = 1e-6 and dropped the shift:
        2e-5 .. 3e-5 (B = 128) and 1.7e-4 (B = 32) of max|cov| from the reference loop at D = 1024, above the 1e-5 bar.
        ``method="factor"`` (needs 2*batch_size <= min(D, 256), sampler="cholesky", no forced samples):
        the state is (mean, F) with cov = F^T F; every iteration samples with F itself and applies the factor-form BaM
        update (engine.bam_factor_update) -- four passes over F, no D x D covariance, no D^3 Cholesky for the accept
        test (the update's own 2B x 2B positive-definiteness test decides accept/revert, counted in ``n_reverts``).
        The covariance is formed once, for the return value (and for each monitor call).  ``jitter``: a diagonal shift is not a
        low-rank change of a factor, so the fit carries the shift it OWES -- jitter x the accepted updates since the last
        absorption, counted on the device -- adds it to every covariance it hands out, and every ``jitter_every`` (default
        BaM.JITTER_EVERY = 4; 0 = never, the round-5 behaviour) accepted updates absorbs it: F <- chol(F^T F + owed I)
        (gsmvi_gram_shift_f64 + gsmvi_potrf_f64).  Deferring is not free -- the distance to the reference's loop grows in
        proportion to the period (4: 7e-6 at B = 128, 1.8e-5 at B = 32; 16: no better than dropping the jitter;
        profiles/r06/jitter_period.json) -- which is why this form is opt-in whenever jitter > 0.  Everything else -- niter+1
        iterations, reg = regf(i) per attempt, retries, monitor cadence -- is the loop above.
        ``method="dense"`` takes any batch size up to 1024 (the device chain's bound; tuned to 128, a blocked MFMA path above)."""
        eng = self._engine if self._engine is not None else get_engine()
        D, B = self.D, int(batch_size)
        assert method in ("auto", "dense", "factor"), "method must be 'auto', 'dense' or 'factor'"
        from . import _oddpad
        if _zero_cols_from is None and _oddpad.applies(eng, D, sampler, forced_samples):
            # odd D: the (D + 1)-dimensional problem with an inert last coordinate runs on the tuned kernels (_oddpad.py).
            # (the corner of cov' picks up the jitter like every diagonal entry; it touches nothing else)
            if method == "auto":
                method = self._auto_method(D, B, sampler, forced_samples, jitter, jitter_every)
            inner = BaM(D + 1, self.lp, _oddpad.wrap_score(eng, self.lp_g, D), use_lowrank=self.use_lowrank,
                        jit_compile=self.jit_compile, engine=eng)
            mp, cp = inner.fit(key, regf, mean=_oddpad.pad_vec(eng, mean, D), cov=_oddpad.pad_mat(eng, cov, D),
                               batch_size=batch_size, niter=niter, nprint=nprint, verbose=verbose,
                               check_goodness=check_goodness, monitor=_oddpad.wrap_monitor(monitor, self.lp, D),
                               retries=retries, jitter=jitter, sampler=sampler, rng=rng, as_torch=True, shard=shard,
                               group=group, check_update_flag=check_update_flag, method=method, root_potrf=root_potrf,
                               graph=graph, jitter_every=jitter_every, _zero_cols_from=D)
            self.method_used, self.n_reverts, self.padded_dim = inner.method_used, inner.n_reverts, D + 1
            self.graph_replays, self.graph_fallback = getattr(inner, "graph_replays", 0), getattr(inner, "graph_fallback", None)
            mean_o, cov_o = mp[:D].contiguous(), cp[:D, :D].contiguous()
            return (mean_o, cov_o) if as_torch else (eng.to_numpy(mean_o), eng.to_numpy(cov_o))
        self._zc = _zero_cols_from
        if method == "auto":
            method = self._auto_method(D, B, sampler, forced_samples, jitter, jitter_every)
        self.method_used = method
        if method == "factor":
            assert sampler == "cholesky" and forced_samples is None, \
                "method='factor' samples with its own factor (sampler='cholesky', no forced samples)"
            return self._fit_factor(eng, key, regf, mean, cov, B, niter, nprint, verbose, monitor, retries, rng, as_torch,
                                    check_update_flag, shard, group, graph, float(jitter),
                                    self.JITTER_EVERY if jitter_every is None else int(jitter_every))
        bmax = getattr(eng, "bam_max_batch", None)
        if bmax is not None and B > bmax:               # deterministic: raised here, not inside the retry loop
            raise ValueError(f"BaM.fit: batch_size {B} exceeds the device update's limit of {bmax}")
        mean_t = eng.zeros(D) if mean is None else eng.clone(mean).reshape(D)
        cov_t = eng.eye(D) if cov is None else eng.clone(cov).reshape(D, D)
        seed = int(np.asarray(key.cpu() if _is_torch(key) else key).flatten()[-1])
        rs = np.random.RandomState(seed)
        assert rng in ("auto", "numpy", "device"), "rng must be 'auto', 'numpy' or 'device'"
        dev_rng = rng == "device" or (rng == "auto" and sampler == "cholesky")   # as GSM.fit: see its docstring
        KB = 16                         # the device draws come a block of KB per launch (the stream does not depend on the state)
        Zblk = eng.empty(KB, B, D) if dev_rng else None
        ndraw = 0                       # counter-based stream: one `call` per draw, retries included
        native = bool(getattr(self.lp_g, "device_native", False))
        mon_native = bool(getattr(monitor, "device_native", False)) if monitor is not None else False

        lo, hi = 0, B
        world = 1
        if shard:
            assert sampler == "cholesky" and forced_samples is None, "shard=True needs the replicated z-stream"
            from .dist import sharded_bam_update, shard_bounds
            import torch.distributed as _dist
            world = _dist.get_world_size(group) if _dist.is_initialized() else 1
            rank = _dist.get_rank(group) if _dist.is_initialized() else 0
            lo, hi = shard_bounds(B, world, rank)
        mean_new, cov_new = eng.empty(D), eng.empty(D, D)
        R, R_new = eng.empty(D, D), eng.empty(D, D)
        Xbuf = eng.empty(hi - lo, D)
        flag, uflag, n_rev = eng.new_flag(), eng.new_flag(), eng.new_flag()
        use_factor = sampler == "cholesky" and forced_samples is None
        if use_factor:                      # the sampling factor of the initial covariance
            eng.potrf(cov_t, out=R, flag=flag)
            if eng.read_flag(flag) != 0:
                raise ValueError("initial covariance is not positive definite")

        nevals = 1
        if nprint > niter:
            nprint = niter
        every = max(1, niter // nprint) if nprint > 0 else 1
        reverts_seen = 0
        i = 0
        for i in range(niter + 1):
            if verbose and i % every == 0:
                print(f"Iteration {i} of {niter}")
                r = eng.read_flag(n_rev)
                if r > reverts_seen:
                    print(f"Bad update for covariance matrix. Revert ({r - reverts_seen} since last print)")
                    reverts_seen = r
            if monitor is not None and i % monitor.checkpoint == 0:
                mc = [mean_t, cov_t] if mon_native else [eng.to_numpy(mean_t).copy(), eng.to_numpy(cov_t).copy()]
                monitor(i, mc, self.lp, key, nevals=nevals)
                nevals = 0
            j = 0
            while True:
                try:
                    if forced_samples is not None:
                        X = eng.asarray(forced_samples[i])
                    elif sampler == "svd":
                        X = eng.asarray(_legacy_mvn(rs, eng.to_numpy(mean_t), eng.to_numpy(cov_t), B))
                    else:
                        if dev_rng:
                            if ndraw % KB == 0:
                                eng.normal_batch(KB, B, D, seed, ndraw, out=Zblk)
                                if self._zc is not None:
                                    Zblk[:, :, self._zc:] = 0.0          # inert coordinates of an odd-D fit (_oddpad.py)
                            Z = Zblk[ndraw % KB]
                            ndraw += 1
                        else:
                            Z = eng.normal_from_host(_host_draw(rs, B, D, self._zc))
                            if self._zc is not None:
                                Z[:, self._zc:] = 0.0
                        X = eng.sample(Z[lo:hi], mean_t, R, out=Xbuf)     # only this rank's rows when sharded
                    err = None
                    try:
                        vs = self.lp_g(X) if native else eng.host_score(self.lp_g, X)
                    except Exception as e_score:            # noqa: BLE001
                        if not (shard and world > 1):
                            raise
                        err, vs = e_score, None
                    if shard and world > 1:                 # agree on failure BEFORE anybody enters the gather
                        import torch
                        fb = torch.tensor([0 if err is None else 1], dtype=torch.int32,
                                          device=X.device if _is_torch(X) else "cpu")
                        _dist.all_reduce(fb, op=_dist.ReduceOp.MAX, group=group)
                        if int(fb.item()) != 0:             # nobody has called regf yet (bam.py:194-196 order)
                            raise err if err is not None else RuntimeError("score evaluation failed on another rank")
                    nevals += B
                    reg = regf(i)
                    if shard:
                        sharded_bam_update(eng, X, vs, mean_t, cov_t, reg, jitter, group=group,
                                           out=(mean_new, cov_new), flag=uflag)
                    else:
                        eng.bam_update(X, vs, mean_t, cov_t, reg, jitter, out=(mean_new, cov_new), flag=uflag)
                    if check_update_flag and self._flag_raised(eng, uflag, shard and world > 1, group):
                        raise FloatingPointError("BaM update flagged a numerical failure (device flag != 0)")
                    break
                except Exception as e:                      # noqa: BLE001 -- reference behaviour
                    if j < retries:
                        j += 1
                        print(f"Failed with exception {e}")
                        print(f"Trying again {j} of {retries}")
                    else:
                        raise e
            if shard and root_potrf and world > 1:          # opt-in: one rank factors, the others receive
                from .dist import root_potrf as _root_potrf
                _root_potrf(eng, cov_new, R_new, flag, group=group)
            else:
                eng.potrf(cov_new, out=R_new, flag=flag)
            eng.commit(flag, mean_new, cov_new, mean_t, cov_t, n_rev)
            if use_factor:
                eng.commit(flag, mean_new, R_new, mean_t, R, None)

        if monitor is not None:
            mc = [mean_t, cov_t] if mon_native else [eng.to_numpy(mean_t).copy(), eng.to_numpy(cov_t).copy()]
            monitor(i, mc, self.lp, key, nevals=nevals)
        self.n_reverts = eng.read_flag(n_rev)
        if as_torch:
            return mean_t, cov_t
        return eng.to_numpy(mean_t), eng.to_numpy(cov_t)

    # How often the factor form absorbs the jitter it owes (bam.py:198: cov_new += jitter * I after EVERY update).  A diagonal
    # shift is not a low-rank change of a square factor: the fit carries the owed shift and every JITTER_EVERY accepted updates
    # re-factorises F^T F + owed * I (gsmvi_gram_shift_f64 + gsmvi_potrf_f64).  Deferring is NOT free: the update responds to the
    # shift of its input covariance with an amplification of ~sqrt(cond Sigma), so the distance to the reference's loop on the
    # same samples grows with the period -- measured on the c4-like target of tests/test_gpu_bam.py (profiles/r06/
    # jitter_period.json): see the table in DESIGN.md section 8.2.  0 = never absorb (the round-5 behaviour: jitter ignored).
    JITTER_EVERY = 4

    @staticmethod
    def _auto_method(D, B, sampler, forced_samples, jitter, jitter_every):
        """method="auto" (round 6).  The factor form wherever it exists (2B <= min(D, 256), the device Cholesky sampler, no
        teacher-forced samples) AND the call asks for no jitter; with the reference's default jitter = 1e-6 (bam.py:140,198)
        the default is the reference's own loop -- "dense": the same update + the same shift + the same Cholesky accept test,
        1e-10 from the reference restatement on the same samples.  Round 5 took the factor form up to jitter = 1e-6 and
        dropped the shift; that moved the default 2e-5 .. 3e-5 of max|cov| away from the reference loop (above the 1e-5 bar of
        BASELINE.json), and absorbing the shift every K iterations only brings that down in proportion to K (K = 16: no better
        than dropping it; K = 4: ~4e-6 at (1024, 128) at the cost of a D^3 factorisation every fourth iteration).  The fast form
        therefore is OPT-IN for calls with jitter: method="factor" (absorbs every ``jitter_every`` = 4 accepted updates), or
        jitter=0."""
        exists = sampler == "cholesky" and forced_samples is None and 2 * B <= min(D, 256)
        if not exists:
            return "dense"
        if float(jitter) == 0.0:
            return "factor"
        return "factor" if (jitter_every is not None and int(jitter_every) > 0) else "dense"

    @staticmethod
    def _flag_raised(eng, flag, collective, group):
        """check_update_flag: is the update's device flag set -- on ANY rank when sharded.  The replicas run the identical
        update on identical inputs, so their flags agree; the all-reduce (MAX) makes the retry decision collective anyway
        (advisor, round 4): a rank that retried alone would pair its next all-gather with its peers' NEXT iteration."""
        bad = eng.read_flag(flag) != 0
        if collective:
            import torch.distributed as _dist
            fb = torch.tensor([1 if bad else 0], dtype=torch.int32, device=flag.device if _is_torch(flag) else "cpu")
            _dist.all_reduce(fb, op=_dist.ReduceOp.MAX, group=group)
            bad = int(fb.item()) != 0
        return bad

    # ------------------------------------------------------------------------------
    def _fit_factor(self, eng, key, regf, mean, cov, B, niter, nprint, verbose, monitor, retries, rng, as_torch,
                    check_update_flag, shard=False, group=None, graph=None, jitter=0.0, jitter_every=0):
        """Factor-form BaM fit (see ``fit(method="factor")``): the loop of gsmvi/bam.py:140-216 on the state (mean, F).
        ``shard=True``: every rank draws the same Z, samples and scores only its batch_size/world rows; the (x_b, g_b) rows
        are all-gathered and every replica applies the identical factor update (dist.sharded_bam_factor_update)."""
        D = self.D
        assert 2 * B <= min(D, 256), "method='factor' needs 2*batch_size <= min(D, 256)"
        mean_t = eng.zeros(D) if mean is None else eng.clone(mean).reshape(D)
        cov0 = eng.eye(D) if cov is None else eng.clone(cov).reshape(D, D)
        flag, n_rev = eng.new_flag(), eng.new_flag()
        F, _ = eng.potrf(cov0, flag=flag)                   # one factorisation for the whole fit
        if eng.read_flag(flag) != 0:
            raise ValueError("initial covariance is not positive definite")
        seed = int(np.asarray(key.cpu() if _is_torch(key) else key).flatten()[-1])
        rs = np.random.RandomState(seed)
        assert rng in ("auto", "numpy", "device"), "rng must be 'auto', 'numpy' or 'device'"
        dev_rng = rng != "numpy"
        KB = 16
        Zblk = eng.empty(KB, B, D) if dev_rng else None
        ndraw = 0
        native = bool(getattr(self.lp_g, "device_native", False))
        mon_native = bool(getattr(monitor, "device_native", False)) if monitor is not None else False
        lo, hi, world = 0, B, 1
        if shard:
            from .dist import sharded_bam_factor_update, shard_bounds
            import torch.distributed as _dist
            world = _dist.get_world_size(group) if _dist.is_initialized() else 1
            rank = _dist.get_rank(group) if _dist.is_initialized() else 0
            lo, hi = shard_bounds(B, world, rank)
        mean_new, F_new, Xbuf = eng.empty(D), eng.empty(D, D), eng.empty(hi - lo, D)
        state_bufs = [(mean_t, F), (mean_new, F_new)]
        a = 0
        # the jitter of bam.py:198 in factor form: owed = jitter * (accepted updates since the last absorption), absorbed every
        # jitter_every updates by F <- chol(F^T F + owed I) (JITTER_EVERY above); the covariance the monitor and the caller see
        # carries what is still owed, so cov_i = the reference's cov_i up to the response of the last < jitter_every updates
        absorb = jitter > 0.0 and jitter_every > 0
        self.jitter_every_used = jitter_every if absorb else 0
        self.n_absorbed = 0
        pend = 0
        if absorb:
            Cbuf, pflag, mark = eng.empty(D, D), eng.new_flag(), eng.new_flag()

        def cov_of(Fm):
            if absorb and pend > 0:
                return eng.gram(Fm, shift_dev=eng.owed_shift(jitter, pend, n_rev, mark, advance=False))
            return eng.gram(Fm)

        def absorb_now():
            mu_c, F_c = state_bufs[a]
            mu_o, F_o = state_bufs[1 - a]
            eng.gram(F_c, out=Cbuf, shift_dev=eng.owed_shift(jitter, pend, n_rev, mark))
            eng.potrf(Cbuf, out=F_o, flag=pflag)            # the other buffer pair is scratch between two updates
            eng.commit(pflag, mu_c, F_o, mu_o, F_c, None)   # F <- R iff the factorisation succeeded (F^T F + owed I is positive
            self.n_absorbed += 1                            # definite unless the state itself is not finite: then F stays)

        def state():
            c = cov_of(state_bufs[a][1])
            m = state_bufs[a][0]
            return [m, c] if mon_native else [eng.to_numpy(m).copy(), eng.to_numpy(c).copy()]

        nevals = 1
        if nprint > niter:
            nprint = niter
        every = max(1, niter // nprint) if nprint > 0 else 1
        reverts_seen = 0
        # Blocks of KB iterations replayed as ONE hipGraph (as GSM.fit): the launches must all be capturable (a `graph_safe`
        # device score, the counter-based draw stream, no collective, no per-iteration flag read) and the one number that changes
        # between iterations -- reg = regf(i), evaluated on the host by the reference (bam.py:196) -- is read by the kernels from
        # a device word at execution time (engine.bam_reg_source / gsmvi_bam_set_reg_source): iteration k of the block reads word k
        # of a KB-word table that is refilled (regf(i) .. regf(i + KB - 1), one small copy) before every replay.  Blocks with a
        # print or a monitor call, the first block and the tail run eagerly with the by-value argument -- same kernels, same
        # numbers (tests/test_gpu_bam.py: bit-identical).  OFF unless graph=True: measured (scripts/bam_graph_ab.py, marginal
        # iteration of 1200- against 400-iteration fits) the replayed block is not faster than eager launches on this host --
        # 70.9 vs 71.5 us at (256, 8), 151.8 vs 150.1 at (1024, 32), 448 vs 440 at (1024, 128) -- the device is never waiting for
        # the host (enqueueing an update takes 45 - 195 us of host time against 69 - 445 us of device time), and the capture
        # costs ~3 ms.  The 148-against-139 us "eager vs replayed" of a single update is an artefact of timing each eager call
        # from an idle device.  It is kept for callers whose host is slower or shared.
        use_graph = (graph is True and dev_rng and native and not shard and not absorb
                     and not check_update_flag and niter + 1 >= 3 * KB and bool(getattr(self.lp_g, "graph_safe", False)))
        takes_out = False
        if native:
            import inspect
            try:
                takes_out = "out" in inspect.signature(self.lp_g).parameters
            except (TypeError, ValueError):
                takes_out = False
        self.graph_replays = 0
        self.graph_fallback = None
        gstate = {"graph": None}
        if use_graph:
            import torch
            ctr = [torch.zeros(1, dtype=torch.int64, device=Zblk.device) for _ in range(2)]
            reg_blk = torch.zeros(KB, dtype=torch.float64, device=Zblk.device)
            reg_host = [torch.zeros(KB, dtype=torch.float64).pin_memory() for _ in range(4)]
            reg_done = [None] * 4
            Gbuf = eng.empty(hi - lo, D)

        def graph_block(i_first):
            """KB iterations (KB even: the ping-pong state ends where it started) as one replayed graph; the draw counter and
            the regulariser table live on the device."""
            if gstate["graph"] is None:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.stream(side):
                        torch.cuda.synchronize()
                        with torch.cuda.graph(g, stream=side):
                            for half in range(2):
                                h0 = half * (KB // 2)
                                eng.normal_batch(KB // 2, B, D, seed, 0, out=Zblk[h0:h0 + KB // 2], call_in=ctr[half],
                                                 call_out=ctr[1 - half])
                                if self._zc is not None:
                                    Zblk[h0:h0 + KB // 2, :, self._zc:] = 0.0
                                for k in range(KB // 2):
                                    eng.bam_reg_source(reg_blk[h0 + k:h0 + k + 1])
                                    mu_a, F_a = state_bufs[k & 1]
                                    mu_b, F_b = state_bufs[1 - (k & 1)]
                                    Zk = Zblk[h0 + k]
                                    Xk = eng.sample(Zk, mu_a, F_a, out=Xbuf)
                                    vk = self.lp_g(Xk, out=Gbuf) if takes_out else self.lp_g(Xk)
                                    eng.bam_factor_update(Zk, Xk, vk, mu_a, F_a, 1.0, out=(mu_b, F_b), flag=flag, n_reverts=n_rev)   # (reg: ignored, word k is read)
                finally:
                    eng.bam_reg_source(None)
                torch.cuda.current_stream().wait_stream(side)
                gstate["graph"] = g
            slot = self.graph_replays % 4
            if reg_done[slot] is not None:
                reg_done[slot].synchronize()                    # the copy that last read this pinned slot has run
            for k in range(KB):
                rk = float(regf(i_first + k))
                if not rk > 0.0:                                 # (what the by-value entry point checks on the host)
                    raise ValueError(f"BaM.fit: regf({i_first + k}) = {rk}: reg must be positive")
                reg_host[slot][k] = rk
            reg_blk.copy_(reg_host[slot], non_blocking=True)
            reg_done[slot] = torch.cuda.Event()
            reg_done[slot].record()
            ctr[0].fill_(ndraw)
            gstate["graph"].replay()

        i = 0
        while i <= niter:
            blk_end = min(i + KB, niter + 1)
            eventful = any((verbose and j % every == 0) or (monitor is not None and j % monitor.checkpoint == 0)
                           for j in range(i, blk_end))
            # (the first block always runs eagerly: every kernel has been launched, and the context sized, before a capture)
            if use_graph and i > 0 and a == 0 and not eventful and blk_end - i == KB and ndraw % KB == 0:
                try:
                    graph_block(i)
                except Exception as exc:                        # noqa: BLE001 -- capture unsupported here: stay eager
                    if gstate["graph"] is not None:
                        raise
                    import warnings
                    warnings.warn(f"BaM.fit: hipGraph capture of an iteration block failed ({type(exc).__name__}: {exc}); "
                                  "the fit continues with eager launches (same numbers, more launch overhead)", RuntimeWarning)
                    self.graph_fallback = exc
                    use_graph = False
                    torch.cuda.synchronize()
                    continue
                self.graph_replays += 1
                ndraw += KB
                nevals += B * KB
                i = blk_end
                continue
            for i in range(i, blk_end):
                if verbose and i % every == 0:
                    print(f"Iteration {i} of {niter}")
                    r = eng.read_flag(n_rev)
                    if r > reverts_seen:
                        print(f"Bad update for covariance matrix. Revert ({r - reverts_seen} since last print)")
                        reverts_seen = r
                if monitor is not None and i % monitor.checkpoint == 0:
                    monitor(i, state(), self.lp, key, nevals=nevals)
                    nevals = 0
                mu_a, F_a = state_bufs[a]
                mu_b, F_b = state_bufs[1 - a]
                j = 0
                while True:
                    try:
                        if dev_rng:
                            if ndraw % KB == 0:
                                eng.normal_batch(KB, B, D, seed, ndraw, out=Zblk)
                                if self._zc is not None:
                                    Zblk[:, :, self._zc:] = 0.0              # inert coordinates of an odd-D fit (_oddpad.py)
                            Z = Zblk[ndraw % KB]
                            ndraw += 1
                        else:
                            Z = eng.normal_from_host(_host_draw(rs, B, D, self._zc))
                            if self._zc is not None:
                                Z[:, self._zc:] = 0.0
                        X = eng.sample(Z[lo:hi], mu_a, F_a, out=Xbuf)          # only this rank's rows when sharded
                        err = None
                        try:
                            vs = self.lp_g(X) if native else eng.host_score(self.lp_g, X)
                        except Exception as e_score:            # noqa: BLE001
                            if not (shard and world > 1):
                                raise
                            err, vs = e_score, None
                        if shard and world > 1:                 # agree on failure BEFORE anybody enters the gather (as the dense fit)
                            import torch
                            fb = torch.tensor([0 if err is None else 1], dtype=torch.int32,
                                              device=X.device if _is_torch(X) else "cpu")
                            _dist.all_reduce(fb, op=_dist.ReduceOp.MAX, group=group)
                            if int(fb.item()) != 0:
                                raise err if err is not None else RuntimeError("score evaluation failed on another rank")
                        nevals += B
                        reg = regf(i)
                        if shard:
                            sharded_bam_factor_update(eng, Z, X, vs, mu_a, F_a, reg, group=group, out=(mu_b, F_b), flag=flag,
                                                      n_reverts=n_rev)
                        else:
                            eng.bam_factor_update(Z, X, vs, mu_a, F_a, reg, out=(mu_b, F_b), flag=flag, n_reverts=n_rev)
                        if check_update_flag and self._flag_raised(eng, flag, shard and world > 1, group):
                            raise FloatingPointError("BaM update flagged a numerical failure (device flag != 0)")
                        break
                    except Exception as e:                      # noqa: BLE001 -- reference behaviour
                        if j < retries:
                            j += 1
                            print(f"Failed with exception {e}")
                            print(f"Trying again {j} of {retries}")
                        else:
                            raise e
                a = 1 - a               # the kernel already returned the reverted state when its test failed: accept = swap
                if absorb:
                    pend += 1
                    if pend >= jitter_every:
                        absorb_now()
                        pend = 0
            i = blk_end
        i = niter
        if monitor is not None:
            monitor(i, state(), self.lp, key, nevals=nevals)
        self.n_reverts = eng.read_flag(n_rev)
        mean_t, F = state_bufs[a]
        cov_t = cov_of(F)
        if as_torch:
            return mean_t, cov_t
        return eng.to_numpy(mean_t), eng.to_numpy(cov_t)
