"""GSM fit driver and update function with the reference's signatures.

Mirrors ``gsmvi/gsm_numpy.py`` / ``gsmvi/gsm.py`` (reference file:line in each docstring).  The
driver is host Python exactly like the reference's; every array operation inside the loop is a
HIP kernel call through the engine (``engine.py`` -> C ABI -> ``csrc/``).
"""
import numpy as np
import torch

from .engine import get_engine


def _is_torch(x):
    return isinstance(x, torch.Tensor)


def gsm_update(samples, vs, mu0, S0, engine=None, assume_symmetric=None):
    """Drop-in for ``gsm_update(samples, vs, mu0, S0)`` (gsmvi/gsm_numpy.py:27-55, gsmvi/gsm.py:31-58).

    Inputs (B,D), (B,D), (D,), (D,D); returns new ``(mu, S)`` and never modifies its inputs.
    numpy in -> float64 numpy out (as gsm_numpy.py:47 does); CUDA torch tensors in -> torch out.
    Shape errors raise AssertionError like the reference (gsm_numpy.py:43-44).

    ``assume_symmetric`` (not in the reference): the fast update kernel reads only the upper triangle of S0 (a
    covariance is symmetric).  ``None`` (default): a HOST S0 -- a numpy array or a CPU torch tensor -- is checked on the
    host before its upload (no device work, no synchronisation) and a non-symmetric one takes the general kernels, which
    read all of S0 and keep the reference's literal semantics S = S0 + mean (gsm_numpy.py:50-53); a DEVICE (CUDA) S0 is
    taken to be symmetric -- no D x D compare and no host synchronisation sit in front of the update: a non-symmetric
    device S0 gives the update of its upper triangle mirrored, NOT the reference's result.  ``"check"`` compares a device
    S0 with its transpose on the device (one D x D pass and one host synchronisation) and routes like the host check;
    ``False`` forces the general kernels, ``True`` skips every check.
    """
    assert len(samples.shape) == 2
    assert len(vs.shape) == 2
    eng = engine if engine is not None else get_engine()
    want_torch = _is_torch(samples)
    if assume_symmetric is None or assume_symmetric == "check":
        if _is_torch(S0) and S0.is_cuda:
            assume_symmetric = bool(torch.equal(S0, S0.T)) if assume_symmetric == "check" else True
        elif _is_torch(S0):
            assume_symmetric = bool(torch.equal(S0, S0.T))
        else:
            assume_symmetric = bool(np.array_equal(np.asarray(S0), np.asarray(S0).T))
    Xd, Gd, m0, S0d = eng.asarray(samples), eng.asarray(vs), eng.asarray(mu0), eng.asarray(S0)
    D = int(m0.shape[0])
    if assume_symmetric and D % 2 == 1 and getattr(eng, "name", "") == "hip":
        # odd D: the (D + 1)-dimensional problem with an inert last coordinate runs on the tuned kernels (_oddpad.py); the
        # padding copies ride on the uploads for host inputs
        from . import _oddpad
        mu, S = eng.gsm_update(_oddpad.pad_rows(eng, Xd, D), _oddpad.pad_rows(eng, Gd, D), _oddpad.pad_vec(eng, m0, D),
                               _oddpad.pad_mat(eng, S0d, D))
        mu, S = mu[:D].contiguous(), S[:D, :D].contiguous()
    elif assume_symmetric:
        mu, S = eng.gsm_update(Xd, Gd, m0, S0d)
    else:
        mu, S = eng.gsm_update(Xd, Gd, m0, S0d, general=True)
    return (mu, S) if want_torch else (eng.to_numpy(mu), eng.to_numpy(S))


def _legacy_mvn(rs, mean, cov, size):
    """Compat sampler: the exact stream of ``np.random.multivariate_normal`` after
    ``np.random.seed(key)`` (gsmvi/gsm_numpy.py:105,116): z from MT19937, SVD factor of cov."""
    D = mean.shape[0]
    z = rs.standard_normal((size, D))
    _, s, vt = np.linalg.svd(cov)
    return mean + z @ (np.sqrt(s)[:, None] * vt)


def _host_draw(rs, B, D, zc):
    """(B, D) standard normals from the host stream (rng="numpy": the reference's z-stream, gsm_numpy.py:105,116).  In the padded
    fit of an odd-D problem (zc = the literal D, _oddpad.py) the stream is drawn at the LITERAL width and the inert columns are
    zeros, so a seed gives the same draws as the literal-D problem (round-5 advice: it drew (B, D + 1) and shifted the stream)."""
    if zc is None:
        return rs.standard_normal((B, D))
    z = np.zeros((B, D))
    z[:, :zc] = rs.standard_normal((B, zc))
    return z


class GSM:
    """Wrapper class for using GSM updates to fit a distribution (gsmvi/gsm_numpy.py:60-75,
    gsmvi/gsm.py:62-77).

    D    : dimensionality.
    lp   : target log-probability; only handed to ``monitor`` (gsm_numpy.py:68-69).
    lp_g : score function (B,D) -> (B,D).  A plain callable receives and returns numpy arrays,
           exactly as in the reference (its samples are host numpy arrays, gsm_numpy.py:116-117).
           A callable marked ``device_native`` (``gsmvi_amd.device_score`` /
           ``GaussianTarget.lp_g`` / ``score_from_logp``) receives and returns float64 CUDA
           tensors and keeps the whole iteration on the GPU.
    """

    def __init__(self, D, lp, lp_g, engine=None):
        self.D = D
        self.lp = lp
        self.lp_g = lp_g
        self._engine = engine

    # ------------------------------------------------------------------------------
    def fit(self, key, mean=None, cov=None, batch_size=2, niter=5000, nprint=10, verbose=True,
            check_goodness=True, monitor=None, *, sampler="cholesky", rng="auto", as_torch=False,
            forced_samples=None, method="auto", shard=False, group=None, graph=None, root_potrf=False, _zero_cols_from=None):
        """Fit N(mean, cov) to the target (gsmvi/gsm_numpy.py:77-129, gsmvi/gsm.py:79-133).

        Same arguments and return value as the reference.  Behaviour kept: ``niter + 1`` updates
        (:106); monitor called every ``monitor.checkpoint`` iterations and once at the end with
        the reference's ``nevals`` bookkeeping (:103,:110-113,:119,:127-128); a covariance that
        fails the Cholesky test reverts BOTH mean and cov (:121-125); ``check_goodness`` is
        accepted and, as in the reference (:77 vs :121), the test always runs.
        Documented deviations: ``nprint`` is clamped to ``niter`` like bam.py:177 instead of
        raising ZeroDivisionError (:107); the RNG is a private ``RandomState(key)`` with the same
        z-stream as ``np.random.seed(key)`` (:105); revert messages are printed at the progress
        prints (count since the last print) so that the loop never synchronises per iteration.

        Extra keyword-only arguments (not in the reference):
          sampler : "cholesky" (default) x = mean + R^T z from the device Cholesky factor, or "svd":
                    the reference's legacy host sampler, bit-compatible sample stream (small D).
          rng     : "auto" (default) = "device" with the Cholesky sampler, "numpy" with sampler="svd".  With the
                    Cholesky sampler the samples differ from the reference's (SVD factor) whatever the z-stream is,
                    so the default takes the stream that keeps the iteration on the GPU (a host MT19937 draw of
                    B x D normals plus its upload costs more than the whole device iteration).
                    "numpy" (host MT19937 z-stream, uploaded: the reference's stream, gsm_numpy.py:105) or
                    "device" (counter-based Philox stream generated by gsmvi_randn_f64 from (key, iteration):
                    nothing crosses PCIe, and sharded ranks draw identical Z).
          forced_samples : (niter+1, B, D) teacher-forced samples replacing the sampler.
          shard   : True / "batch": batch-sharded multi-GPU fit (one process per GPU, torch.distributed initialised; both
                    methods).  "cols" (round 6): the COLUMN-sharded factor form -- every rank keeps D / P
                    columns of the square factor and its entries of the mean, samples its slice x[:, C] = mean[C] + z F[:, C]
                    (all-gathered: B D / P doubles per rank), evaluates lp_g on all B samples, contributes the partial
                    product G[:, C] F[:, C]^T to one all-reduce (B D doubles) and updates its block alone: the D^2-sized
                    traffic and memory of the factor form are divided by P (the batch-sharded form divides neither).  Needs
                    D % (64 P) == 0 and 2 batch_size <= min(D, 256).  All ranks return the same (mean, cov).
                    batch-sharded (True):  Every rank draws the same z-stream (same key, either rng), so samples are
                    replicated; each rank evaluates ``lp_g`` only on its batch_size/world rows, the
                    per-sample records are all-gathered (RCCL) and every replica applies the identical
                    combined update (gsm-vi_amd/dist.py).  All ranks return the same (mean, cov).
          root_potrf : sharded DENSE fit only.  False (default): every rank factors its (bit-identical) covariance replica --
                    the factorisations run in parallel and cost no wall-clock time over a single GPU's.  True: rank 0 factors
                    and broadcasts the D x D factor and its flag (dist.root_potrf): saves the redundant D^3 work (energy, or
                    a GPU shared with other work) at the price of a serial factorisation + 8 D^2-byte broadcast per
                    iteration; not measured on more than one GPU, hence opt-in (round-3 advice).
          graph   : the factor-form fit can replay blocks of 16 iterations as one hipGraph when every launch in them is
                    capturable (a score marked ``graph_safe`` such as ``GaussianTarget.lp_g``, the device draw stream, no
                    sharding).  None (default): do so for D <= 512, where the Python / launch overhead is the bound; True:
                    always; False: never.  Same numbers either way.
          method  : "auto" (default) = "factor" whenever 2*batch_size <= min(D, 128) -- or <= 256 for D >= 1024 --, the device
                    Cholesky sampler, no teacher-forced samples) and "dense" otherwise.  Why that is a drop-in
                    default: for the same draws the two forms give the same (mean, cov) to round-off
                    (tests/test_gpu_factor.py, <= 1e-14 even at cond 1e8), the 2B x 2B positive-definite test is
                    the algebraic equivalent of the reference's Cholesky test (gsm_numpy.py:132-146), and the
                    iteration costs no O(D^3) work (8.7k against 1.5k iterations/s at D=1024, B=32).
                    "dense" keeps Sigma and re-factorises it every iteration exactly like the
                    reference (Cholesky = its _check_goodness).  "factor" keeps a square factor F with
                    Sigma = F^T F instead (BASELINE config 5): samples are mean + z F, the
                    update is a rank-2B correction of F and the positive-definite test is a Cholesky of
                    a 2B x 2B matrix, so no O(D^3) work per iteration.  Same (mean, cov) up to round-off
                    for the same samples; needs 2B <= min(D, 256) and the device sampler.  It converges to
                    machine precision on Gaussian targets like the dense form (tests/test_gpu_factor.py).
                    Linearly dependent rows of [Z; U] -- an isotropic state on an isotropic target (every
                    u_b - a_b z_b is parallel to mean - m), or a state that is EXACTLY the fixed point (U = 0) --
                    make the 2B x 2B Gram matrix singular; they are handled by a rank-revealing rule (a pivot at the
                    rounding floor of its row drops the row; csrc/gsmvi_chol64.h, SEMIDEF), agree with the dense
                    update to 1e-15 (test_factor_update_with_linearly_dependent_rows) and are not reverts.
                    Accept/revert in factor form: Sigma' = F'^T F' is positive semi-definite by construction, so
                    the ROUNDING failures that make the dense Sigma' indefinite (the reference's only source of
                    reverts on finite inputs, gsm_numpy.py:121-125) cannot occur; what the 2B x 2B test rejects
                    are NaN/inf inputs, updates whose small matrix is numerically indefinite, and whitened rows
                    of absurd size (|z|^2 >= 2^32) that are dependent to rounding.  The reference's decisive revert
                    fixture G4 (tests/golden/g4_revert.npz: eigenvalues 1e-14 next to O(1e3) displacements, |z| ~
                    1e10) is ALSO reverted in factor form (tests/test_gpu_factor.py::test_g4_...), state kept bit
                    for bit.
        """
        D_, B_ = self.D, int(batch_size)
        from . import _oddpad
        eng0 = self._engine if self._engine is not None else get_engine()
        if _zero_cols_from is None and _oddpad.applies(eng0, D_, sampler, forced_samples):
            # odd D: the (D + 1)-dimensional problem with an inert last coordinate runs on the tuned kernels (_oddpad.py)
            if method == "auto":
                nmax = 256 if D_ >= 1024 else 128
                method = "factor" if 2 * B_ <= min(D_, nmax) else "dense"
            inner = GSM(D_ + 1, self.lp, _oddpad.wrap_score(eng0, self.lp_g, D_), engine=eng0)
            mp, cp = inner.fit(key, mean=_oddpad.pad_vec(eng0, mean, D_), cov=_oddpad.pad_mat(eng0, cov, D_),
                               batch_size=batch_size, niter=niter, nprint=nprint, verbose=verbose,
                               check_goodness=check_goodness, monitor=_oddpad.wrap_monitor(monitor, self.lp, D_),
                               sampler=sampler, rng=rng, as_torch=True, method=method, shard=shard, group=group, graph=graph,
                               root_potrf=root_potrf, _zero_cols_from=D_)
            for a in ("method_used", "n_reverts", "graph_replays", "graph_fallback"):
                if hasattr(inner, a):
                    setattr(self, a, getattr(inner, a))
            self.padded_dim = D_ + 1
            mean_o, cov_o = mp[:D_].contiguous(), cp[:D_, :D_].contiguous()
            return (mean_o, cov_o) if as_torch else (eng0.to_numpy(mean_o), eng0.to_numpy(cov_o))
        self._zc = _zero_cols_from
        if shard == "cols" and method == "auto":
            method = "factor"
        if method == "auto":
            # 2B <= 128: always (the measured range of rounds 2-4).  128 < 2B <= 256 (two-level chain): where the dense loop's
            # O(D^3) Cholesky costs more than the whole factor update -- measured at (1024, 128): factor update 271 us against
            # 330 us for the Cholesky alone; at D = 512 the dense iteration (165 + ~40 us) still wins -- so from D = 1024 on.
            nmax = 256 if D_ >= 1024 else 128
            method = "factor" if (sampler == "cholesky" and forced_samples is None
                                  and 2 * B_ <= min(D_, nmax)) else "dense"
        if shard == "cols":
            assert method in ("auto", "factor") and sampler == "cholesky" and forced_samples is None, \
                "shard='cols' is the column-sharded FACTOR form (device sampler, no forced samples)"
            self.method_used = "factor"
            return self._fit_factor_cols(key, mean, cov, batch_size, niter, nprint, verbose, monitor, rng, as_torch, group)
        self.method_used = method
        if method == "factor":
            return self._fit_factor(key, mean, cov, batch_size, niter, nprint, verbose, monitor, rng, as_torch,
                                    shard, group, graph)
        assert method == "dense", "method must be 'auto', 'dense' or 'factor'"
        eng = self._engine if self._engine is not None else get_engine()
        D, B = self.D, int(batch_size)
        mean_t = eng.zeros(D) if mean is None else eng.clone(mean).reshape(D)
        cov_t = eng.eye(D) if cov is None else eng.clone(cov).reshape(D, D)
        nevals = 1
        seed = int(key) if not _is_torch(key) else int(key.flatten()[0])
        rs = np.random.RandomState(seed)
        assert rng in ("auto", "numpy", "device"), "rng must be 'auto', 'numpy' or 'device'"
        dev_rng = rng == "device" or (rng == "auto" and sampler == "cholesky")
        KB = 16
        Zblk = eng.empty(KB, B, D) if dev_rng else None
        native = bool(getattr(self.lp_g, "device_native", False))
        mon_native = bool(getattr(monitor, "device_native", False)) if monitor is not None else False

        # working buffers (the user's arrays are never aliased or mutated)
        mean_new, cov_new = eng.empty(D), eng.empty(D, D)
        R, R_new = eng.empty(D, D), eng.empty(D, D)
        Xbuf = eng.empty(B, D)
        flag, n_rev = eng.new_flag(), eng.new_flag()
        use_factor = sampler == "cholesky" and forced_samples is None
        if use_factor:                      # the sampling factor of the initial covariance
            eng.potrf(cov_t, out=R, flag=flag)
            if eng.read_flag(flag) != 0:
                raise ValueError("initial covariance is not positive definite")

        nprint = max(1, min(int(nprint), int(niter))) if niter > 0 else 1   # bam.py:177 guard
        every = max(1, niter // nprint) if niter > 0 else 1
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

            if forced_samples is not None:
                X = eng.asarray(forced_samples[i])
            elif sampler == "svd":
                X = eng.asarray(_legacy_mvn(rs, eng.to_numpy(mean_t), eng.to_numpy(cov_t), B))
            else:
                if dev_rng:                                     # a block of KB iterations' draws per launch (same stream)
                    if i % KB == 0:
                        eng.normal_batch(min(KB, niter + 1 - i), B, D, seed, i, out=Zblk[:min(KB, niter + 1 - i)])
                        if self._zc is not None:
                            Zblk[:, :, self._zc:] = 0.0                  # inert coordinates of an odd-D fit (_oddpad.py)
                    Z = Zblk[i % KB]
                else:
                    Z = eng.normal_from_host(_host_draw(rs, B, D, self._zc))
                    if self._zc is not None:
                        Z[:, self._zc:] = 0.0
                X = eng.sample(Z, mean_t, R, out=Xbuf)
            if shard:
                from .dist import sharded_gsm_update, shard_bounds
                import torch.distributed as _dist
                world = _dist.get_world_size(group) if _dist.is_initialized() else 1
                rank = _dist.get_rank(group) if _dist.is_initialized() else 0
                lo, hi = shard_bounds(B, world, rank)
                Xl = X[lo:hi]
                vl = self.lp_g(Xl) if native else eng.host_score(self.lp_g, Xl)
                sharded_gsm_update(eng, Xl, vl, mean_t, cov_t, group=group, out=(mean_new, cov_new))
            else:
                vs = self.lp_g(X) if native else eng.host_score(self.lp_g, X)
                eng.gsm_update(X, vs, mean_t, cov_t, out=(mean_new, cov_new))
            nevals += B
            if shard and root_potrf:                              # opt-in: one rank factors, the others receive
                from .dist import root_potrf as _root_potrf
                _root_potrf(eng, cov_new, R_new, flag, group=group)
            else:
                eng.potrf(cov_new, out=R_new, flag=flag)          # _check_goodness, :121,:132-146
            eng.commit(flag, mean_new, cov_new, mean_t, cov_t, n_rev)
            if use_factor:
                eng.commit(flag, mean_new, R_new, mean_t, R, None)

        if verbose:
            r = eng.read_flag(n_rev)
            if r > reverts_seen:
                print(f"Bad update for covariance matrix. Revert ({r - reverts_seen} since last print)")
        if monitor is not None:
            mc = [mean_t, cov_t] if mon_native else [eng.to_numpy(mean_t).copy(), eng.to_numpy(cov_t).copy()]
            monitor(i, mc, self.lp, key, nevals=nevals)
        self.n_reverts = eng.read_flag(n_rev)
        if as_torch:
            return mean_t, cov_t
        return eng.to_numpy(mean_t), eng.to_numpy(cov_t)

    # ------------------------------------------------------------------------------
    def _fit_factor_cols(self, key, mean, cov, batch_size, niter, nprint, verbose, monitor, rng, as_torch, group=None):
        """Column-sharded factor-form fit (see ``fit(shard="cols")``): the loop of gsm_numpy.py:77-129 on the state
        (mean[C], F[:, C]) per rank.  Per iteration and rank: one pass over the block for the sample slice, one for the partial
        W, two for the update (read + write) and the V F product -- 32 D (D / P) bytes --, one all-gather of B D / P doubles and
        one all-reduce of B D doubles; the 2B x 2B chain and the score evaluation are replicated.  The initial factorisation is
        replicated too (once per fit); the covariance is assembled only for the monitor and the return value."""
        import torch.distributed as _dist
        from .dist import col_bounds, col_gather_samples, col_sharded_gsm_factor_update, _all_gather, _as_torch
        eng = self._engine if self._engine is not None else get_engine()
        D, B = self.D, int(batch_size)
        assert 2 * B <= min(D, 256), "the factor form needs 2*batch_size <= min(D, 256)"
        world = _dist.get_world_size(group) if _dist.is_initialized() else 1
        rank = _dist.get_rank(group) if _dist.is_initialized() else 0
        lo, hi = col_bounds(D, world, rank)
        nc = hi - lo
        mean_a = eng.zeros(D) if mean is None else eng.clone(mean).reshape(D)
        cov0 = eng.eye(D) if cov is None else eng.clone(cov).reshape(D, D)
        flag, n_rev = eng.new_flag(), eng.new_flag()
        F0, _ = eng.potrf(cov0, flag=flag)
        if eng.read_flag(flag) != 0:
            raise ValueError("initial covariance is not positive definite")
        Fc_a = eng.clone(F0[:, lo:hi])                      # the owned block, D x D / P, its own leading dimension
        del F0, cov0
        mean_b, Fc_b = eng.clone(mean_a), eng.empty(D, nc)
        bufs = [(mean_a, Fc_a), (mean_b, Fc_b)]
        a = 0
        seed = int(key) if not _is_torch(key) else int(key.flatten()[0])
        rs = np.random.RandomState(seed)
        assert rng in ("auto", "numpy", "device"), "rng must be 'auto', 'numpy' or 'device'"
        dev_rng = rng != "numpy"
        native = bool(getattr(self.lp_g, "device_native", False))
        mon_native = bool(getattr(monitor, "device_native", False)) if monitor is not None else False
        KB = 16
        Zblk = eng.empty(KB, B, D) if dev_rng else None
        self.shard_stats = {}

        def assemble():
            """(mean, F) in full on every rank: all-gather of the owned mean entries and column blocks"""
            m_c, F_c = bufs[a]
            if world == 1:
                return m_c, F_c
            recv = eng.empty(world * D, nc)
            _all_gather(_as_torch(recv), _as_torch(F_c), group)
            r = recv if _is_torch(recv) else _as_torch(recv).numpy()
            F = eng.empty(D, D)
            for p in range(world):
                F[:, p * nc:(p + 1) * nc] = r[p * D:(p + 1) * D]
            mrecv = eng.empty(world * nc)
            _all_gather(_as_torch(mrecv), _as_torch(eng.clone(m_c[lo:hi])), group)
            m = eng.clone(mrecv) if _is_torch(mrecv) else np.array(_as_torch(mrecv).numpy(), copy=True)
            return m, F

        def state():
            m, F = assemble()
            c = eng.gram(F)
            return [m, c] if mon_native else [eng.to_numpy(m).copy(), eng.to_numpy(c).copy()]

        nevals = 1
        nprint = max(1, min(int(nprint), int(niter))) if niter > 0 else 1
        every = max(1, niter // nprint) if niter > 0 else 1
        reverts_seen = 0
        for i in range(niter + 1):
            if verbose and i % every == 0:
                print(f"Iteration {i} of {niter}")
                r = eng.read_flag(n_rev)
                if r > reverts_seen:
                    print(f"Bad update for covariance matrix. Revert ({r - reverts_seen} since last print)")
                    reverts_seen = r
            if monitor is not None and i % monitor.checkpoint == 0:
                monitor(i, state(), self.lp, key, nevals=nevals)
                nevals = 0
            if dev_rng:
                if i % KB == 0:
                    eng.normal_batch(min(KB, niter + 1 - i), B, D, seed, i, out=Zblk[:min(KB, niter + 1 - i)])
                Z = Zblk[i % KB]
            else:
                Z = eng.normal_from_host(rs.standard_normal((B, D)))
            m_c, F_c = bufs[a]
            m_n, F_n = bufs[1 - a]
            Xc = eng.sample_cols(Z, m_c[lo:hi], F_c)
            X = col_gather_samples(eng, Xc, group, stats=self.shard_stats if i == 0 else None)
            vs = self.lp_g(X) if native else eng.host_score(self.lp_g, X)
            col_sharded_gsm_factor_update(eng, Z, X, vs, m_c, F_c, group=group, out=(m_n, F_n), flag=flag, n_reverts=n_rev,
                                          stats=self.shard_stats if i == 0 else None)
            nevals += B
            a = 1 - a
        if verbose:
            r = eng.read_flag(n_rev)
            if r > reverts_seen:
                print(f"Bad update for covariance matrix. Revert ({r - reverts_seen} since last print)")
        if monitor is not None:
            monitor(niter, state(), self.lp, key, nevals=nevals)
        self.n_reverts = eng.read_flag(n_rev)
        self.shard_stats["block_bytes"] = D * nc * 8
        m, F = assemble()
        cov_t = eng.gram(F)
        if as_torch:
            return m, cov_t
        return eng.to_numpy(m), eng.to_numpy(cov_t)

    # ------------------------------------------------------------------------------
    def _fit_factor(self, key, mean, cov, batch_size, niter, nprint, verbose, monitor, rng, as_torch, shard=False,
                    group=None, graph=None):
        """Factor-form fit loop (see ``fit(method="factor")``): same driver logic as gsm_numpy.py:77-129,
        state (mean, F) with cov = F^T F materialised only for the monitor and the return value.
        ``shard=True``: every rank draws the same Z, samples and scores only its batch_size/world rows, and the
        per-sample records are all-gathered (dist.sharded_gsm_factor_update); replicas stay identical."""
        eng = self._engine if self._engine is not None else get_engine()
        D, B = self.D, int(batch_size)
        assert 2 * B <= min(D, 256), "method='factor' needs 2*batch_size <= min(D, 256)"
        mean_t = eng.zeros(D) if mean is None else eng.clone(mean).reshape(D)
        cov0 = eng.eye(D) if cov is None else eng.clone(cov).reshape(D, D)
        flag, n_rev = eng.new_flag(), eng.new_flag()
        F, _ = eng.potrf(cov0, flag=flag)                   # one factorisation for the whole fit
        if eng.read_flag(flag) != 0:
            raise ValueError("initial covariance is not positive definite")
        seed = int(key) if not _is_torch(key) else int(key.flatten()[0])
        rs = np.random.RandomState(seed)
        assert rng in ("auto", "numpy", "device"), "rng must be 'auto', 'numpy' or 'device'"
        dev_rng = rng != "numpy"                            # the factor form always samples with its own factor
        Zbuf = eng.empty(B, D) if dev_rng else None
        native = bool(getattr(self.lp_g, "device_native", False))
        mon_native = bool(getattr(monitor, "device_native", False)) if monitor is not None else False
        lo, hi = 0, B
        if shard:
            from .dist import sharded_gsm_factor_update, shard_bounds
            import torch.distributed as _dist
            world = _dist.get_world_size(group) if _dist.is_initialized() else 1
            rank = _dist.get_rank(group) if _dist.is_initialized() else 0
            lo, hi = shard_bounds(B, world, rank)
        mean_new, F_new, Xbuf = eng.empty(D), eng.empty(D, D), eng.empty(hi - lo, D)

        def state():
            c = eng.gram(F)
            return [mean_t, c] if mon_native else [eng.to_numpy(mean_t).copy(), eng.to_numpy(c).copy()]

        nevals = 1
        nprint = max(1, min(int(nprint), int(niter))) if niter > 0 else 1
        every = max(1, niter // nprint) if niter > 0 else 1
        reverts_seen = 0
        # The draw stream does not depend on the state: the device draws come a BLOCK of KB iterations per launch.
        KB = 16
        Zblk = eng.empty(KB, B, D) if dev_rng else None
        Gbuf = eng.empty(hi - lo, D)
        # A block of KB iterations whose launches are all capturable (a `graph_safe` device score, the counter-based draw
        # stream, no sharding collective) is captured ONCE into a hipGraph and replayed: at small D the Python / launch
        # overhead of ~10 calls per iteration is the bound (D = 256, B = 8: 58 us eager against 50 us replayed).  Blocks
        # that contain a print or a monitor call run eagerly, so the reference's cadence is untouched.
        # By default only where it pays: at D >= 1024 the iteration is GPU-bound (D = 1024, B = 32: 85 us eager = 85 us replayed)
        # and capturing ~200 kernel nodes costs a few ms, so graph=None takes the graph for D <= 512; graph=True forces it.
        use_graph = ((graph is True or (graph is None and D <= 512)) and dev_rng and native and not shard
                     and niter + 1 >= 3 * KB and bool(getattr(self.lp_g, "graph_safe", False)))
        takes_out = False
        if use_graph:
            import inspect
            try:
                takes_out = "out" in inspect.signature(self.lp_g).parameters
            except (TypeError, ValueError):
                takes_out = False
        gstate = {"graph": None}
        self.graph_replays = 0                                  # blocks replayed from the captured graph (tests / diagnostics)
        self.graph_fallback = None                              # the exception that sent a graph fit back to eager launches
        if use_graph:
            ctr = [torch.zeros(1, dtype=torch.int64, device=Zblk.device) for _ in range(2)]
        state_bufs = [(mean_t, F), (mean_new, F_new)]

        def iteration(Zi, a):
            mu_a, F_a = state_bufs[a]
            mu_b, F_b = state_bufs[1 - a]
            X = eng.sample(Zi[lo:hi], mu_a, F_a, out=Xbuf)       # only this rank's rows when sharded
            if native:
                vs = self.lp_g(X, out=Gbuf) if takes_out else self.lp_g(X)
            else:
                vs = eng.host_score(self.lp_g, X, out=Gbuf)
            if shard:
                sharded_gsm_factor_update(eng, Zi, X, vs, mu_a, F_a, lo, group=group, out=(mu_b, F_b),
                                          flag=flag, n_reverts=n_rev)
            else:
                eng.gsm_factor_update(Zi, X, vs, mu_a, F_a, out=(mu_b, F_b), flag=flag, n_reverts=n_rev)

        def graph_block():
            """KB iterations (KB even: the ping-pong state ends where it started) as one replayed graph; the draw counter
            lives on the device and advances by KB per replay (two half-block draws on a ping-pong word pair)."""
            if gstate["graph"] is None:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g, stream=side):
                        for half in range(2):
                            eng.normal_batch(KB // 2, B, D, seed, 0, out=Zblk[half * (KB // 2):(half + 1) * (KB // 2)],
                                             call_in=ctr[half], call_out=ctr[1 - half])
                            if self._zc is not None:
                                Zblk[half * (KB // 2):(half + 1) * (KB // 2), :, self._zc:] = 0.0
                            for k in range(KB // 2):
                                iteration(Zblk[half * (KB // 2) + k], k & 1)
                torch.cuda.current_stream().wait_stream(side)
                gstate["graph"] = g
            gstate["graph"].replay()

        a = 0                                                   # which buffer pair holds the current state
        i = 0
        while i <= niter:
            blk_end = min(i + KB, niter + 1)
            eventful = any((verbose and j % every == 0) or (monitor is not None and j % monitor.checkpoint == 0)
                           for j in range(i, blk_end))
            # (the first block always runs eagerly: every kernel has been launched, and the context sized, before a capture)
            if use_graph and i > 0 and a == 0 and not eventful and blk_end - i == KB:
                ctr[0].fill_(i)                                 # (stream-ordered; the graph reads it on the device)
                try:
                    graph_block()
                except Exception as exc:                        # capture unsupported here: stay eager for the rest of the fit
                    if gstate["graph"] is not None:
                        raise
                    import warnings
                    warnings.warn(f"GSM.fit: hipGraph capture of an iteration block failed ({type(exc).__name__}: {exc}); "
                                  "the fit continues with eager launches (same numbers, more launch overhead)", RuntimeWarning)
                    self.graph_fallback = exc
                    use_graph = False
                    torch.cuda.synchronize()
                    continue
                self.graph_replays += 1
                nevals += B * KB
                i = blk_end
                continue
            if dev_rng:
                eng.normal_batch(blk_end - i, B, D, seed, i, out=Zblk[:blk_end - i])
                if self._zc is not None:
                    Zblk[:, :, self._zc:] = 0.0                          # inert coordinates of an odd-D fit (_oddpad.py)
            for j in range(i, blk_end):
                mean_t, F = state_bufs[a]
                if verbose and j % every == 0:
                    print(f"Iteration {j} of {niter}")
                    r = eng.read_flag(n_rev)
                    if r > reverts_seen:
                        print(f"Bad update for covariance matrix. Revert ({r - reverts_seen} since last print)")
                        reverts_seen = r
                if monitor is not None and j % monitor.checkpoint == 0:
                    monitor(j, state(), self.lp, key, nevals=nevals)
                    nevals = 0
                Zi = Zblk[j - i] if dev_rng else eng.normal_from_host(_host_draw(rs, B, D, self._zc))
                if not dev_rng and self._zc is not None:
                    Zi[:, self._zc:] = 0.0
                iteration(Zi, a)
                nevals += B
                a = 1 - a                                       # the kernel already returned the reverted state when its PD
            i = blk_end                                         # test failed: accept = swapping the buffer pair
        mean_t, F = state_bufs[a]
        i = niter
        if verbose:
            r = eng.read_flag(n_rev)
            if r > reverts_seen:
                print(f"Bad update for covariance matrix. Revert ({r - reverts_seen} since last print)")
        if monitor is not None:
            monitor(i, state(), self.lp, key, nevals=nevals)
        self.n_reverts = eng.read_flag(n_rev)
        cov_t = eng.gram(F)
        if as_torch:
            return mean_t, cov_t
        return eng.to_numpy(mean_t), eng.to_numpy(cov_t)
