"""Odd D on the tuned kernels (round 5).  The tuned kernels take any EVEN D (16-byte row alignment of a row-major D x D
matrix needs it); the reference takes any D (gsm_numpy.py:27-55, bam.py:31-114; its own example is D = 5).  An odd-D problem
is therefore run as the (D + 1)-dimensional problem whose last coordinate is INERT:

    mean' = [mean, 0]     cov' = blockdiag(cov, 1)  (factor form: F' = blockdiag(F, 1))     x' = [x, 0]     g' = [g, 0]

With a zero last column in the draws (z' = [z, 0]) the samples keep x'_D = 0, so d_D = mu'_D - x'_D = 0 and g'_D = 0 for every
sample: the extra row / column of every rank-2B (GSM: gsm_numpy.py:17-23) or low-rank (BaM: bam.py:105-112) correction is exactly
zero, the border of cov' (or F') never changes, and the leading D x D block is the update of the original problem -- the same
arithmetic on the same numbers plus exact zeros.  Costs one D^2-sized padding copy per FIT (not per iteration)."""
import numpy as np
import torch


def pad_vec(eng, v, D):
    out = eng.zeros(D + 1)
    if v is not None:
        out[:D] = eng.asarray(v).reshape(D)
    return out


def pad_mat(eng, M, D, corner=1.0):
    out = eng.zeros(D + 1, D + 1)
    if M is None:
        out[:D, :D] = eng.eye(D)
    else:
        out[:D, :D] = eng.asarray(M).reshape(D, D)
    out[D, D] = corner
    return out


def pad_rows(eng, X, D):
    X = eng.asarray(X)
    out = eng.zeros(X.shape[0], D + 1)
    out[:, :D] = X
    return out


def wrap_score(eng, lp_g, D):
    """lp_g of the padded problem: evaluates the user's callable on the first D columns, returns [g, 0]."""
    native = bool(getattr(lp_g, "device_native", False))
    padded_builtin = getattr(lp_g, "padded", None)          # GaussianTarget: the same target with a zero-padded precision matrix
    if callable(padded_builtin):
        return padded_builtin(D + 1)
    if native:
        def g(Xp, out=None):
            G = lp_g(Xp[:, :D].contiguous())        # (round-5 advice) a contiguous, 16-byte aligned (B, D) tensor like every even-D fit hands over
            Gp = eng.zeros(Xp.shape[0], D + 1) if out is None else out
            Gp[:, :D] = G
            if out is not None:
                Gp[:, D:] = 0.0
            return Gp
        g.device_native = True
        g.graph_safe = bool(getattr(lp_g, "graph_safe", False))
        return g

    def gh(xp):
        gv = np.asarray(lp_g(np.ascontiguousarray(xp[:, :D])))
        out = np.zeros((gv.shape[0], D + 1), dtype=np.float64)
        out[:, :D] = gv
        return out
    return gh


def wrap_monitor(monitor, lp, D):
    """The user's monitor sees the ORIGINAL problem: (mean, cov) without the inert coordinate and the original lp."""
    if monitor is None:
        return None

    class _Mon:
        checkpoint = monitor.checkpoint
        device_native = bool(getattr(monitor, "device_native", False))

        def __call__(self, i, params, _lp, key, nevals=1):
            m, c = params[0][:D], params[1][:D, :D]
            if isinstance(c, torch.Tensor):
                m, c = m.contiguous(), c.contiguous()
            else:
                m, c = np.ascontiguousarray(m), np.ascontiguousarray(c)
            return monitor(i, [m, c], lp, key, nevals=nevals)
    return _Mon()


def applies(eng, D, sampler, forced_samples):
    """odd D, the device sampler (the legacy SVD sampler and teacher-forced samples keep the literal D), a HIP engine"""
    return D % 2 == 1 and sampler == "cholesky" and forced_samples is None and getattr(eng, "name", "") == "hip"
