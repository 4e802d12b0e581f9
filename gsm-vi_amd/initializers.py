"""Initialisers for the fit drivers (off the hot path, host side).

``lbfgs_init`` follows gsmvi/initializers.py:5-17: maximise ``lp`` with scipy's L-BFGS-B and hand back the
optimum together with the optimiser's dense inverse-Hessian estimate as the starting (mean, cov) of
``GSM.fit`` / ``BaM.fit``, plus the scipy result object.  Additions: ``lp`` / ``lp_g`` may be the
device-native callables of this package (they are fed a (1, D) CUDA tensor then), and ``lp`` may return a
one-element array instead of a scalar.
"""
import numpy as np
from scipy.optimize import minimize


def _host_callable(fn, D, vector):
    """Wrap ``fn`` (numpy or device-native, taking (D,) or (1, D)) as a float64 numpy function of a (D,) point."""
    native = getattr(fn, "device_native", False)

    def call(x):
        x = np.asarray(x, dtype=np.float64)
        if native:
            import torch
            out = fn(torch.as_tensor(x[None, :], device="cuda"))
            out = out.detach().to("cpu").numpy()
        else:
            out = fn(x)
            if hasattr(out, "detach"):
                out = out.detach().to("cpu").numpy()
            out = np.asarray(out, dtype=np.float64)
        return out.reshape(D) if vector else float(out.reshape(-1)[0])
    return call


def lbfgs_init(x0, lp, lp_g=None, maxiter=1000, maxfun=1000):
    """Returns ``(mu, cov, res)``: the L-BFGS-B maximiser of ``lp`` started at ``x0``, the dense inverse
    Hessian approximation at it, and the ``scipy.optimize.OptimizeResult`` (gsmvi/initializers.py:5-17).
    Without ``lp_g`` scipy differentiates numerically, as the reference does."""
    x0 = np.asarray(x0, dtype=np.float64)
    D = x0.shape[0]
    value = _host_callable(lp, D, vector=False)
    neg = lambda x: -value(x)
    jac = None
    if lp_g is not None:
        score = _host_callable(lp_g, D, vector=True)
        jac = lambda x: -score(x)
    res = minimize(neg, x0, method="L-BFGS-B", jac=jac, options={"maxiter": maxiter, "maxfun": maxfun})
    return res.x, res.hess_inv.todense(), res
