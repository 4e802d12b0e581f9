"""Score providers for the fit drivers.

``GaussianTarget`` is the benchmark target of the reference's examples
(examples/example_gsm_numpy.py:8-31: ``lp_g(x) = -icov (x - mean)``) evaluated by the HIP panel
kernel.  ``device_score`` marks a user callable as taking/returning CUDA tensors;
``score_from_logp`` is the sum-then-autograd helper matching the JAX examples
(examples/example_gsm.py:34-35: ``lp_g = jit(grad(lambda x: sum(lp(x))))``).
"""
import numpy as np
import torch

from .engine import get_engine


def device_score(fn=None, *, graph_safe=False):
    """Decorator: ``fn`` maps a float64 CUDA tensor (B,D) to a float64 CUDA tensor (B,D).
    ``@device_score(graph_safe=True)`` additionally promises that a call is capturable into a hipGraph (only stream-ordered
    device work on the current stream, no host synchronisation, deterministic launch sequence): the factor-form fit then
    replays blocks of iterations as one graph instead of issuing every launch from Python."""
    def mark(f):
        f.device_native = True
        f.graph_safe = bool(graph_safe)
        return f
    return mark(fn) if fn is not None else mark


def score_from_logp(logp, graph_safe=False):
    """Score via torch autograd of a *summed* log-probability (examples/example_gsm.py:34-35: jit(grad(sum lp))).
    ``graph_safe=True`` promises that ``logp`` is capturable (stream-ordered torch ops of fixed shapes, no host
    synchronisation, no data-dependent control flow): the factor-form fit then records forward AND backward of every
    iteration of a block into its hipGraph -- the ~12 torch dispatches per score evaluation (90 - 160 us of host time at
    D = 1024) replay as part of one graph launch, which is what jit does for the reference's JAX score."""
    def lp_g(x):
        xg = x.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            total = logp(xg).sum()
            (g,) = torch.autograd.grad(total, xg)
        return g.detach()
    lp_g.device_native = True
    lp_g.graph_safe = bool(graph_safe)
    return lp_g


class GaussianTarget:
    """N(mean, cov) with device-resident precision matrix; ``lp`` / ``lp_g`` follow
    examples/example_gsm_numpy.py:17-29."""

    def __init__(self, mean, cov=None, precision=None, engine=None):
        self.engine = engine if engine is not None else get_engine()
        eng = self.engine
        if precision is None:
            precision = np.linalg.inv(np.asarray(cov, dtype=np.float64))
        P = np.asarray(precision, dtype=np.float64)
        self.mean = eng.asarray(np.asarray(mean, dtype=np.float64))
        self.P = eng.asarray(0.5 * (P + P.T))
        self.D = int(self.mean.shape[0])

        def lp_g(x, out=None):
            return eng.gaussian_score(x, self.mean, self.P, out=out)
        lp_g.device_native = True
        lp_g.graph_safe = True          # one capturable kernel launch, no allocation when `out` is given, no host work

        def padded(Dp):
            """the score of the same target with (Dp - D) inert coordinates appended (zero rows / columns in the precision
            matrix: g' = [g, 0] for x' = [x, anything]) -- what the fit loops use for odd D (gsm-vi_amd/_oddpad.py)"""
            if getattr(self, "_padded", None) is None or self._padded[0] != Dp:
                mp, Pp = eng.zeros(Dp), eng.zeros(Dp, Dp)
                mp[:self.D] = self.mean
                Pp[:self.D, :self.D] = self.P

                def lp_g_p(x, out=None):
                    return eng.gaussian_score(x, mp, Pp, out=out)
                lp_g_p.device_native = True
                lp_g_p.graph_safe = True
                self._padded = (Dp, lp_g_p)
            return self._padded[1]
        lp_g.padded = padded
        self.lp_g = lp_g

    def lp(self, x):
        """sum_b -1/2 (m - x_b)^T P (m - x_b); monitor-only, so plain torch is fine here."""
        x = self.engine.asarray(x)
        r = self.mean[None, :] - x
        return -0.5 * torch.einsum("bi,ij,bj->", r, self.P, r)
