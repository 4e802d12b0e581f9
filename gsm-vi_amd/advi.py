"""ADVI comparison harness: full-rank Gaussian fit by stochastic ELBO maximisation (gsmvi/advi.py:8-112).

Off the hot path: this is the baseline the reference's examples compare GSM against
(examples/example_initializers.py:39-65), restated on torch autograd in place of JAX + optax + numpyro.
The variational family is N(loc, L L^T) with L lower triangular, optimised through its D(D+1)/2 free
entries; the ELBO estimate uses reparameterised draws x = loc + z L^T (advi.py:34-47).

Differences from the reference, all forced by the missing JAX stack:
  * ``opt`` is a factory ``params -> torch.optim.Optimizer`` (e.g. ``lambda p: torch.optim.Adam(p, lr=1e-2)``)
    or an optimiser class, instead of an optax transformation;
  * ``key`` is an int seed for a ``torch.Generator`` (a fresh draw per iteration, where the reference splits
    a PRNGKey per iteration, advi.py:103);
  * ``lp`` maps a float64 tensor (B, D) on ``device`` to per-sample log-probabilities (summed as advi.py:43)
    and must be differentiable by torch autograd;
  * the ``niter < nprint`` division by zero of advi.py:92 is guarded as in bam.py:177.
"""
import math

import numpy as np
import torch


class ADVI:
    def __init__(self, D, lp, device=None):
        self.D = D
        self.lp = lp
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self._rows, self._cols = torch.tril_indices(D, D, device=self.device)

    def _scale_tril(self, scales):
        L = torch.zeros(self.D, self.D, dtype=scales.dtype, device=scales.device)
        return L.index_put((self._rows, self._cols), scales)

    def scales_to_cov(self, scales):
        """Free Cholesky entries -> covariance L L^T (advi.py:23-27)."""
        L = self._scale_tril(torch.as_tensor(scales, dtype=torch.float64, device=self.device))
        return (L @ L.T).detach().to("cpu").numpy()

    def neg_elbo(self, params, gen, batch_size):
        """-(sum_b lp(x_b) - sum_b log q(x_b)) over reparameterised draws (advi.py:29-47)."""
        loc, scales = params
        L = self._scale_tril(scales)
        z = torch.randn(batch_size, self.D, dtype=torch.float64, device=self.device, generator=gen)
        x = loc[None, :] + z @ L.T
        logl = self.lp(x).sum()
        # log q(x_b) = -1/2 |z_b|^2 - sum log|L_ii| - D/2 log(2 pi)
        logq = -0.5 * (z * z).sum() - batch_size * (torch.log(torch.abs(torch.diagonal(L))).sum()
                                                   + 0.5 * self.D * math.log(2.0 * math.pi))
        return -(logl - logq)

    def fit(self, key, opt, mean=None, cov=None, batch_size=8, niter=1000, nprint=10, monitor=None):
        """Returns ``(mean, cov, losses)`` (advi.py:49-112); monitor protocol as GSM.fit."""
        D = self.D
        mean = np.zeros(D) if mean is None else np.asarray(mean, dtype=np.float64)
        cov = np.identity(D) if cov is None else np.asarray(cov, dtype=np.float64)
        L0 = np.linalg.cholesky(cov)
        loc = torch.tensor(mean, dtype=torch.float64, device=self.device, requires_grad=True)
        scales = torch.tensor(L0[np.tril_indices(D)], dtype=torch.float64, device=self.device, requires_grad=True)
        params = [loc, scales]
        optimizer = opt(params)
        gen = torch.Generator(device=self.device)
        gen.manual_seed(int(key))
        every = max(niter // nprint, 1) if nprint else 0
        losses, nevals = [], 1

        def snapshot():
            return loc.detach().to("cpu").numpy().copy(), self.scales_to_cov(scales.detach())

        for i in range(niter + 1):
            if every and i % every == 0:
                print(f"Iteration {i} of {niter}")
            if monitor is not None and i % monitor.checkpoint == 0:
                m, c = snapshot()
                monitor(i, [m, c], self.lp, key, nevals=nevals)
                nevals = 0
            optimizer.zero_grad(set_to_none=True)
            loss = self.neg_elbo(params, gen, batch_size)
            loss.backward()
            optimizer.step()
            losses.append(float(loss.detach()))
            nevals += batch_size
        m, c = snapshot()
        if monitor is not None:
            monitor(i, [m, c], self.lp, key, nevals=nevals)
        return m, c, losses
