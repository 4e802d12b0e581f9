"""KLMonitor: the diagnostics callback of the reference (gsmvi/monitors.py:43-125) without JAX/numpyro.

Off the hot path (called every ``checkpoint`` iterations by ``GSM.fit`` / ``BaM.fit`` with host copies of
(mean, cov)); plain numpy.  Same fields, same call protocol ``monitor(i, [mean, cov], lp, key, nevals=n)``,
same bookkeeping (``rkl``, ``fkl``, ``nevals`` lists; ``offset_evals`` accumulates).  Deviations: the
JAX key split + numpy re-seed (monitors.py:101-102) is replaced by a private ``RandomState`` seeded from
the key on first use; ``np.NaN`` (removed in numpy 2) is ``float('nan')``.
"""
from dataclasses import dataclass

import numpy as np


def _to_numpy(a):
    try:
        import torch
        if isinstance(a, torch.Tensor):
            return a.detach().to("cpu").numpy()
    except ImportError:
        pass
    return np.asarray(a)


def _sum_lp(lp, samples):
    """np.sum(lp(samples)) as in monitors.py:11,18; tries numpy input first, then a CUDA tensor for
    device-native log-probabilities (GaussianTarget.lp)."""
    try:
        return float(np.sum(_to_numpy(lp(samples))))
    except (TypeError, AttributeError, RuntimeError):
        import torch
        return float(np.sum(_to_numpy(lp(torch.as_tensor(samples, device="cuda")))))


def mvn_logpdf(x, mean, cov):
    """Row-wise log N(x; mean, cov) (what numpyro's MultivariateNormal.log_prob returns, monitors.py:107)."""
    L = np.linalg.cholesky(cov)
    r = np.linalg.solve(L, (x - mean).T)
    D = mean.shape[0]
    return -0.5 * np.sum(r * r, axis=0) - np.sum(np.log(np.diag(L))) - 0.5 * D * np.log(2 * np.pi)


def reverse_kl(samples, lpq, lpp):
    """monitors.py:10-15."""
    return (float(np.sum(lpq(samples))) - _sum_lp(lpp, samples)) / samples.shape[0]


def forward_kl(samples, lpq, lpp):
    """monitors.py:17-22."""
    return (_sum_lp(lpp, samples) - float(np.sum(lpq(samples)))) / samples.shape[0]


@dataclass
class KLMonitor:
    """Monitor reverse (and forward) KL divergence during the fit (gsmvi/monitors.py:43-67)."""
    batch_size_kl: int = 8
    checkpoint: int = 20
    offset_evals: int = 0
    ref_samples: np.ndarray = None

    def __post_init__(self):
        self.rkl = []
        self.fkl = []
        self.nevals = []
        self._rs = None

    def reset(self, batch_size_kl=None, checkpoint=None, offset_evals=None, ref_samples=None):
        """monitors.py:69-81."""
        self.nevals, self.rkl, self.fkl = [], [], []
        if batch_size_kl is not None:
            self.batch_size_kl = batch_size_kl
        if checkpoint is not None:
            self.checkpoint = checkpoint
        if offset_evals is not None:
            self.offset_evals = offset_evals
        if ref_samples is not None:
            self.ref_samples = ref_samples
        print("offset evals reset to : ", self.offset_evals)

    def __call__(self, i, params, lp, key, nevals=1):
        """monitors.py:83-125."""
        mu, cov = _to_numpy(params[0]), _to_numpy(params[1])
        if self._rs is None:
            self._rs = np.random.RandomState(int(np.asarray(_to_numpy(key)).flatten()[-1]) % (2 ** 32))
        try:
            qsamples = self._rs.multivariate_normal(mean=mu, cov=cov, size=self.batch_size_kl)
            lpq = lambda x: mvn_logpdf(x, mu, cov)                     # noqa: E731
            self.rkl.append(reverse_kl(qsamples, lpq, lp))
            if self.ref_samples is not None:
                ref = _to_numpy(self.ref_samples)
                idx = self._rs.permutation(ref.shape[0])[:self.batch_size_kl]
                self.fkl.append(forward_kl(ref[idx], lpq, lp))
            else:
                self.fkl.append(float("nan"))
        except Exception as e:                                         # noqa: BLE001 (reference behaviour)
            print(f"Exception occured in monitor : {e}.\nAppending NaN")
            self.rkl.append(float("nan"))
            self.fkl.append(float("nan"))
        self.nevals.append(self.offset_evals + nevals)
        self.offset_evals = self.nevals[-1]
        return key
