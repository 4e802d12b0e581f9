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


@dataclass
class DeviceKLMonitor(KLMonitor):
    """KLMonitor (gsmvi/monitors.py:43-125) whose sampling and Gaussian log density run on the GPU.

    ``device_native = True`` makes the fit drivers hand over the device tensors ``[mean, cov]`` (no D x D copy to
    the host, no numpy SVD sampler).  Per call: ``gsmvi_potrf_f64`` (upper factor R of cov), q-samples
    x = mean + z R from the counter-based draw stream (``gsmvi_randn_f64`` + ``gsmvi_sample_f64``), log q from the
    whitened residuals and the factor's diagonal (``gsmvi_whiten_rows_f64``); only ``batch_size_kl`` x D numbers
    come back to the host for the final sums.  ``lp`` gets the device tensor first and numpy if it refuses it.
    Same fields, bookkeeping and NaN-on-exception behaviour as the reference's monitor.
    """
    engine: object = None
    device_native = True
    _CHUNK = 128

    def _sum_lp(self, lp, x_dev, eng):
        try:
            return float(np.sum(_to_numpy(lp(x_dev))))
        except (TypeError, AttributeError, RuntimeError, ValueError):
            return float(np.sum(_to_numpy(lp(eng.to_numpy(x_dev)))))

    def _sum_logq(self, eng, X, mu, R):
        Z, logdiag = eng.whiten_rows(X, mu, R)
        Zh = eng.to_numpy(Z)
        n, D = Zh.shape
        return -0.5 * float(np.sum(Zh * Zh)) - n * float(eng.to_numpy(logdiag).reshape(-1)[0]) \
            - 0.5 * n * D * np.log(2 * np.pi)

    def __call__(self, i, params, lp, key, nevals=1):
        if self.engine is None:
            from .engine import get_engine
            self.engine = get_engine()
        eng = self.engine
        if self._rs is None:
            self._seed = int(np.asarray(_to_numpy(key)).flatten()[-1]) % (2 ** 32)
            self._rs = np.random.RandomState(self._seed)
            self._calls = 0
        try:
            mu, cov = eng.asarray(params[0]), eng.asarray(params[1])
            D, n = int(mu.shape[0]), int(self.batch_size_kl)
            R, flag = eng.potrf(cov)
            if eng.read_flag(flag) != 0:
                raise ValueError("covariance is not positive definite")
            Z = eng.normal(n, D, self._seed ^ 0x5DEECE66D, self._calls)
            self._calls += 1
            acc = 0.0
            for s0 in range(0, n, self._CHUNK):              # bounded engine workspace whatever batch_size_kl is
                X = eng.sample(Z[s0:s0 + self._CHUNK], mu, R)
                acc += self._sum_logq(eng, X, mu, R) - self._sum_lp(lp, X, eng)
            self.rkl.append(acc / n)
            if self.ref_samples is not None:
                ref = _to_numpy(self.ref_samples)
                idx = self._rs.permutation(ref.shape[0])[:n]
                acc = 0.0
                for s0 in range(0, len(idx), self._CHUNK):
                    Xp = eng.asarray(np.ascontiguousarray(ref[idx[s0:s0 + self._CHUNK]]))
                    acc += self._sum_lp(lp, Xp, eng) - self._sum_logq(eng, Xp, mu, R)
                self.fkl.append(acc / len(idx))
            else:
                self.fkl.append(float("nan"))
        except Exception as e:                                         # noqa: BLE001 (reference behaviour)
            print(f"Exception occured in monitor : {e}.\nAppending NaN")
            self.rkl.append(float("nan"))
            self.fkl.append(float("nan"))
        self.nevals.append(self.offset_evals + nevals)
        self.offset_evals = self.nevals[-1]
        return key
