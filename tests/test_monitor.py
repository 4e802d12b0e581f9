"""KLMonitor (gsm-vi_amd/monitors.py; reference gsmvi/monitors.py:43-125): protocol, bookkeeping and the KL
estimators against closed forms.  CPU only; the monitor is host-side diagnostics."""
import numpy as np

from gsmvi_amd.monitors import KLMonitor, mvn_logpdf, reverse_kl, forward_kl
from gsmvi_amd.gsm import GSM
from oracle import gsm_oracle as orc
from engines import OracleEngine


def _kl_gauss(m0, S0, m1, S1):
    """KL(N0 || N1) closed form."""
    D = m0.shape[0]
    iS1 = np.linalg.inv(S1)
    d = m1 - m0
    return 0.5 * (np.trace(iS1 @ S0) + d @ iS1 @ d - D + np.linalg.slogdet(S1)[1] - np.linalg.slogdet(S0)[1])


def test_logpdf_and_estimators_against_closed_form():
    rs = np.random.RandomState(0)
    D = 4
    m, cov_t, P = orc.make_gaussian_target(D, 1)
    mq, Sq = m + 0.3, cov_t * 1.5 + 0.1 * np.eye(D)
    lp = lambda x: mvn_logpdf(x, m, cov_t)                   # normalised target, per-sample values
    lq = lambda x: mvn_logpdf(x, mq, Sq)
    xq = rs.multivariate_normal(mq, Sq, size=200000)
    xp = rs.multivariate_normal(m, cov_t, size=200000)
    assert abs(reverse_kl(xq, lq, lp) - _kl_gauss(mq, Sq, m, cov_t)) < 2e-2
    assert abs(forward_kl(xp, lq, lp) - _kl_gauss(m, cov_t, mq, Sq)) < 2e-2
    from scipy.stats import multivariate_normal
    assert np.allclose(mvn_logpdf(xq[:50], mq, Sq), multivariate_normal(mq, Sq).logpdf(xq[:50]))


def test_monitor_protocol_and_bookkeeping():
    D = 5
    m, cov_t, P = orc.make_gaussian_target(D, 2)
    ref = np.random.RandomState(3).multivariate_normal(m, cov_t, size=1000)
    mon = KLMonitor(batch_size_kl=64, checkpoint=50, offset_evals=7, ref_samples=ref)
    lp = lambda x: mvn_logpdf(np.asarray(x), m, cov_t)
    gsm = GSM(D, lp, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine())
    gsm.fit(99, niter=200, batch_size=2, verbose=False, monitor=mon)
    assert len(mon.rkl) == len(mon.fkl) == len(mon.nevals) == 6          # i = 0,50,100,150,200 + final call
    assert mon.nevals[0] == 7 + 1 and mon.nevals[1] == mon.nevals[0] + 100   # offset accumulates (monitors.py:122-123)
    assert mon.rkl[0] > 1.0 and abs(mon.rkl[-1]) < 1e-6 and abs(mon.fkl[-1]) < 1e-6  # converged: KL -> 0
    mon.reset(offset_evals=0)
    assert mon.rkl == [] and mon.offset_evals == 0
    bad = KLMonitor(batch_size_kl=4, checkpoint=1)
    bad(0, [np.zeros(2), -np.eye(2)], lp, 5, nevals=3)                   # non-PD covariance -> NaN, not an exception
    assert np.isnan(bad.rkl[0]) and np.isnan(bad.fkl[0]) and bad.nevals == [3]


def test_device_monitor_host_logic_with_the_oracle_engine():
    """DeviceKLMonitor's bookkeeping and estimators with the oracle-backed test engine (the HIP kernels behind
    potrf / randn / sample / whiten_rows are checked on the GPU in tests/test_gpu_aux.py)."""
    from gsmvi_amd.monitors import DeviceKLMonitor
    D = 4
    m, cov_t, P = orc.make_gaussian_target(D, 1)
    mq, Sq = m + 0.3, cov_t * 1.5 + 0.1 * np.eye(D)
    lp = lambda x: mvn_logpdf(np.asarray(x), m, cov_t)
    ref = np.random.RandomState(3).multivariate_normal(m, cov_t, size=200000)
    mon = DeviceKLMonitor(batch_size_kl=100000, checkpoint=1, offset_evals=2, ref_samples=ref, engine=OracleEngine())
    mon(0, [mq, Sq], lp, 11, nevals=5)
    assert abs(mon.rkl[0] - _kl_gauss(mq, Sq, m, cov_t)) < 3e-2
    assert abs(mon.fkl[0] - _kl_gauss(m, cov_t, mq, Sq)) < 3e-2
    assert mon.nevals == [7] and mon.offset_evals == 7
    mon(1, [mq, -np.eye(D)], lp, 11, nevals=1)
    assert np.isnan(mon.rkl[1]) and mon.nevals == [7, 8]
    gsm = GSM(D, lp, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine())
    mon2 = DeviceKLMonitor(batch_size_kl=32, checkpoint=50, engine=OracleEngine())
    gsm.fit(99, niter=200, batch_size=2, verbose=False, monitor=mon2)
    assert len(mon2.rkl) == 6 and abs(mon2.rkl[-1]) < 1e-6 and np.isnan(mon2.fkl[-1])
