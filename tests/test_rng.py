"""Counter-based whitened-draw stream (csrc/gsmvi_rng.hip, replaces the z-stream of gsm_numpy.py:105,116).

CPU: the oracle's Philox4x32-10 restatement against the Random123 known-answer vectors, distribution checks,
statelessness, and the fit drivers using rng="device" through the oracle-backed engine.
GPU: the kernel's Philox words bit-exact against the restatement, normals to 1e-13, ragged lengths."""
import numpy as np
import pytest

from oracle import gsm_oracle as orc

# Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key, expected)
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_known_answer_vectors():
    for ctr, key, want in KAT:
        got = orc.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(x) for x in got) == want


def test_stream_is_a_pure_function_of_seed_call_index():
    z = orc.philox_randn(7, 3, 1001)
    assert np.array_equal(z, orc.philox_randn(7, 3, 1001))
    assert np.array_equal(z[:500], orc.philox_randn(7, 3, 500))          # prefix property (pairs are independent)
    assert not np.array_equal(z, orc.philox_randn(7, 4, 1001))
    assert not np.array_equal(z, orc.philox_randn(8, 3, 1001))
    assert np.isfinite(z).all()


def test_stream_is_standard_normal():
    z = orc.philox_randn(2024, 0, 400_000)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1.0) < 5e-3
    assert abs(np.mean(z ** 3)) < 2e-2 and abs(np.mean(z ** 4) - 3.0) < 5e-2
    assert abs(np.corrcoef(z[0::2], z[1::2])[0, 1]) < 5e-3               # the two Box-Muller outputs of a pair
    from scipy import stats
    assert stats.kstest(z[:50_000], "norm").pvalue > 1e-3


def test_fit_with_device_stream_on_the_host_engine():
    """rng='device' in GSM.fit / BaM.fit: Z comes from engine.normal(seed=key, call=iteration)."""
    from engines import OracleEngine
    from gsmvi_amd.gsm import GSM
    from gsmvi_amd.bam import BaM, Regularizers
    D, B = 8, 4
    m, cov_t, P = orc.make_gaussian_target(D, 3)
    seen = []

    def lp_g(x):
        seen.append(np.array(x))
        return orc.gaussian_score(x, m, P)
    mean, cov = GSM(D, None, lp_g, engine=OracleEngine()).fit(11, niter=300, batch_size=B, verbose=False, rng="device")
    assert np.allclose(seen[0], orc.philox_randn(11, 0, B * D).reshape(B, D))       # mean 0, cov I at iteration 0
    assert np.abs(mean - m).max() < 1e-6 and np.abs(cov - cov_t).max() / np.abs(cov_t).max() < 1e-6
    mean2, cov2 = GSM(D, None, lp_g, engine=OracleEngine()).fit(11, niter=300, batch_size=B, verbose=False,
                                                                rng="device", method="factor")
    assert np.abs(mean2 - m).max() < 1e-6 and np.abs(cov2 - cov_t).max() / np.abs(cov_t).max() < 1e-6
    reg = Regularizers()
    mean3, cov3 = BaM(D, None, lp_g, engine=OracleEngine()).fit(11, reg.constant(50.0), niter=60, batch_size=B,
                                                               verbose=False, rng="device")
    assert np.abs(mean3 - m).max() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 7, 64, 32 * 1024, 100_003])
def test_gpu_randn_matches_restatement(n):
    import torch
    import gsmvi_amd
    eng = gsmvi_amd.get_engine()
    seed, call = 0x1234_5678_9abc_def0, 5 + (1 << 33)
    raw = torch.zeros(4 * ((n + 1) // 2), dtype=torch.int32, device=eng.device)
    out = eng.empty(1, n)
    Z = eng.normal(1, n, seed, call, out=out, raw=raw)
    z_ref, w_ref = orc.philox_randn(seed, call, n, return_raw=True)
    assert np.array_equal(raw.cpu().numpy().view(np.uint32).reshape(-1, 4), w_ref)   # integer part: bit-exact
    z = Z.cpu().numpy().reshape(-1)
    assert np.abs(z - z_ref).max() < 1e-13                                            # fp64 log / sincos round-off
    assert np.array_equal(z, eng.normal(1, n, seed, call).cpu().numpy().reshape(-1)) # replayable


@pytest.mark.gpu
def test_gpu_fit_with_device_stream_converges_and_is_reproducible():
    import gsmvi_amd
    D, B = 16, 8
    m, cov_t, P = orc.make_gaussian_target(D, 1)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    out = []
    for method in ("dense", "factor", "dense"):
        gsm = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
        out.append(gsm.fit(5, niter=600, batch_size=B, verbose=False, rng="device", method=method))
        assert gsm.n_reverts == 0
    for mean, cov in out:
        assert np.abs(mean - m).max() < 1e-6 and np.abs(cov - cov_t).max() / np.abs(cov_t).max() < 1e-6
    assert np.array_equal(out[0][0], out[2][0]) and np.array_equal(out[0][1], out[2][1])   # same key, same fit
