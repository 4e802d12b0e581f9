"""Sizes beyond BASELINE's largest configuration (D = 4096): D = 6144 ... 12288, where an index that overflowed 32 bits or a
workspace sized for the benchmark shapes would show.  No oracle run at these sizes (numpy would take minutes): the checks are
the size-independent identities the domain offers -- the factor form and the dense form of the SAME update are two different
kernel chains and must agree (F^T F = S, same mean) for GSM and for BaM, and the Cholesky factor must reproduce its matrix."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D,B", [(6144, 32), (8192, 64), (12288, 16)])
def test_dense_and_factor_forms_agree_beyond_the_baseline_sizes(D, B):
    import torch
    import gsmvi_amd
    eng = gsmvi_amd.get_engine()
    dev = eng.device
    g = torch.Generator(device=dev)
    g.manual_seed(D)
    kw = dict(dtype=torch.float64, device=dev, generator=g)
    F0 = torch.randn(D, D, **kw) / D ** 0.5 + 0.7 * torch.eye(D, dtype=torch.float64, device=dev)
    mu0 = torch.randn(D, **kw)
    Z = torch.randn(B, D, **kw)
    X = mu0 + Z @ F0                                             # the factor forms' contract: x_b = mu0 + z_b F0
    m, prec = torch.rand(D, **kw), 0.5 + torch.rand(D, **kw)     # a diagonal-precision Gaussian target
    G = -(X - m) * prec
    S0 = F0.T @ F0
    S0 = 0.5 * (S0 + S0.T)
    mu, S = eng.gsm_update(X, G, mu0, S0)
    mu_f, F, fl = eng.gsm_factor_update(Z, X, G, mu0, F0)
    assert eng.read_flag(fl) == 0
    assert rel_err(eng.gram(F).cpu().numpy(), S.cpu().numpy()) < 1e-12
    assert rel_err(mu_f.cpu().numpy(), mu.cpu().numpy()) < 1e-12
    del S, F
    mu_d, S_d, fd = eng.bam_update(X, G, mu0, S0, 1.0, 0.0)
    mu_b, Fb, fb = eng.bam_factor_update(Z, X, G, mu0, F0, 1.0)
    assert eng.read_flag(fd) == 0 and eng.read_flag(fb) == 0
    assert rel_err(eng.gram(Fb).cpu().numpy(), S_d.cpu().numpy()) < 1e-12
    assert rel_err(mu_b.cpu().numpy(), mu_d.cpu().numpy()) < 1e-11
    del S_d, Fb
    R, fp = eng.potrf(S0)
    assert eng.read_flag(fp) == 0
    assert float(((R.T @ R - S0).abs().max() / S0.abs().max()).item()) < 1e-13
    assert float(torch.tril(R, -1).abs().max().item()) == 0.0    # the strictly lower triangle is zero (the sampler reads all of R)
    del R, S0, F0
    torch.cuda.empty_cache()
