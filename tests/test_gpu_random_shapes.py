"""Randomised parity of the dense update through the public API (gsm_numpy.py:27-55): hypothesis draws D, B, the
input dtype, the memory layout (row strides, offsets) and the container (numpy / torch device tensors); every
draw is checked against the pinned oracle, for purity of the inputs and for the reference's dtype contract
(float32 in -> float64 out, gsm_numpy.py:47)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

from conftest import rel_err

pytestmark = pytest.mark.gpu


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(D=st.integers(1, 200), B=st.integers(1, 40), seed=st.integers(0, 10_000), f32=st.booleans(),
       pad=st.integers(0, 3), off=st.integers(0, 2), as_torch=st.booleans())
def test_random_shapes_layouts_dtypes(D, B, seed, f32, pad, off, as_torch):
    import torch
    import gsmvi_amd
    from oracle import gsm_oracle as orc
    s = orc.make_update_state(D, B, seed)
    dt = np.float32 if f32 else np.float64
    X, G, mu0, S0 = (np.asarray(s[k], dtype=dt) for k in ("samples", "vs", "mu0", "S0"))
    S0 = 0.5 * (S0 + S0.T)                                     # exactly symmetric also after the float32 cast
    mu_o, S_o = orc.gsm_update_batched(X.astype(np.float64), G.astype(np.float64), mu0.astype(np.float64),
                                       S0.astype(np.float64))

    def laid_out(a):                                           # a view into a larger buffer: row stride D + pad, offset
        if a.ndim == 1:
            return a
        buf = np.zeros((a.shape[0], a.shape[1] + pad + off), dtype=a.dtype)
        buf[:, off:off + a.shape[1]] = a
        return buf[:, off:off + a.shape[1]]

    args = [laid_out(X), laid_out(G), mu0, laid_out(S0)]
    keep = [a.copy() for a in args]
    if as_torch:
        dev = [torch.as_tensor(np.ascontiguousarray(a)).cuda() for a in args]
        if pad + off:                                          # strided device views as well
            big = [torch.zeros(a.shape[0], a.shape[1] + pad + off, dtype=a.dtype, device="cuda") if a.dim() == 2 else a
                   for a in dev]
            for b_, a in zip(big, dev):
                if a.dim() == 2:
                    b_[:, off:off + a.shape[1]] = a
            dev = [b_[:, off:off + a.shape[1]] if a.dim() == 2 else a for b_, a in zip(big, dev)]
        mu, S = gsmvi_amd.gsm_update(*dev)
        assert isinstance(mu, torch.Tensor) and mu.dtype == torch.float64 and S.dtype == torch.float64
        mu, S = mu.cpu().numpy(), S.cpu().numpy()
    else:
        mu, S = gsmvi_amd.gsm_update(*args)
        assert mu.dtype == np.float64 and S.dtype == np.float64           # gsm_numpy.py:47
        for a, k in zip(args, keep):
            assert np.array_equal(a, k)                                   # inputs untouched
    assert mu.shape == (D,) and S.shape == (D, D)
    tol = 1e-10
    assert rel_err(mu, mu_o) < tol and rel_err(S, S_o) < tol, (D, B, seed, f32, pad, off, as_torch)


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(B=st.integers(1, 64), extra=st.integers(0, 200), seed=st.integers(0, 10_000), reg=st.sampled_from([0.3, 1.0, 7.0]),
       pad=st.integers(0, 2))
def test_random_shapes_of_the_factor_forms(B, extra, seed, reg, pad):
    """Both factor-form updates (GSM: gsm_numpy.py:27-55, BaM: bam.py:72-114, on Sigma = F^T F) at random (D, B) with
    2B <= D -- every route of the 2B x 2B chain (rider / stand-alone launch, one workgroup / the 128-row kernels, folded
    and generic update kernels, D not a multiple of 64, strided factor) -- against the DENSE device updates on F0^T F0."""
    import torch
    import gsmvi_amd
    eng = gsmvi_amd.get_engine()
    D = 2 * B + extra
    rs = np.random.RandomState(seed)
    F0 = rs.standard_normal((D, D)) / np.sqrt(D) + 0.6 * np.eye(D)
    mu0 = rs.standard_normal(D)
    Z = rs.standard_normal((B, D))
    X = mu0 + Z @ F0
    A = rs.standard_normal((D, D)) / np.sqrt(D)
    P = A @ A.T + 0.4 * np.eye(D)
    G = -(X - rs.standard_normal(D)) @ P
    Fbuf = torch.zeros(D, D + pad, dtype=torch.float64, device="cuda")
    Fbuf[:, :D] = eng.asarray(F0)
    F0d = Fbuf[:, :D]                                           # row stride D + pad
    Zd, Xd, Gd, mud = (eng.asarray(a) for a in (Z, X, G, mu0))
    S0 = eng.asarray(F0.T @ F0)
    mu_d, S_d = eng.gsm_update(Xd, Gd, mud, S0)
    eng.last_path()
    mu_f, F, flag = eng.gsm_factor_update(Zd, Xd, Gd, mud, F0d)
    path = eng.last_path()
    if D % 2 == 0 and pad % 2 == 0:          # even D, even row stride: the tuned kernels (round 5), whatever B is
        assert not [k for k in path if k.endswith("_generic")], (D, B, pad, path)
    assert eng.read_flag(flag) == 0
    Fn = F.cpu().numpy()
    assert rel_err(Fn.T @ Fn, S_d.cpu().numpy()) < 1e-10 and rel_err(mu_f.cpu().numpy(), mu_d.cpu().numpy()) < 1e-10, (D, B)
    mu_bd, S_bd, _ = eng.bam_update(Xd, Gd, mud, S0, reg, 0.0)
    mu_bf, Fb, flag = eng.bam_factor_update(Zd, Xd, Gd, mud, F0d, reg)
    assert eng.read_flag(flag) == 0
    Fbn = Fb.cpu().numpy()
    assert rel_err(Fbn.T @ Fbn, S_bd.cpu().numpy()) < 1e-8 and rel_err(mu_bf.cpu().numpy(), mu_bd.cpu().numpy()) < 1e-8, (D, B, reg)


@pytest.mark.parametrize("D,B", [(256, 8), (1024, 32), (1024, 64), (4096, 64), (200, 20), (320, 48)])
def test_no_kernel_reads_outside_its_inputs(D, B):
    """Every input of the four update families sits inside a NaN-filled buffer (NaN rows in front, behind, and in the row
    padding): a kernel that reads one element outside an input -- instead of clamping to a valid one -- poisons its output.
    Covers the fast families (rider, wide panels, side-stream fork at D = 4096, the one-launch BaM chain) and ragged shapes."""
    import torch
    import gsmvi_amd
    eng = gsmvi_amd.get_engine()
    rs = np.random.RandomState(D * 7 + B)

    def guarded(a, pad=2):                       # a 2-D (or 1-D) array as a view into a NaN-filled device buffer
        a = np.atleast_2d(a)
        buf = torch.full((a.shape[0] + 2, a.shape[1] + pad), float("nan"), dtype=torch.float64, device="cuda")
        buf[1:-1, :a.shape[1]] = eng.asarray(a)
        return buf[1:-1, :a.shape[1]]

    F0 = rs.standard_normal((D, D)) / np.sqrt(D) + 0.6 * np.eye(D)
    mu0 = rs.standard_normal(D)
    Z = rs.standard_normal((B, D))
    X = mu0 + Z @ F0
    A = rs.standard_normal((D, D)) / np.sqrt(D)
    P = A @ A.T + 0.4 * np.eye(D)
    m = rs.standard_normal(D)
    G = -(X - m) @ P
    S0 = F0.T @ F0
    S0 = 0.5 * (S0 + S0.T)
    Zg, Xg, Gg, Fg, Sg, Pg = (guarded(a) for a in (Z, X, G, F0, S0, P))
    mug, mg = guarded(mu0)[0], guarded(m)[0]
    outs = []
    outs += list(eng.gsm_update(Xg, Gg, mug, Sg))
    outs += list(eng.bam_update(Xg, Gg, mug, Sg, 1.0, 0.0)[:2])
    outs.append(eng.sample(Zg, mug, Fg))
    outs.append(eng.gaussian_score(Xg, mg, Pg))
    outs.append(eng.potrf(Sg)[0])
    if 2 * B <= min(D, 128):
        outs += list(eng.gsm_factor_update(Zg, Xg, Gg, mug, Fg)[:2])
        outs += list(eng.bam_factor_update(Zg, Xg, Gg, mug, Fg, 1.0)[:2])
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        assert bool(torch.isfinite(o).all()), (D, B, k)
