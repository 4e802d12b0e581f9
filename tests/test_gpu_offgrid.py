"""Off-grid shapes on the tuned kernels (round 5).  The reference takes any (D, B) (gsm_numpy.py:27-55, bam.py:31-114;
its own example is D = 5); until round 4 the tuned kernels here were gated to D % 64 == 0 and B in {8, 16, 32, 64, 128} and
everything else took the guarded round-1 family.  Now any EVEN D (even leading dimensions, 16-byte aligned bases) and any
batch size stay on the tuned kernels: parity against the pinned oracle, plus gsmvi_last_path() as the proof that no
guarded kernel ran.  Odd D: an engine call on the caller's own arrays keeps the guarded family (rows of an odd-D matrix are not
16-byte aligned); the Python drop-ins (gsm_update, bam_update, GSM.fit, BaM.fit) run the (D + 1)-dimensional problem with an
inert last coordinate on the tuned kernels (gsm-vi_amd/_oddpad.py)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-11
# the six shapes of profiles/r05/offgrid.json + small and awkward ones (D not a multiple of 16 / 32 / 64; B odd, tiny, > 64)
SHAPES = [(1000, 32), (1024, 20), (1000, 30), (784, 50), (500, 10), (2000, 24), (250, 6), (66, 3), (130, 65), (1022, 127),
          (98, 49), (36, 1), (2, 1)]


@pytest.fixture(scope="module")
def eng():
    import gsmvi_amd
    return gsmvi_amd.get_engine()


def _generic(path):
    return sorted(k for k in path if k.endswith("_generic"))


def _state(D, B, seed=0):
    rs = np.random.RandomState(seed + 17 * D + B)
    F0 = rs.standard_normal((D, D)) / np.sqrt(D) + 0.6 * np.eye(D)
    mu0 = rs.standard_normal(D)
    Z = rs.standard_normal((B, D))
    X = mu0 + Z @ F0
    A = rs.standard_normal((D, D)) / np.sqrt(D)
    P = A @ A.T + 0.4 * np.eye(D)
    m = rs.standard_normal(D)
    G = -(X - m) @ P
    S0 = F0.T @ F0
    return dict(F0=F0, mu0=mu0, Z=Z, X=X, G=G, S0=0.5 * (S0 + S0.T), m=m, P=0.5 * (P + P.T))


@pytest.mark.parametrize("D,B", SHAPES)
def test_dense_gsm_update_offgrid_runs_the_tuned_kernels(eng, D, B):
    """gsm_numpy.py:27-55 at off-grid (D, B): oracle parity and no guarded kernel."""
    from oracle import gsm_oracle as orc
    s = _state(D, B)
    mu_o, S_o = orc.gsm_update_batched(s["X"], s["G"], s["mu0"], s["S0"])
    d = {k: eng.asarray(s[k]) for k in ("X", "G", "mu0", "S0")}
    eng.last_path()
    mu, S = eng.gsm_update(d["X"], d["G"], d["mu0"], d["S0"])
    path = eng.last_path()
    assert rel_err(mu.cpu().numpy(), mu_o) < TOL and rel_err(S.cpu().numpy(), S_o) < TOL
    Sn = S.cpu().numpy()
    assert np.array_equal(Sn, Sn.T)                                  # exactly symmetric, edge tiles included
    if B <= 128:
        assert not _generic(path), path
        assert {"panel_fast", "scalars_fast", "cov_sym"} <= path, path


@pytest.mark.parametrize("D,B", SHAPES)
def test_bam_update_offgrid_runs_the_tuned_kernels(eng, D, B):
    """bam.py:72-114 at off-grid (D, B) against the scipy restatement (restatement-derived: BaM parity is unpinned, see
    DESIGN section 2) and against K8's defining equation on the device result; no guarded kernel."""
    from oracle import bam_oracle as borc
    s = _state(D, B)
    mu_o, S_o = borc.bam_lowrank_update_exact(s["X"], s["G"], s["mu0"], s["S0"], 2.0)
    d = {k: eng.asarray(s[k]) for k in ("X", "G", "mu0", "S0")}
    eng.last_path()
    mu, S, flag = eng.bam_update(d["X"], d["G"], d["mu0"], d["S0"], 2.0, 0.0)
    path = eng.last_path()
    assert eng.read_flag(flag) == 0
    assert rel_err(mu.cpu().numpy(), mu_o) < 1e-8 and rel_err(S.cpu().numpy(), S_o) < 1e-8
    Sn = S.cpu().numpy()
    assert np.array_equal(Sn, Sn.T)
    if 2 * B <= 288:
        assert not _generic(path), path
        assert "lowrank_fast" in path, path


@pytest.mark.parametrize("D,B", [s for s in SHAPES if 2 * s[1] <= min(s[0], 256)])
def test_factor_forms_offgrid_run_the_tuned_kernels(eng, D, B):
    """Both factor-form updates at off-grid (D, B) against the dense device updates on F0^T F0 (which the tests above tie to
    the oracle); no guarded kernel."""
    s = _state(D, B)
    d = {k: eng.asarray(s[k]) for k in ("Z", "X", "G", "mu0", "F0", "S0")}
    mu_d, S_d = eng.gsm_update(d["X"], d["G"], d["mu0"], d["S0"])
    eng.last_path()
    mu_f, F, flag = eng.gsm_factor_update(d["Z"], d["X"], d["G"], d["mu0"], d["F0"])
    path = eng.last_path()
    assert eng.read_flag(flag) == 0
    Fn = F.cpu().numpy()
    assert rel_err(Fn.T @ Fn, S_d.cpu().numpy()) < 1e-10 and rel_err(mu_f.cpu().numpy(), mu_d.cpu().numpy()) < 1e-10
    assert not _generic(path), path
    assert {"panel_t_fast", "fupd_fast"} <= path, path
    mu_bd, S_bd, _ = eng.bam_update(d["X"], d["G"], d["mu0"], d["S0"], 1.5, 0.0)
    eng.last_path()
    mu_bf, Fb, flag = eng.bam_factor_update(d["Z"], d["X"], d["G"], d["mu0"], d["F0"], 1.5)
    path = eng.last_path()
    assert eng.read_flag(flag) == 0
    Fbn = Fb.cpu().numpy()
    assert rel_err(Fbn.T @ Fbn, S_bd.cpu().numpy()) < 1e-8 and rel_err(mu_bf.cpu().numpy(), mu_bd.cpu().numpy()) < 1e-8
    assert not _generic(path), path


@pytest.mark.parametrize("D,B", [(1000, 30), (250, 6), (66, 3), (130, 65), (2000, 24)])
def test_offgrid_kernels_read_nothing_outside_their_inputs(eng, D, B):
    """Edge tiles re-read CLAMPED rows and columns: every input sits in a NaN-filled buffer (NaN rows in front and behind,
    NaN in the row padding), so one read outside an input poisons an output."""
    import torch
    s = _state(D, B)

    def guarded(a, pad=2):
        a = np.atleast_2d(a)
        buf = torch.full((a.shape[0] + 2, a.shape[1] + pad), float("nan"), dtype=torch.float64, device="cuda")
        buf[1:-1, :a.shape[1]] = eng.asarray(a)
        return buf[1:-1, :a.shape[1]]

    Zg, Xg, Gg, Fg, Sg, Pg = (guarded(s[k]) for k in ("Z", "X", "G", "F0", "S0", "P"))
    mug, mg = guarded(s["mu0"])[0], guarded(s["m"])[0]
    eng.last_path()
    outs = list(eng.gsm_update(Xg, Gg, mug, Sg)) + list(eng.bam_update(Xg, Gg, mug, Sg, 1.0, 0.0)[:2])
    outs += [eng.sample(Zg, mug, Fg), eng.gaussian_score(Xg, mg, Pg), eng.potrf(Sg)[0]]
    if 2 * B <= min(D, 256):
        outs += list(eng.gsm_factor_update(Zg, Xg, Gg, mug, Fg)[:2]) + list(eng.bam_factor_update(Zg, Xg, Gg, mug, Fg, 1.0)[:2])
    path = eng.last_path()
    torch.cuda.synchronize()
    assert not _generic(path), path
    for k, o in enumerate(outs):
        assert bool(torch.isfinite(o).all()), (D, B, k)


@pytest.mark.parametrize("D,B", [(1000, 30), (250, 6), (130, 65)])
def test_offgrid_outputs_are_written_only_inside_the_matrix(eng, D, B):
    """Edge tiles store only what lies inside: outputs that are views into larger buffers keep their surroundings."""
    import torch
    s = _state(D, B)
    d = {k: eng.asarray(s[k]) for k in ("Z", "X", "G", "mu0", "F0", "S0")}

    def framed():
        buf = torch.full((D + 2, D + 4), -7.0, dtype=torch.float64, device="cuda")
        return buf, buf[1:-1, 2:2 + D]

    for call in ("gsm", "bam", "gsmf", "bamf"):
        if call in ("gsmf", "bamf") and 2 * B > min(D, 256):
            continue
        buf, out = framed()
        mu = eng.empty(D)
        if call == "gsm":
            eng.gsm_update(d["X"], d["G"], d["mu0"], d["S0"], out=(mu, out))
        elif call == "bam":
            eng.bam_update(d["X"], d["G"], d["mu0"], d["S0"], 1.0, 0.0, out=(mu, out))
        elif call == "gsmf":
            eng.gsm_factor_update(d["Z"], d["X"], d["G"], d["mu0"], d["F0"], out=(mu, out))
        else:
            eng.bam_factor_update(d["Z"], d["X"], d["G"], d["mu0"], d["F0"], 1.0, out=(mu, out))
        torch.cuda.synchronize()
        frame = buf.clone()
        frame[1:-1, 2:2 + D] = -7.0
        assert bool((frame == -7.0).all()), call
        assert bool(torch.isfinite(out).all()) and not bool((out == -7.0).all()), call


def test_odd_d_engine_calls_keep_the_guarded_family(eng):
    """Rows of an odd-D matrix are not 16-byte aligned: an ENGINE call (a C caller's own arrays) on such a shape runs the
    guarded kernels (same arithmetic).  The Python drop-ins pad instead: tests below."""
    from oracle import gsm_oracle as orc
    s = _state(129, 16)
    mu_o, S_o = orc.gsm_update_batched(s["X"], s["G"], s["mu0"], s["S0"])
    eng.last_path()
    mu, S = eng.gsm_update(*(eng.asarray(s[k]) for k in ("X", "G", "mu0", "S0")))
    path = eng.last_path()
    assert rel_err(S.cpu().numpy(), S_o) < TOL and rel_err(mu.cpu().numpy(), mu_o) < TOL
    assert "cov_generic" in path, path


@pytest.mark.parametrize("D,B", [(5, 2), (129, 16), (785, 20), (1001, 32)])
def test_odd_d_one_shot_updates_run_the_tuned_kernels(eng, D, B):
    """gsm_update / bam_update (the drop-ins of gsm_numpy.py:27-55, bam.py:31-114) at ODD D: run as the (D + 1)-dimensional
    problem with an inert last coordinate (gsm-vi_amd/_oddpad.py) -- oracle parity and no guarded kernel."""
    import gsmvi_amd
    from oracle import gsm_oracle as orc
    from oracle import bam_oracle as borc
    s = _state(D, B)
    mu_o, S_o = orc.gsm_update_batched(s["X"], s["G"], s["mu0"], s["S0"])
    eng.last_path()
    mu, S = gsmvi_amd.gsm_update(s["X"], s["G"], s["mu0"], s["S0"])
    path = eng.last_path()
    assert mu.shape == (D,) and S.shape == (D, D)
    assert rel_err(mu, mu_o) < TOL and rel_err(S, S_o) < TOL and np.array_equal(S, S.T)
    assert not _generic(path) and "cov_sym" in path, path
    mu_bo, S_bo = borc.bam_lowrank_update_exact(s["X"], s["G"], s["mu0"], s["S0"], 2.0)
    eng.last_path()
    mu_b, S_b = gsmvi_amd.bam_update(s["X"], s["G"], s["mu0"], s["S0"], 2.0)
    path = eng.last_path()
    assert rel_err(mu_b, mu_bo) < 1e-8 and rel_err(S_b, 0.5 * (S_bo + S_bo.T)) < 1e-8
    assert not _generic(path) and "lowrank_fast" in path, path


@pytest.mark.parametrize("D,B", [(5, 2), (33, 4), (129, 16)])
def test_odd_d_fits_run_padded_and_converge(eng, D, B):
    """GSM.fit / BaM.fit at odd D (the reference's own example is D = 5, examples/example_gsm_numpy.py:38): the fit runs on the
    tuned kernels as the (D + 1)-dimensional problem; the inert coordinate stays EXACTLY inert (mean 0, unit corner, zero
    border), the user's score and monitor see the original D, and the fits converge to the Gaussian target."""
    import torch
    import gsmvi_amd
    from oracle import gsm_oracle as orc
    m, cov_t, P = orc.make_gaussian_target(D, 5)
    tgt = gsmvi_amd.GaussianTarget(m, precision=P)
    shapes, mon_shapes = [], []

    m_t, P_t = torch.as_tensor(m, device="cuda"), torch.as_tensor(np.asarray(P), device="cuda")

    @gsmvi_amd.device_score
    def lp_g(x):                                   # a user's device score in plain torch: sees (B, D)
        shapes.append(tuple(x.shape))
        return -(x - m_t) @ P_t

    class Mon:
        checkpoint = 50

        def __call__(self, i, params, lp, key, nevals=1):
            mon_shapes.append((params[0].shape, params[1].shape))

    niter = 3000 if D <= 33 else 600
    eng.last_path()
    gsm = gsmvi_amd.GSM(D, tgt.lp, lp_g)
    mean, cov = gsm.fit(3, niter=niter, batch_size=B, verbose=False, monitor=Mon())
    path = eng.last_path()
    assert gsm.padded_dim == D + 1 and not _generic(path), path
    assert set(shapes) == {(B, D)} and set(mon_shapes) == {((D,), (D, D))}
    assert mean.shape == (D,) and cov.shape == (D, D)
    if D <= 33:
        assert rel_err(mean, m) < 1e-6 and rel_err(cov, cov_t) < 1e-6
    # the inert coordinate, looked at directly: the inner fit of the padded problem
    inner = gsmvi_amd.GSM(D + 1, None, tgt.lp_g.padded(D + 1))
    mp, cp = inner.fit(3, mean=torch.zeros(D + 1, dtype=torch.float64, device="cuda"),
                       cov=torch.eye(D + 1, dtype=torch.float64, device="cuda"), niter=60, batch_size=B, verbose=False,
                       as_torch=True, _zero_cols_from=D)
    assert float(mp[D]) == 0.0 and float(cp[D, D]) == 1.0 and bool((cp[D, :D] == 0).all()) and bool((cp[:D, D] == 0).all())
    # the same with a HOST score (numpy in, numpy out: gsm_numpy.py:117) and the built-in target (padded precision matrix)
    Ph = np.asarray(P)
    g2 = gsmvi_amd.GSM(D, None, lambda x: -(x - m) @ Ph)
    mean_h, cov_h = g2.fit(3, niter=200, batch_size=B, verbose=False)
    g3 = gsmvi_amd.GSM(D, None, tgt.lp_g)
    mean_b, cov_b = g3.fit(3, niter=200, batch_size=B, verbose=False)
    assert rel_err(mean_h, mean_b) < 1e-9 and rel_err(cov_h, cov_b) < 1e-9           # same draws, same scores
    # BaM, both forms
    for method in ("dense", "factor"):
        if method == "factor" and 2 * B > D:
            continue
        bam = gsmvi_amd.BaM(D, None, tgt.lp_g)
        eng.last_path()
        mb, cb = bam.fit(3, gsmvi_amd.Regularizers().custom(lambda c: 100.0 / c), niter=150, batch_size=B, verbose=False,
                         method=method, jitter=0.0)
        path = eng.last_path()
        assert bam.padded_dim == D + 1 and bam.method_used == method and not _generic(path), (method, path)
        assert rel_err(mb, m) < 1e-3 and rel_err(cb, cov_t) < 1e-2, method


@pytest.mark.parametrize("method", ["dense", "factor"])
def test_odd_d_host_stream_is_the_literal_problems_stream(eng, method):
    """Round-5 advice (_oddpad.py): with rng="numpy" the padded (D + 1)-dimensional fit of an odd-D problem draws its host normals
    at the LITERAL width, so the first samples of key k are RandomState(k).standard_normal((B, D)) like the literal-D problem's
    (mean 0, cov I: x = z) -- not a (B, D + 1) draw that shifts the stream.  The device score also gets a contiguous tensor."""
    import gsmvi_amd
    from gsmvi_amd.targets import device_score
    D, B, key = 7, 2, 11
    seen = []

    @device_score
    def lp_g(x):
        assert x.is_contiguous() and x.shape == (B, D)
        seen.append(x.clone())
        return -2.0 * (x - 0.5)

    gsmvi_amd.GSM(D, None, lp_g).fit(key, niter=2, batch_size=B, verbose=False, rng="numpy", method=method)
    rs = np.random.RandomState(key)
    assert rel_err(seen[0].cpu().numpy(), rs.standard_normal((B, D))) < 1e-15
