"""Two-launch dense update (csrc/gsmvi_fused.hip: k_panel_seam + k_gsm_cov_fused) against the three-launch path
(tuning knob fused=0, the parity reference of round 1) and the pinned oracle; the in-kernel seam of k_panel_seam is
run under uneven load with every output word checked (MI355X_MICROARCH.md: "test every hand-off under UNEVEN
load")."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

SIZES = [(1024, 32), (1024, 16), (1024, 64), (512, 32), (768, 16), (256, 16), (256, 64), (64, 16), (128, 32)]


def _state(D, B, seed):
    from oracle import gsm_oracle as orc
    return orc, orc.make_update_state(D, B, seed)


@pytest.fixture
def eng():
    import gsmvi_amd
    e = gsmvi_amd.get_engine()
    e.set_tuning("fused", 1)            # opt-in path (default off: measured no faster than three launches)
    yield e
    e.set_tuning("fused", 0)
    e.set_tuning("seam_finish", 1)
    e.set_tuning("fused_flags", 0)
    e.set_tuning("timeline", 0)


def _is_fused(eng, X, G, mu0, S0):
    """True when gsm_update really took the two-launch path: the timeline stamps of the per-sample kernel (slot 1)
    stay untouched."""
    import ctypes as C
    eng.set_tuning("timeline", 1)
    buf = (C.c_ulonglong * (4 * 4096))()
    eng.gsm_update(X, G, mu0, S0)
    eng.lib.gsmvi_debug_read_stamps(eng._ctx, buf, 4 * 4096)
    a = np.array(buf, dtype=np.uint64).reshape(4, 4096)
    eng.set_tuning("timeline", 0)
    return bool(a[0].any() and a[2].any() and not a[1].any())


@pytest.mark.parametrize("D,B", SIZES)
def test_fused_equals_three_launch_path_and_oracle(eng, D, B):
    orc, st = _state(D, B, D + B)
    X, G, mu0, S0 = (eng.asarray(st[k]) for k in ("samples", "vs", "mu0", "S0"))
    eng.set_tuning("fused", 0)
    mu_r, S_r = eng.gsm_update(X, G, mu0, S0)
    eng.set_tuning("fused", 1)
    mu_f, S_f = eng.gsm_update(X, G, mu0, S0)
    assert rel_err(mu_f.cpu().numpy(), mu_r.cpu().numpy()) < 1e-13
    assert rel_err(S_f.cpu().numpy(), S_r.cpu().numpy()) < 1e-13
    assert np.array_equal(S_f.cpu().numpy(), S_f.cpu().numpy().T)
    mu_o, S_o = orc.gsm_update_batched(st["samples"], st["vs"], st["mu0"], st["S0"])
    assert rel_err(mu_f.cpu().numpy(), mu_o) < 1e-11 and rel_err(S_f.cpu().numpy(), S_o) < 1e-11
    for flags in (1,):                                  # mirror tile stored without the LDS transpose
        eng.set_tuning("fused_flags", flags)
        mu_2, S_2 = eng.gsm_update(X, G, mu0, S0)
        eng.set_tuning("fused_flags", 0)
        assert np.array_equal(S_2.cpu().numpy(), S_f.cpu().numpy()) and np.array_equal(mu_2.cpu().numpy(),
                                                                                         mu_f.cpu().numpy())


def test_the_fused_path_is_the_one_that_runs_at_the_headline_config(eng):
    orc, st = _state(1024, 32, 1)
    X, G, mu0, S0 = (eng.asarray(st[k]) for k in ("samples", "vs", "mu0", "S0"))
    assert _is_fused(eng, X, G, mu0, S0)
    eng.set_tuning("fused", 0)
    assert not _is_fused(eng, X, G, mu0, S0)
    eng.set_tuning("fused", 1)
    # not eligible: B = 8, odd leading dimension -> falls back to the three-launch / generic kernels
    orc, st = _state(256, 8, 1)
    X, G, mu0, S0 = (eng.asarray(st[k]) for k in ("samples", "vs", "mu0", "S0"))
    assert not _is_fused(eng, X, G, mu0, S0)


def test_fused_update_is_deterministic_under_uneven_load(eng):
    """The strip's last arriver differs from run to run; the result must not (fixed summation order), and no word of
    a partial piece may ever be read stale.  Load: GEMMs of varying size on a second stream and a back-to-back
    stream of updates on a ring of states (consumer caches warm with the previous update's pieces)."""
    import torch
    D, B = 1024, 32
    cases = []
    for seed in range(4):
        orc, st = _state(D, B, 100 + seed)
        cases.append(tuple(eng.asarray(st[k]) for k in ("samples", "vs", "mu0", "S0")))
    quiet = []
    for X, G, mu0, S0 in cases:
        mu, S = eng.gsm_update(X, G, mu0, S0)
        quiet.append((mu.clone(), S.clone()))
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    A = torch.randn(3072, 3072, device=eng.device)
    outs = [(eng.empty(D), eng.empty(D, D)) for _ in range(8)]
    bad = 0
    for rep in range(60):
        with torch.cuda.stream(side):                   # uneven background load
            n = 256 * (1 + rep % 11)
            (A[:n, :n] @ A[:n, :n]).sum()
        for k in range(8):
            X, G, mu0, S0 = cases[k % 4]
            eng.gsm_update(X, G, mu0, S0, out=outs[k])
        torch.cuda.synchronize()
        for k in range(8):
            mu_q, S_q = quiet[k % 4]
            if not (torch.equal(outs[k][0], mu_q) and torch.equal(outs[k][1], S_q)):
                bad += 1
    assert bad == 0, f"{bad} of 480 updates differ from the quiet run"


def test_fused_update_in_a_graph_and_counters_return_to_zero(eng):
    import torch
    orc, st = _state(1024, 32, 7)
    X, G, mu0, S0 = (eng.asarray(st[k]) for k in ("samples", "vs", "mu0", "S0"))
    out = (eng.empty(1024), eng.empty(1024, 1024))
    eng.gsm_update(X, G, mu0, S0, out=out)
    ref = (out[0].clone(), out[1].clone())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5):
            eng.gsm_update(X, G, mu0, S0, out=out)
    for _ in range(20):
        out[0].zero_(); out[1].zero_()
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])


def test_chained_fused_updates_follow_the_oracle(eng):
    """30 dependent updates (state fed back) at D=256, B=16 on a Gaussian target vs the batched oracle."""
    from oracle import gsm_oracle as orc
    D, B = 256, 16
    m, cov_t, P = orc.make_gaussian_target(D, 2)
    rs = np.random.RandomState(5)
    mu_o, S_o = np.zeros(D), np.eye(D)
    mu_d, S_d = eng.zeros(D), eng.eye(D)
    for it in range(30):
        L = np.linalg.cholesky(S_o)
        X = mu_o + rs.standard_normal((B, D)) @ L.T
        Gs = orc.gaussian_score(X, m, P)
        mu_o, S_o = orc.gsm_update_batched(X, Gs, mu_o, S_o)
        mu_d, S_d = eng.gsm_update(eng.asarray(X), eng.asarray(Gs), mu_d, S_d)
        assert rel_err(mu_d.cpu().numpy(), mu_o) < 1e-9 and rel_err(S_d.cpu().numpy(), S_o) < 1e-9, it


def test_seam_finished_panel_products_equal_product_plus_finish(eng):
    """sample / score / U F / Gram products with the split-K slabs combined inside the product launch (per-strip seam of
    k_panel_fast) against the same products followed by k_panel_finish (knob seam_finish=0): bit-identical, also under
    uneven load and in a tight loop (the hand-off is the one of k_panel_seam)."""
    import torch
    from oracle import gsm_oracle as orc
    for D, B in ((1024, 32), (1024, 16), (512, 64), (256, 8), (1024, 4)):
        st = orc.make_update_state(D, B, D + B)
        X, mu0, S0 = (eng.asarray(st[k]) for k in ("samples", "mu0", "S0"))
        Z = eng.asarray(st["Z"])
        m, P = eng.asarray(st["m"]), eng.asarray(st["P"])
        R, _ = eng.potrf(S0)
        eng.set_tuning("seam_finish", 0)
        Xr, Gr = eng.sample(Z, mu0, R).clone(), eng.gaussian_score(X, m, P).clone()
        eng.set_tuning("seam_finish", 2)
        side = torch.cuda.Stream()
        A = torch.randn(2048, 2048, device=eng.device)
        bad = 0
        for rep in range(25):
            with torch.cuda.stream(side):
                n = 256 * (1 + rep % 8)
                (A[:n, :n] @ A[:n, :n]).sum()
            outs = [(eng.sample(Z, mu0, R), eng.gaussian_score(X, m, P)) for _ in range(4)]
            torch.cuda.synchronize()
            bad += sum(0 if (torch.equal(a, Xr) and torch.equal(b, Gr)) else 1 for a, b in outs)
        assert bad == 0, (D, B, bad)
        assert rel_err(Gr.cpu().numpy(), orc.gaussian_score(st["samples"], st["m"], st["P"])) < 1e-9
    # the factor update (U F and Gram products through the seam) against its finish-kernel form
    st = orc.make_update_state(1024, 32, 5)
    F0 = eng.asarray(st["L"].T.copy())
    args = (eng.asarray(st["Z"]), eng.asarray(st["samples"]), eng.asarray(st["vs"]), eng.asarray(st["mu0"]), F0)
    eng.set_tuning("seam_finish", 0)
    mu_r, F_r, _ = eng.gsm_factor_update(*args)
    eng.set_tuning("seam_finish", 2)
    for _ in range(20):
        mu_s, F_s, fl = eng.gsm_factor_update(*args)
        assert torch.equal(mu_s, mu_r) and torch.equal(F_s, F_r) and eng.read_flag(fl) == 0
