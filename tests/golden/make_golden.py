#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [/root/reference]

The reference's Python never travels to the GPU box; only the .npz vectors written here do.
Fixtures are data: inputs and the reference's outputs.  Sections follow SURVEY.md 8(c):

  G1  gsm_numpy.gsm_update / _gsm_update_single on seeded inputs           -> g1_update.npz
  G2  teacher-forced trajectories of GSM.fit(key=99, niter=500), D=5, 10   -> g2_traj_D{5,10}.npz
  G3  converged endpoint of G2 and the true target (inside g2_*.npz)
  G4  crafted update whose covariance fails the Cholesky test (revert)      -> g4_revert.npz
  G5  monitor cadence (i, nevals) for checkpoint=3, B=2, niter=10            -> g5_monitor.npz
  G6  legacy sampler np.random.seed(s); multivariate_normal(mean, cov, B)    -> g6_sampler.npz
  R1  BaM vectors from THIS REPO's scipy restatement (bam.py needs jax, absent) -> r1_bam.npz
      labelled "restatement-derived, not reference-derived".
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import gsmvi.gsm_numpy as ref  # noqa: E402  (the reference)

from oracle import gsm_oracle as orc  # noqa: E402  (only for seeded input recipes)
from oracle import bam_oracle as borc  # noqa: E402


def g1():
    out = {}
    cases = [(5, 2, s) for s in (0, 1, 2)] + [(10, 2, s) for s in (0, 1, 2)] + \
            [(64, 8, s) for s in (0, 1, 2)] + [(256, 8, 0)] + [(7, 1, 0), (33, 5, 1), (16, 16, 2)]
    names = []
    for D, B, seed in cases:
        st = orc.make_update_state(D, B, seed)
        mu, S = ref.gsm_update(st["samples"], st["vs"], st["mu0"], st["S0"])
        tag = f"D{D}_B{B}_s{seed}"
        names.append(tag)
        out[f"{tag}/samples"] = st["samples"]
        out[f"{tag}/vs"] = st["vs"]
        out[f"{tag}/mu0"] = st["mu0"]
        out[f"{tag}/S0"] = st["S0"]
        out[f"{tag}/mu"] = mu
        out[f"{tag}/S"] = S
        if D <= 64:
            dmu = np.zeros((B, D))
            rho = np.zeros(B)
            for b in range(B):
                dmu[b], _ = ref._gsm_update_single(st["samples"][b], st["vs"][b], st["mu0"], st["S0"])
                v, x = st["vs"][b], st["samples"][b]
                S0v = st["S0"] @ v
                rho[b] = 0.5 * np.sqrt(1 + 4 * (v @ S0v + ((st["mu0"] - x) @ v) ** 2)) - 0.5
            out[f"{tag}/dmu_b"] = dmu
            out[f"{tag}/rho_b"] = rho
    out["cases"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "g1_update.npz"), **out)
    print("g1:", len(names), "cases")


class _Rec:
    """monitor with checkpoint=1 that records (i, mean, cov, nevals) (gsm_numpy.py:110-113)."""
    checkpoint = 1

    def __init__(self):
        self.log = []

    def __call__(self, i, mc, lp, key, nevals=0):
        self.log.append((i, mc[0].copy(), mc[1].copy(), nevals))


def g2():
    for D in (5, 10):
        m, cov_t, P = orc.make_gaussian_target(D, seed=7 + D)
        samples_log, vs_log = [], []

        def lp(x):
            return orc.gaussian_logp(x, m, P)

        def lp_g(x):
            g = orc.gaussian_score(x, m, P)
            samples_log.append(x.copy())
            vs_log.append(g.copy())
            return g

        rec = _Rec()
        gsm = ref.GSM(D=D, lp=lp, lp_g=lp_g)
        mean_fit, cov_fit = gsm.fit(99, niter=500, batch_size=2, verbose=False, monitor=rec)
        # rec.log[i] is the state BEFORE iteration i; final call (i=500 again) is the end state.
        means = np.stack([e[1] for e in rec.log])
        covs = np.stack([e[2] for e in rec.log])
        np.savez_compressed(
            os.path.join(HERE, f"g2_traj_D{D}.npz"),
            target_m=m, target_cov=cov_t, target_P=P, key=np.int64(99), niter=np.int64(500),
            samples=np.stack(samples_log), vs=np.stack(vs_log),
            means=means, covs=covs, mean_fit=mean_fit, cov_fit=cov_fit)
        print(f"g2 D={D}: |mean_fit-m|={abs(mean_fit - m).max():.2e} |cov_fit-cov|={abs(cov_fit - cov_t).max():.2e}",
              "states", means.shape)


def g4():
    """Floating-point failure of the covariance update (exact arithmetic always gives a PD matrix):
    an S0 with many eigenvalues at 1e-14 and O(1e3) displacements, so that rounding noise (~1e-10)
    swamps the small eigenvalues.  Only cases where the failure is DECISIVE are kept: the reference's
    S and the batched restatement's S both have min eigenvalue < -1e-12 * max eigenvalue, so the
    verdict does not depend on summation order.  The fit must keep (mean, cov)."""
    rs = np.random.RandomState(4)
    D, B = 12, 2
    A = rs.standard_normal((D, D))
    _, Q = np.linalg.eigh(A @ A.T)
    w = np.concatenate([np.full(8, 1e-14), np.logspace(-2, 2, D - 8)])
    S0 = (Q * w) @ Q.T
    S0 = 0.5 * (S0 + S0.T)
    assert ref.GSM(D, None, None)._check_goodness(S0)
    mu0 = rs.standard_normal(D)
    found = None
    for trial in range(5000):
        X = mu0 + rs.standard_normal((B, D)) * 1e3
        Gs = rs.standard_normal((B, D)) * 10.0 ** rs.uniform(2, 9)
        mu, S = ref.gsm_update(X, Gs, mu0, S0)
        _, S2 = orc.gsm_update_batched(X, Gs, mu0, S0)
        good = ref.GSM(D, None, None)._check_goodness(S)
        e1 = np.linalg.eigvalsh(0.5 * (S + S.T))
        e2 = np.linalg.eigvalsh(0.5 * (S2 + S2.T))
        if (not good) and e1.min() < -1e-12 * e1.max() and e2.min() < -1e-12 * e2.max() \
                and not orc.cov_is_good(S2):
            found = (X, Gs, mu, S)
            break
    assert found is not None, "could not craft a decisively failing update"
    X, Gs, mu, S = found
    np.savez_compressed(os.path.join(HERE, "g4_revert.npz"), samples=X, vs=Gs, mu0=mu0, S0=S0,
                        mu=mu, S=S, is_good=np.bool_(False),
                        nan_is_good=np.bool_(ref.GSM(D, None, None)._check_goodness(np.full((D, D), np.nan))))
    print("g4: decisively failing update found at trial", trial, "min eig ratio", e1.min() / e1.max())


def g5():
    class Mon:
        checkpoint = 3

        def __init__(self):
            self.calls = []

        def __call__(self, i, mc, lp, key, nevals=0):
            self.calls.append((i, nevals))

    D = 4
    m, cov_t, P = orc.make_gaussian_target(D, seed=3)
    mon = Mon()
    n_lpg = [0]

    def lp_g(x):
        n_lpg[0] += 1
        return orc.gaussian_score(x, m, P)

    ref.GSM(D, None, lp_g).fit(5, niter=10, batch_size=2, verbose=False, monitor=mon)
    np.savez_compressed(os.path.join(HERE, "g5_monitor.npz"), calls=np.array(mon.calls),
                        n_lp_g=np.int64(n_lpg[0]), target_m=m, target_P=P)
    print("g5:", mon.calls, "lp_g calls", n_lpg[0])


def g6():
    out = {}
    for D, B, seed in [(5, 2, 99), (10, 2, 99), (16, 4, 1)]:
        st = orc.make_update_state(D, B, seed)
        np.random.seed(seed)
        x = np.random.multivariate_normal(mean=st["mu0"], cov=st["S0"], size=B)
        x2 = np.random.multivariate_normal(mean=st["mu0"], cov=st["S0"], size=B)
        tag = f"D{D}_B{B}_s{seed}"
        out[f"{tag}/mean"] = st["mu0"]
        out[f"{tag}/cov"] = st["S0"]
        out[f"{tag}/x"] = x
        out[f"{tag}/x2"] = x2
    np.savez_compressed(os.path.join(HERE, "g6_sampler.npz"), **out)
    print("g6 done")


def r1():
    """restatement-derived, not reference-derived (bam.py cannot be imported: no jax)."""
    out = {"label": np.array("restatement-derived, not reference-derived")}
    names = []
    for D, B, seed, reg in [(5, 2, 0, 1.0), (10, 2, 1, 100.0), (16, 4, 2, 0.5), (64, 8, 0, 10.0),
                            (12, 12, 1, 2.0), (8, 16, 3, 1.0)]:
        st = orc.make_update_state(D, B, seed)
        mu_f, S_f = borc.bam_update_full(st["samples"], st["vs"], st["mu0"], st["S0"], reg)
        mu_l, S_l = borc.bam_lowrank_update_exact(st["samples"], st["vs"], st["mu0"], st["S0"], reg)
        tag = f"D{D}_B{B}_s{seed}"
        names.append(tag)
        for k in ("samples", "vs", "mu0", "S0"):
            out[f"{tag}/{k}"] = st[k]
        out[f"{tag}/reg"] = np.float64(reg)
        out[f"{tag}/mu_full"], out[f"{tag}/S_full"] = mu_f, S_f
        out[f"{tag}/mu_lowrank"], out[f"{tag}/S_lowrank"] = mu_l, S_l
    out["cases"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "r1_bam.npz"), **out)
    print("r1:", len(names), "cases")


if __name__ == "__main__":
    g1()
    g2()
    g4()
    g5()
    g6()
    r1()
    # the import above may have dropped a __pycache__ in the read-only tree; remove it if we made one
    pc = os.path.join(REF, "gsmvi", "__pycache__")
    if os.path.isdir(pc):
        import shutil
        shutil.rmtree(pc, ignore_errors=True)
