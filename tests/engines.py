"""Test-only engine with the HipEngine interface backed by the numpy oracle.  It lets the CPU suite
exercise the fit drivers' HOST logic (RNG stream, monitor cadence, revert, retries, sharding).  It is
never importable from the product package."""
import numpy as np

from oracle import gsm_oracle as orc
from oracle import bam_oracle as borc


class Flag:
    def __init__(self):
        self.v = 0


class OracleEngine:
    name = "oracle(test-only)"
    device = "cpu"

    def asarray(self, x):
        return np.array(x, dtype=np.float64, copy=True) if not isinstance(x, np.ndarray) else x.astype(np.float64)

    def clone(self, x):
        return np.array(x, dtype=np.float64, copy=True)

    def to_numpy(self, t):
        return np.asarray(t)

    def host_score(self, lp_g, X, out=None):
        g = np.asarray(lp_g(np.array(X, copy=True)), dtype=np.float64)
        if out is not None:
            out[...] = g
            return out
        return g

    def empty(self, *shape):
        return np.empty(shape)

    def zeros(self, *shape):
        return np.zeros(shape)

    def eye(self, D):
        return np.eye(D)

    def new_flag(self):
        return Flag()

    def read_flag(self, f):
        return f.v

    def flag_tensor(self, f):                    # what dist.root_potrf broadcasts for a flag
        import torch
        return torch.tensor([f.v], dtype=torch.int32)

    def flag_assign(self, f, t):
        f.v = int(t[0])
        return f

    def normal_from_host(self, z):
        return np.asarray(z, dtype=np.float64)

    def normal(self, B, D, seed, call=0, out=None, raw=None):
        z = orc.philox_randn(seed, call, B * D).reshape(B, D)
        if out is not None:
            out[...] = z
            return out
        return z

    def normal_batch(self, ncalls, B, D, seed, call0=0, out=None, call_in=None, call_out=None):
        z = np.stack([orc.philox_randn(seed, call0 + c, B * D).reshape(B, D) for c in range(ncalls)])
        if out is not None:
            out[...] = z
            return out
        return z

    def gsm_update(self, X, G, mu0, S0, out=None, general=False):
        mu, S = orc.gsm_update_faithful(X, G, mu0, S0) if general else orc.gsm_update_batched(X, G, mu0, S0)
        if out is not None:
            out[0][...] = mu
            out[1][...] = S
            return out
        return mu, S

    def record_len(self, D):
        return 3 * D + (D & 1)

    def gsm_local_stage(self, X, G, mu0, S0, out=None):
        t = orc.gsm_per_sample_terms(X, G, mu0, S0)
        D = mu0.shape[0]
        rec = np.zeros((X.shape[0], self.record_len(D)))
        rec[:, :D], rec[:, D:2 * D], rec[:, 2 * D:3 * D] = t["dvec"], t["evec"], t["dmu"]
        if out is not None:
            out[...] = rec
            return out
        return rec

    def gsm_apply(self, rec, mu0, S0, out=None):
        D = mu0.shape[0]
        B = rec.shape[0]
        d, e, dmu = rec[:, :D], rec[:, D:2 * D], rec[:, 2 * D:3 * D]
        mu = mu0 + dmu.mean(axis=0)
        S = S0 + (d.T @ d - e.T @ e) / B
        if out is not None:
            out[0][...] = mu
            out[1][...] = S
            return out
        return mu, S

    def gsm_rows_stage(self, G, S0_rows, out=None):
        SGc = G @ S0_rows.T
        if out is not None:
            out[...] = SGc
            return out
        return SGc

    def gsm_records(self, X, G, mu0, SG, out=None):
        D = mu0.shape[0]
        d = mu0[None, :] - X
        gSg = np.einsum("bi,bi->b", G, SG)
        mv = np.einsum("bi,bi->b", d, G)
        rho = 0.5 * np.sqrt(1.0 + 4.0 * (gSg + mv * mv)) - 0.5
        den = 1.0 + rho + mv
        dmu = ((SG - d) - d * ((gSg - mv) / den)[:, None]) / (1.0 + rho)[:, None]
        rec = np.zeros((X.shape[0], self.record_len(D)))
        rec[:, :D], rec[:, D:2 * D], rec[:, 2 * D:3 * D] = d, d + dmu, dmu
        if out is not None:
            out[...] = rec
            return out
        return rec

    def gsm_apply_rows(self, rec, mu0, S0_rows, row0, out=None):
        D = mu0.shape[0]
        B = rec.shape[0]
        nr = S0_rows.shape[0]
        d, e, dmu = rec[:, :D], rec[:, D:2 * D], rec[:, 2 * D:3 * D]
        mu = mu0 + dmu.mean(axis=0)
        S = S0_rows + (d[:, row0:row0 + nr].T @ d - e[:, row0:row0 + nr].T @ e) / B
        if out is not None:
            out[0][...] = mu
            out[1][...] = S
            return out
        return mu, S

    def gsm_factor_update(self, Z, X, G, mu0, F0, out=None, flag=None, n_reverts=None):
        flag = Flag() if flag is None else flag
        mu, Fo, ok = orc.gsm_factor_update(Z, G, mu0, F0.T)      # oracle convention: Sigma = F F^T
        if ok and np.isfinite(Fo).all():
            Fn, flag.v = Fo.T, 0
        else:
            mu, Fn, flag.v = mu0.copy(), F0.copy(), 1
            if n_reverts is not None:
                n_reverts.v += 1
        if out is not None:
            out[0][...] = mu
            out[1][...] = Fn
            return out[0], out[1], flag
        return mu, Fn, flag

    def gsm_factor_local_stage(self, Z_l, X_l, G_l, mu0, F0, out=None):
        D = mu0.shape[0]
        t = orc.gsm_factor_terms(Z_l, G_l, F0.T)                 # oracle convention: Sigma = F F^T
        rec = np.zeros((Z_l.shape[0], self.record_len(D)))
        rec[:, :D], rec[:, D:2 * D], rec[:, 2 * D:3 * D] = X_l - mu0[None, :], t["U"], t["U"] @ F0
        if out is not None:
            out[...] = rec
            return out
        return rec

    def gsm_factor_apply(self, Z, rec, mu0, F0, out=None, flag=None, n_reverts=None):
        flag = Flag() if flag is None else flag
        B, D = Z.shape
        U = rec[:, D:2 * D]
        Y = U - Z
        M = np.eye(D) + (Z.T @ Z - Y.T @ Y) / B
        mu = mu0 + rec[:, 2 * D:3 * D].mean(axis=0)
        try:
            Cm = np.linalg.cholesky(0.5 * (M + M.T))
            ok = bool(np.isfinite(Cm).all())
        except np.linalg.LinAlgError:
            ok = False
        if ok:
            Fn, flag.v = Cm.T @ F0, 0                            # (C^T F0)^T (C^T F0) = F0^T M F0
        else:
            mu, Fn, flag.v = mu0.copy(), F0.copy(), 1
            if n_reverts is not None:
                n_reverts.v += 1
        if out is not None:
            out[0][...] = mu
            out[1][...] = Fn
            return out[0], out[1], flag
        return mu, Fn, flag

    def sample_cols(self, Z, mu_cols, Fcols, out=None):
        X = mu_cols[None, :] + Z @ Fcols
        if out is not None:
            out[...] = X
            return out
        return X

    def gsm_factor_w_partial(self, G, col0, Fcols, out=None):
        Wp = G[:, col0:col0 + Fcols.shape[1]] @ Fcols.T
        if out is not None:
            out[...] = Wp
            return out
        return Wp

    def gsm_factor_apply_cols(self, Z, W, X, mu0, F0cols, col0, out=None, flag=None, n_reverts=None):
        """SURVEY A.2 from the whitened quantities alone: u from (z, w), M = I + (Z^T Z - Y^T Y) / B = C C^T, F' = C^T F, so the
        owned columns of F' are C^T F[:, C] and mu'[C] = mu0[C] + mean_b u_b F[:, C]."""
        flag = Flag() if flag is None else flag
        B, D = Z.shape
        nc = F0cols.shape[1]
        ww = np.einsum("bi,bi->b", W, W)
        zw = np.einsum("bi,bi->b", Z, W)
        rho = 0.5 * np.sqrt(1.0 + 4.0 * (ww + zw * zw)) - 0.5
        den = 1.0 + rho - zw
        U = ((W + Z) + Z * ((ww + zw) / den)[:, None]) / (1.0 + rho)[:, None]
        Y = U - Z
        M = np.eye(D) + (Z.T @ Z - Y.T @ Y) / B
        mu = np.array(mu0, copy=True)
        try:
            Cm = np.linalg.cholesky(0.5 * (M + M.T))
            ok = bool(np.isfinite(Cm).all())
        except np.linalg.LinAlgError:
            ok = False
        if ok:
            Fn, flag.v = Cm.T @ F0cols, 0
            mu[col0:col0 + nc] = mu0[col0:col0 + nc] + (U @ F0cols).mean(axis=0)
        else:
            Fn, flag.v = F0cols.copy(), 1
            if n_reverts is not None:
                n_reverts.v += 1
        if out is not None:
            out[0][...] = mu
            out[1][...] = Fn
            return out[0], out[1], flag
        return mu, Fn, flag

    def gram(self, F, out=None, shift=0.0, shift_dev=None):
        Cm = F.T @ F + (shift + (float(shift_dev[0]) if shift_dev is not None else 0.0)) * np.eye(F.shape[0])
        if out is not None:
            out[...] = Cm
            return out
        return Cm

    def owed_shift(self, jitter, pend, n_rev, mark, advance=True):
        s = np.array([(pend - (n_rev.v - mark.v)) * jitter])
        if advance:
            mark.v = n_rev.v
        return s

    def whiten_rows(self, X, mu, R):
        r = X - (mu[None, :] if mu is not None else 0.0)
        Z = np.linalg.solve(R.T, r.T).T                 # z R = r
        return Z, np.array([np.sum(np.log(np.diag(R)))])

    def gaussian_score(self, X, m, P, out=None):
        return orc.gaussian_score(X, m, P)

    def potrf(self, S, out=None, flag=None):
        flag = Flag() if flag is None else flag
        R = np.zeros_like(S) if out is None else out
        if orc.cov_is_good(S):
            R[...] = np.linalg.cholesky(S).T
            flag.v = 0
        else:
            flag.v = 1
        return R, flag

    def sample(self, Z, mu, R, out=None):
        X = mu[None, :] + Z @ R
        if out is not None:
            out[...] = X
            return out
        return X

    def commit(self, flag, mu_new, S_new, mu, S, n_reverts=None):
        if flag.v == 0:
            mu[...] = mu_new
            S[...] = S_new
        elif n_reverts is not None:
            n_reverts.v += 1

    def bam_factor_update(self, Z, X, G, mu0, F0, reg, out=None, flag=None, n_reverts=None):
        """Factor-form BaM on the host: the dense restatement on S0 = F0^T F0 and ANY factor of its result (here the
        upper Cholesky factor; the device kernel returns a different, equally valid F with the same F^T F)."""
        flag = Flag() if flag is None else flag
        mu, S = borc.bam_lowrank_update_exact(X, G, mu0, F0.T @ F0, reg)
        S = 0.5 * (S + S.T)
        try:
            Fn, flag.v = np.linalg.cholesky(S).T, 0
            if not (np.isfinite(Fn).all() and np.isfinite(mu).all()):
                raise np.linalg.LinAlgError
        except np.linalg.LinAlgError:
            mu, Fn, flag.v = mu0.copy(), F0.copy(), 1
            if n_reverts is not None:
                n_reverts.v += 1
        if out is not None:
            out[0][...] = mu
            out[1][...] = Fn
            return out[0], out[1], flag
        return mu, Fn, flag

    def bam_update(self, X, G, mu0, S0, reg, jitter=0.0, out=None, flag=None):
        mu, S = borc.bam_lowrank_update_exact(X, G, mu0, S0, reg)
        S = 0.5 * (S + S.T) + jitter * np.eye(S.shape[0])
        flag = Flag() if flag is None else flag
        if out is not None:
            out[0][...] = mu
            out[1][...] = S
            return out[0], out[1], flag
        return mu, S, flag
