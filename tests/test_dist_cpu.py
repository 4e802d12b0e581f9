"""The N>1 (batch-sharded) path on CPU: world_size=2 gloo processes, oracle-backed test engine.
Checks the sharding/exchange logic of gsm-vi_amd/dist.py; the HIP kernels behind the same two
stage calls are checked on the GPU in test_gpu_gsm_update.py::test_two_stage_equals_fused."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import gsm_oracle as orc
        from engines import OracleEngine
        from gsmvi_amd.dist import sharded_gsm_update, shard_bounds
        eng = OracleEngine()
        st = orc.make_update_state(24, 8, 3)
        lo, hi = shard_bounds(8, world, rank)
        mu, S = sharded_gsm_update(eng, st["samples"][lo:hi], st["vs"][lo:hi], st["mu0"], st["S0"])
        mu_o, S_o = orc.gsm_update_faithful(st["samples"], st["vs"], st["mu0"], st["S0"])
        err = max(np.abs(mu - mu_o).max(), np.abs(S - S_o).max())
        # replicas must be bit-identical
        t = torch.from_numpy(np.concatenate([mu, S.ravel()]))
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        same = all(torch.equal(gathered[0], x) for x in gathered)
        # BaM: all-gather of samples and scores, identical update on every replica
        from oracle import bam_oracle as borc
        from gsmvi_amd.dist import sharded_bam_update
        mu_b, S_b, _ = sharded_bam_update(eng, st["samples"][lo:hi], st["vs"][lo:hi], st["mu0"], st["S0"], 2.0)
        mu_bo, S_bo = borc.bam_lowrank_update_exact(st["samples"], st["vs"], st["mu0"], st["S0"], 2.0)
        err = max(err, np.abs(mu_b - mu_bo).max(), np.abs(S_b - 0.5 * (S_bo + S_bo.T)).max())
        # sharded FIT: same key on every rank, lp_g sees only the local rows, replicas end identical
        from gsmvi_amd.gsm import GSM
        m, cov_t, P = orc.make_gaussian_target(6, 4)
        seen = []

        def lp_g(x):
            seen.append(x.shape[0])
            return orc.gaussian_score(x, m, P)

        mean_s, cov_s = GSM(6, None, lp_g, engine=OracleEngine()).fit(7, niter=30, batch_size=4, verbose=False,
                                                                       shard=True)
        mean_1, cov_1 = GSM(6, None, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine()).fit(
            7, niter=30, batch_size=4, verbose=False)
        assert set(seen) == {4 // world}
        err = max(err, np.abs(mean_s - mean_1).max(), np.abs(cov_s - cov_1).max())
        # batch-sharded FACTOR path (BASELINE config 5 on several GPUs): update and fit
        from gsmvi_amd.dist import sharded_gsm_factor_update
        stf = orc.make_update_state(12, 4, 9)
        F0 = stf["L"].T.copy()                                    # engine convention: Sigma = F0^T F0
        flo, fhi = shard_bounds(4, world, rank)
        mu_f, F_f, fl = sharded_gsm_factor_update(eng, stf["Z"], stf["samples"][flo:fhi], stf["vs"][flo:fhi],
                                                   stf["mu0"], F0, flo)
        mu_o, S_o = orc.gsm_update_faithful(stf["samples"], stf["vs"], stf["mu0"], stf["S0"])
        assert fl.v == 0
        err = max(err, np.abs(mu_f - mu_o).max(), np.abs(F_f.T @ F_f - S_o).max())
        m2, cov_t2, P2 = orc.make_gaussian_target(10, 4)          # factor form needs 2B <= D
        seen2 = []

        def lp_g2(x):
            seen2.append(x.shape[0])
            return orc.gaussian_score(x, m2, P2)

        mean_fs, cov_fs = GSM(10, None, lp_g2, engine=OracleEngine()).fit(7, niter=30, batch_size=4, verbose=False,
                                                                           shard=True, method="factor")
        mean_f1, cov_f1 = GSM(10, None, lambda x: orc.gaussian_score(x, m2, P2), engine=OracleEngine()).fit(
            7, niter=30, batch_size=4, verbose=False, method="factor")
        assert set(seen2) == {4 // world}
        err = max(err, np.abs(mean_fs - mean_f1).max(), np.abs(cov_fs - cov_f1).max())
        # sharded BaM FIT (BASELINE config 4 in miniature): same key everywhere, local rows scored, replicas identical
        from gsmvi_amd.bam import BaM, Regularizers
        seen3 = []

        def lp_g3(x):
            seen3.append(x.shape[0])
            return orc.gaussian_score(x, m, P)

        mean_bs, cov_bs = BaM(6, None, lp_g3, engine=OracleEngine()).fit(7, Regularizers().constant(5.0), niter=20,
                                                                         batch_size=4, verbose=False, shard=True)
        mean_b1, cov_b1 = BaM(6, None, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine()).fit(
            7, Regularizers().constant(5.0), niter=20, batch_size=4, verbose=False)
        assert set(seen3) == {4 // world}
        err = max(err, np.abs(mean_bs - mean_b1).max(), np.abs(cov_bs - cov_b1).max())
        # sharded FACTOR-FORM BaM (round 4; BASELINE config 4 is "B sharded per GPU"): update and fit, replicas identical
        from gsmvi_amd.dist import sharded_bam_factor_update
        mu_bf, F_bf, flb = sharded_bam_factor_update(eng, stf["Z"], stf["samples"][flo:fhi], stf["vs"][flo:fhi],
                                                      stf["mu0"], F0, 2.0)
        mu_bfo, S_bfo = borc.bam_lowrank_update_exact(stf["samples"], stf["vs"], stf["mu0"], stf["S0"], 2.0)
        assert flb.v == 0
        err = max(err, np.abs(mu_bf - mu_bfo).max(), np.abs(F_bf.T @ F_bf - 0.5 * (S_bfo + S_bfo.T)).max())
        seen4 = []

        def lp_g4(x):
            seen4.append(x.shape[0])
            return orc.gaussian_score(x, m2, P2)

        mean_bfs, cov_bfs = BaM(10, None, lp_g4, engine=OracleEngine()).fit(7, Regularizers().constant(5.0), niter=20,
                                                                            batch_size=4, verbose=False, shard=True,
                                                                            method="factor")
        mean_bf1, cov_bf1 = BaM(10, None, lambda x: orc.gaussian_score(x, m2, P2), engine=OracleEngine()).fit(
            7, Regularizers().constant(5.0), niter=20, batch_size=4, verbose=False, method="factor")
        assert set(seen4) == {4 // world}
        err = max(err, np.abs(mean_bfs - mean_bf1).max(), np.abs(cov_bfs - cov_bf1).max())
        t2 = torch.from_numpy(np.concatenate([mean_bfs, cov_bfs.ravel()]))
        g2 = [torch.empty_like(t2) for _ in range(world)]
        dist.all_gather(g2, t2)
        same = same and all(torch.equal(g2[0], x) for x in g2)
        # opt-in root Cholesky + broadcast (dist.root_potrf) in both dense fits: same numbers as the replicated factorisation
        mean_rp, cov_rp = GSM(6, None, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine()).fit(
            7, niter=30, batch_size=4, verbose=False, shard=True, method="dense", root_potrf=True)
        mean_rd, cov_rd = GSM(6, None, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine()).fit(
            7, niter=30, batch_size=4, verbose=False, shard=True, method="dense")
        err = max(err, np.abs(mean_rp - mean_rd).max(), np.abs(cov_rp - cov_rd).max())
        mean_brp, cov_brp = BaM(6, None, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine()).fit(
            7, Regularizers().constant(5.0), niter=20, batch_size=4, verbose=False, shard=True, root_potrf=True)
        err = max(err, np.abs(mean_brp - mean_bs).max(), np.abs(cov_brp - cov_bs).max())
        # row-block sharded covariance (ragged: D = 25 rows over 2 ranks = 13 + 12)
        from gsmvi_amd.dist import row_sharded_gsm_update, row_bounds
        st = orc.make_update_state(25, 6, 5)
        rlo, rhi = row_bounds(25, world, rank)
        mu_r, S_r = row_sharded_gsm_update(eng, st["samples"], st["vs"], st["mu0"], st["S0"][rlo:rhi].copy())
        mu_o, S_o = orc.gsm_update_faithful(st["samples"], st["vs"], st["mu0"], st["S0"])
        assert S_r.shape == (rhi - rlo, 25)
        err = max(err, np.abs(mu_r - mu_o).max(), np.abs(S_r - S_o[rlo:rhi]).max())
        q.put((rank, float(err), bool(same)))
    finally:
        dist.destroy_process_group()


def test_sharded_update_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, same in res:
        assert err < 1e-12 and same, (rank, err, same)


def test_row_bounds_cover_and_are_ragged_at_the_end():
    from gsmvi_amd.dist import row_bounds
    for D, P in ((25, 2), (1024, 8), (10, 3), (4096, 8), (7, 7)):
        b = [row_bounds(D, P, r) for r in range(P)]
        assert b[0][0] == 0 and b[-1][1] == D
        assert all(b[i][1] == b[i + 1][0] for i in range(P - 1))
        assert all(hi - lo == -(-D // P) for lo, hi in b[:-1])


def test_row_sharded_update_single_process_equals_oracle():
    """world = 1 (no process group): the three row-block stages with the oracle-backed engine."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import gsm_oracle as orc
    from engines import OracleEngine
    from gsmvi_amd.dist import row_sharded_gsm_update
    st = orc.make_update_state(17, 4, 2)
    mu, S = row_sharded_gsm_update(OracleEngine(), st["samples"], st["vs"], st["mu0"], st["S0"])
    mu_o, S_o = orc.gsm_update_faithful(st["samples"], st["vs"], st["mu0"], st["S0"])
    assert np.abs(mu - mu_o).max() < 1e-12 and np.abs(S - S_o).max() < 1e-12


def test_shard_bounds():
    from gsmvi_amd.dist import shard_bounds
    assert [shard_bounds(32, 8, r) for r in (0, 7)] == [(0, 4), (28, 32)]
    with pytest.raises(AssertionError):
        shard_bounds(10, 4, 0)


def test_world1_no_process_group():
    from oracle import gsm_oracle as orc
    from engines import OracleEngine
    from gsmvi_amd.dist import sharded_gsm_update
    st = orc.make_update_state(12, 4, 1)
    mu, S = sharded_gsm_update(OracleEngine(), st["samples"], st["vs"], st["mu0"], st["S0"])
    mu_o, S_o = orc.gsm_update_batched(st["samples"], st["vs"], st["mu0"], st["S0"])
    assert np.abs(mu - mu_o).max() < 1e-12 and np.abs(S - S_o).max() < 1e-12


def _retry_worker(rank, world, port, q, method="dense"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import gsm_oracle as orc
        from engines import OracleEngine
        from gsmvi_amd.bam import BaM, Regularizers
        D, B = (6, 4) if method == "dense" else (10, 4)      # factor form: 2B <= D
        m, cov_t, P = orc.make_gaussian_target(D, 4)
        calls = [0]

        def lp_g(x):
            calls[0] += 1
            if rank == 1 and calls[0] in (3, 4, 9):          # fails on ONE rank only, twice in a row once
                raise FloatingPointError("synthetic score failure")
            return orc.gaussian_score(x, m, P)

        reg = Regularizers()
        mean, cov = BaM(D, None, lp_g, engine=OracleEngine()).fit(5, reg.linear(10.0), niter=12, batch_size=B,
                                                                   verbose=False, shard=True, retries=3, method=method)
        t = torch.from_numpy(np.concatenate([mean, cov.ravel(), [float(reg.counter), float(calls[0])]]))
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        same = all(torch.equal(gathered[0], x) for x in gathered)
        q.put((rank, same, reg.counter, calls[0], bool(np.isfinite(cov).all())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("method", ["dense", "factor"])
def test_sharded_bam_retries_are_collective(method):
    """A score failure on ONE rank must make EVERY rank retry (fail bit all-reduced before the gather): replicas
    stay identical, the regulariser advances equally, nobody re-enters a collective alone (round-1 advice).  Both the
    dense and (round 4) the factor-form sharded fit."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + (2000 if method == "factor" else 0)
    procs = [ctx.Process(target=_retry_worker, args=(r, world, port, q, method)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, counter, ncalls, finite in res:
        assert same and finite, (rank, same, finite)
        assert counter == 13                # niter + 1 successful updates; failed attempts never reached regf
        assert ncalls == 13 + 3             # three collective retries: every rank redrew and re-scored


def _cols_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    try:
        from oracle import gsm_oracle as orc
        from engines import OracleEngine
        from gsmvi_amd.gsm import GSM
        from gsmvi_amd.dist import col_bounds, col_gather_samples, col_sharded_gsm_factor_update
        eng = OracleEngine()
        D, B = 128, 4
        st = orc.make_update_state(D, B, 5)
        F0 = st["L"].T.copy()                                     # Sigma = F0^T F0, x = mu + z F0
        lo, hi = col_bounds(D, world, rank)
        stats = {}
        Xc = eng.sample_cols(st["Z"], st["mu0"][lo:hi], F0[:, lo:hi])
        X = col_gather_samples(eng, Xc, stats=stats)
        out["x_err"] = float(np.abs(X - st["samples"]).max())
        mu, Fc, fl = col_sharded_gsm_factor_update(eng, st["Z"], X, st["vs"], st["mu0"], F0[:, lo:hi].copy(), stats=stats)
        mu_o, S_o = orc.gsm_update_faithful(st["samples"], st["vs"], st["mu0"], st["S0"])
        # assemble the factor from the blocks of both ranks
        t = torch.from_numpy(np.ascontiguousarray(Fc))
        blocks = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(blocks, t)
        F = np.concatenate([b.numpy() for b in blocks], axis=1)
        mt = torch.from_numpy(np.ascontiguousarray(mu[lo:hi]))
        mparts = [torch.empty_like(mt) for _ in range(world)]
        dist.all_gather(mparts, mt)
        mu_full = np.concatenate([p.numpy() for p in mparts])
        out["upd_err"] = float(max(np.abs(F.T @ F - S_o).max() / np.abs(S_o).max(), np.abs(mu_full - mu_o).max()))
        out["flag"] = fl.v
        out["stats"] = stats
        # the FIT: column-sharded against the replicated factor fit, same key
        m, cov_t, P = orc.make_gaussian_target(D, 4)
        seen = []

        def lp_g(x):
            seen.append(x.shape)
            return orc.gaussian_score(x, m, P)

        g = GSM(D, None, lp_g, engine=OracleEngine())
        mean_c, cov_c = g.fit(7, niter=25, batch_size=B, verbose=False, shard="cols", rng="device")
        mean_1, cov_1 = GSM(D, None, lambda x: orc.gaussian_score(x, m, P), engine=OracleEngine()).fit(
            7, niter=25, batch_size=B, verbose=False, method="factor", rng="device")
        out["fit_err"] = float(max(np.abs(mean_c - mean_1).max(), np.abs(cov_c - cov_1).max() / np.abs(cov_1).max()))
        out["seen"] = sorted(set(seen))
        out["fit_stats"] = g.shard_stats
        out["method"] = g.method_used
        t = torch.from_numpy(np.concatenate([mean_c, cov_c.ravel()]))
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        out["same"] = all(torch.equal(gathered[0], x) for x in gathered)
        out["ok"] = True
    except Exception:                                            # noqa: BLE001
        import traceback
        out["ok"] = False
        out["exc"] = traceback.format_exc()
    q.put((rank, out))
    dist.destroy_process_group()


def test_column_sharded_factor_form_world2_gloo():
    """the factor form sharded by COLUMN blocks of the square factor.  Two gloo ranks, the
    oracle-backed engine: the gathered sample slices are the samples, the one-shot update assembled from the two blocks is the
    reference update (gsm_numpy.py:27-55), the sharded fit walks the replicated factor fit's trajectory, every rank scores all
    B samples, and the exchange is what dist.py says: one all-gather of B D / P doubles and one all-reduce of B D doubles."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_cols_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        o = res[r]
        assert o["ok"], o.get("exc")
        assert o["x_err"] < 1e-12 and o["flag"] == 0 and o["upd_err"] < 1e-10, o
        assert o["stats"] == {"all_gather_bytes_per_rank": 4 * 64 * 8, "collectives": 2, "all_reduce_bytes": 4 * 128 * 8}, o["stats"]
        assert o["fit_err"] < 1e-9 and o["same"] and o["method"] == "factor", o
        assert o["seen"] == [(4, 128)]
        assert o["fit_stats"]["block_bytes"] == 128 * 64 * 8


def test_col_bounds():
    from gsmvi_amd.dist import col_bounds
    assert [col_bounds(1024, 8, r) for r in (0, 7)] == [(0, 128), (896, 1024)]
    assert col_bounds(4096, 8, 3) == (1536, 2048)
    with pytest.raises(AssertionError):
        col_bounds(1000, 8, 0)
