"""Batch-sharded GSM update: one process per GPU, records exchanged with ONE all-gather (RCCL).

The reference has no multi-device path; BASELINE.json's north_star asks for the batch of samples
to be sharded across the GPUs of a node with an exchange "of the per-sample update contributions
before the combined rank-B update is applied".  The contributions are low-rank, so ranks exchange
per-sample RECORDS [d_b | e_b | dmu_b] (3D doubles each: the two factor rows of the sample's
rank-2 covariance increment and its mean increment) -- never D x D matrices -- and every replica
applies the identical combined update (fixed summation order => replicas stay bit-identical).
Per update and rank: (B/P) 3D * 8 bytes sent; at D=1024, B=32, P=8 that is 96 KiB, i.e.
latency-bound on xGMI.
"""
import numpy as np
import torch
import torch.distributed as dist


def _as_torch(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


def _all_gather(t_all, t_loc, group=None):
    """all_gather_into_tensor on the group's backend.  RCCL ("nccl") takes device tensors as they are; a gloo group
    (CPU rendezvous, used by the tests that run two HIP-backed ranks on ONE GPU -- RCCL refuses two ranks per
    device) gets the device tensors staged through the host."""
    if t_all.is_cuda and dist.get_backend(group) == "gloo":
        h_all = torch.empty(t_all.shape, dtype=t_all.dtype)
        dist.all_gather_into_tensor(h_all, t_loc.contiguous().cpu(), group=group)
        t_all.copy_(h_all)
        return
    dist.all_gather_into_tensor(t_all, t_loc.contiguous(), group=group)


def _broadcast(t, src, group=None):
    """broadcast on the group's backend (device tensors staged through the host for a gloo group, as in _all_gather)."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
        return
    dist.broadcast(t, src=src, group=group)


def root_potrf(eng, S, R, flag, group=None, root=0):
    """The Cholesky accept test of a batch-sharded DENSE fit (gsm_numpy.py:121-125,132-146): the replicas hold identical
    covariances, so ONE rank factors (O(D^3)) and broadcasts the factor and its flag instead of every rank repeating the
    factorisation (D^2 doubles over xGMI: 8 MiB at D = 1024 against a ~0.3 ms Cholesky).  Replicas stay bit-identical."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return eng.potrf(S, out=R, flag=flag)
    if rank == root:
        eng.potrf(S, out=R, flag=flag)
    src = dist.get_global_rank(group, root) if group is not None else root
    tR, tf = _as_torch(R), eng.flag_tensor(flag)           # (the engine knows what its flags are: a device int32 tensor here)
    _broadcast(tR, src, group)
    _broadcast(tf, src, group)
    eng.flag_assign(flag, tf)
    return R, flag


def shard_bounds(B, world, rank):
    """Contiguous equal shards; B must be divisible by the world size (one fixed-size all-gather)."""
    assert B % world == 0, f"batch size {B} must be divisible by the number of ranks {world}"
    per = B // world
    return rank * per, (rank + 1) * per


def sharded_gsm_update(eng, X_local, G_local, mu0, S0, group=None, rec_all=None, out=None, force_collective=False):
    """(mu, S) of gsm_update for the union of all ranks' samples (gsmvi/gsm_numpy.py:27-55).

    X_local, G_local: this rank's (B/P, D) samples and scores; mu0, S0 replicated.  ``rec_all``
    (B, 2D+4) may be passed to avoid allocating the gather buffer every call."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rec_local = eng.gsm_local_stage(X_local, G_local, mu0, S0)
    if world == 1 and not (force_collective and dist.is_initialized()):
        rec = rec_local
    else:
        Bl, L = rec_local.shape
        if rec_all is None:
            rec_all = eng.empty(Bl * world, L)
        t_all, t_loc = _as_torch(rec_all), _as_torch(rec_local)
        _all_gather(t_all, t_loc, group)
        rec = rec_all if isinstance(rec_all, torch.Tensor) else t_all.numpy()
    return eng.gsm_apply(rec, mu0, S0, out=out)


def sharded_gsm_factor_update(eng, Z, X_local, G_local, mu0, F0, lo, group=None, rec_all=None, out=None, flag=None,
                              n_reverts=None, force_collective=False):
    """Batch-sharded factor-form update (BASELINE config 5 on several GPUs): (mu, F, flag).

    Z (B, D): the whitened draws of ALL samples, replicated (same key on every rank); X_local, G_local: samples
    and scores of this rank's rows [lo, lo + B/P).  Each rank runs the per-sample stage for its rows (two of
    the three passes over F0 divided by P), records [x - mu0 | v | v F0] (v = w + z, the whitened residual) are all-gathered (3D doubles per
    sample, like the dense path) and every replica applies the identical rank-2B factor update."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    Bl = X_local.shape[0]
    rec_local = eng.gsm_factor_local_stage(Z[lo:lo + Bl], X_local, G_local, mu0, F0)
    if world == 1 and not (force_collective and dist.is_initialized()):
        rec = rec_local
    else:
        if rec_all is None:
            rec_all = eng.empty(Bl * world, rec_local.shape[1])
        _all_gather(_as_torch(rec_all), _as_torch(rec_local), group)
        rec = rec_all if isinstance(rec_all, torch.Tensor) else _as_torch(rec_all).numpy()
    return eng.gsm_factor_apply(Z, rec, mu0, F0, out=out, flag=flag, n_reverts=n_reverts)


def row_bounds(D, world, rank):
    """Rows [lo, hi) of the covariance owned by ``rank``: blocks of ceil(D / world) rows, the last one ragged
    (possibly empty ranks are not supported: D >= world)."""
    per = -(-D // world)
    lo = min(rank * per, D)
    return lo, min(lo + per, D)


def row_sharded_gsm_update(eng, X, G, mu0, S0_rows, group=None, out=None):
    """gsm_update with the covariance sharded by ROW BLOCKS.

    Every rank holds rows ``row_bounds(D, P, r)`` of S0 and gets back (mu, S_rows): the full new mean and the
    same rows of the new covariance.  X, G, mu0 are replicated.  The only exchange is an all-gather of the
    owned COLUMNS of SG = G S0 (B x D/P doubles per rank: 256 KiB at D=4096, B=64, P=8); the D^2-sized reads
    and writes -- the HBM-bound part -- are divided by P, unlike the batch-sharded form above."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, D = X.shape
    lo, hi = row_bounds(D, world, rank)
    assert S0_rows.shape == (hi - lo, D), f"rank {rank} must hold rows [{lo}, {hi}) of S0"
    SGc = eng.gsm_rows_stage(G, S0_rows)
    if world == 1:
        SG = SGc
    else:
        per = -(-D // world)
        send = eng.zeros(B, per)
        send[:, :hi - lo] = SGc
        recv = eng.empty(world * B, per)
        _all_gather(_as_torch(recv), _as_torch(send), group)
        SG = eng.empty(B, D)
        for p in range(world):
            plo, phi = row_bounds(D, world, p)
            SG[:, plo:phi] = recv[p * B:(p + 1) * B, :phi - plo]
    rec = eng.gsm_records(X, G, mu0, SG)
    return eng.gsm_apply_rows(rec, mu0, S0_rows, lo, out=out)


def _all_reduce_sum(t, group=None):
    """in-place SUM all-reduce on the group's backend (device tensors staged through the host for a gloo group)"""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def col_bounds(D, world, rank):
    """Columns [lo, hi) of the square factor owned by ``rank``: equal, tile-aligned blocks (D must be a multiple of
    64 * world: one fixed-size all-gather of the sample slices, whole 64-column tiles for the update kernels)."""
    assert D % (64 * world) == 0, f"column sharding needs D ({D}) to be a multiple of 64 x the number of ranks ({world})"
    per = D // world
    return rank * per, (rank + 1) * per


def col_gather_samples(eng, X_cols, group=None, stats=None):
    """(B, D) samples from every rank's (B, D / P) slice: ONE all-gather of B D / P doubles per rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return X_cols
    B, nc = X_cols.shape
    recv = eng.empty(world * B, nc)
    if stats is not None:
        stats["all_gather_bytes_per_rank"] = B * nc * 8
        stats["collectives"] = stats.get("collectives", 0) + 1
    _all_gather(_as_torch(recv), _as_torch(X_cols), group)
    r = recv if isinstance(recv, torch.Tensor) else _as_torch(recv).numpy()
    X = eng.empty(B, world * nc)
    for p in range(world):
        X[:, p * nc:(p + 1) * nc] = r[p * B:(p + 1) * B]
    return X


def col_sharded_gsm_factor_update(eng, Z, X, G, mu0, F0_cols, group=None, out=None, flag=None, n_reverts=None, stats=None):
    """Factor-form GSM update with the square factor sharded by COLUMN BLOCKS (gsm_numpy.py:27-55 in
    the form of SURVEY A.2): (mu, F_cols, flag).  Every rank holds the columns ``col_bounds(D, P, r)`` of F0 (Sigma = F0^T F0)
    and gets back the same columns of the new factor and ITS entries of the new mean (``mu`` is full length; only the owned
    entries are written).  Z (the whitened draws), X (the samples, e.g. from ``col_gather_samples``) and G (their scores) are
    replicated.  One exchange: the all-reduce of the partial products G[:, C] F0[:, C]^T to W = G F0^T (B D doubles); the
    2B x 2B chain is replicated (identical inputs and arithmetic: every rank takes the same accept / revert decision) and the
    update reads and writes the owned block only -- the D^2-sized traffic and memory of the factor form are divided by P."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, D = Z.shape
    lo, hi = col_bounds(D, world, rank)
    assert tuple(F0_cols.shape) == (D, hi - lo), f"rank {rank} must hold columns [{lo}, {hi}) of F0"
    W = eng.gsm_factor_w_partial(G, lo, F0_cols)
    if world > 1:
        if stats is not None:
            stats["all_reduce_bytes"] = B * D * 8
            stats["collectives"] = stats.get("collectives", 0) + 1
        if isinstance(W, torch.Tensor):
            _all_reduce_sum(W, group)
        else:
            t = _as_torch(W)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            W = t.numpy()
    return eng.gsm_factor_apply_cols(Z, W, X, mu0, F0_cols, lo, out=out, flag=flag, n_reverts=n_reverts)


def sharded_bam_update(eng, X_local, G_local, mu0, S0, reg, jitter=0.0, group=None, out=None, flag=None, stats=None):
    """(mu, S, flag) of the BaM update for the union of all ranks' samples (gsmvi/bam.py:72-114;
    BASELINE config 4: B=128 sharded 16 per GPU).  BaM's statistics couple all samples (batch means
    and the (B+1) x (B+1) matrix function), so ranks all-gather their (B/P, D) samples and scores
    (2 (B/P) D doubles per rank: 256 KiB at D=1024, B=128, P=8 -- the part worth sharding is the
    user's score evaluation) and every replica runs the identical update."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return eng.bam_update(X_local, G_local, mu0, S0, reg, jitter, out=out, flag=flag)
    Bl, D = X_local.shape
    packed = eng.empty(Bl, 2 * D)
    packed[:, :D] = X_local
    packed[:, D:] = G_local
    allp = eng.empty(Bl * world, 2 * D)
    if stats is not None:                        # what this rank contributes to the ONE collective of the update (tests: SURVEY 8(e))
        stats["bytes_per_rank"] = int(np.prod(packed.shape)) * 8
        stats["collectives"] = stats.get("collectives", 0) + 1
    _all_gather(_as_torch(allp), _as_torch(packed), group)
    if not isinstance(allp, torch.Tensor):
        allp = _as_torch(allp).numpy()
    return eng.bam_update(allp[:, :D], allp[:, D:], mu0, S0, reg, jitter, out=out, flag=flag)


def sharded_bam_factor_update(eng, Z, X_local, G_local, mu0, F0, reg, group=None, out=None, flag=None, n_reverts=None,
                              stats=None):
    """(mu, F, flag) of the factor-form BaM update (Sigma = F^T F; engine.bam_factor_update) for the union of all ranks'
    samples -- BASELINE config 4 ("B=128 sharded 16/GPU") without a D x D covariance or a D^3 step on any rank.
    Z (B, D): the whitened draws of ALL samples, replicated (every rank draws the same counter-based stream);
    X_local, G_local: samples and scores of this rank's B/P rows.  As in ``sharded_bam_update`` the statistics couple all
    samples, so the (x_b, g_b) rows are all-gathered (2 (B/P) D doubles per rank) and every replica runs the identical
    update: fixed summation orders => replicas stay bit-identical."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return eng.bam_factor_update(Z, X_local, G_local, mu0, F0, reg, out=out, flag=flag, n_reverts=n_reverts)
    Bl, D = X_local.shape
    packed = eng.empty(Bl, 2 * D)
    packed[:, :D] = X_local
    packed[:, D:] = G_local
    allp = eng.empty(Bl * world, 2 * D)
    if stats is not None:
        stats["bytes_per_rank"] = int(np.prod(packed.shape)) * 8
        stats["collectives"] = stats.get("collectives", 0) + 1
    _all_gather(_as_torch(allp), _as_torch(packed), group)
    if not isinstance(allp, torch.Tensor):
        allp = _as_torch(allp).numpy()
    return eng.bam_factor_update(Z, allp[:, :D], allp[:, D:], mu0, F0, reg, out=out, flag=flag, n_reverts=n_reverts)
