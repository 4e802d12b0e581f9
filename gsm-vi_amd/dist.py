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


def shard_bounds(B, world, rank):
    """Contiguous equal shards; B must be divisible by the world size (one fixed-size all-gather)."""
    assert B % world == 0, f"batch size {B} must be divisible by the number of ranks {world}"
    per = B // world
    return rank * per, (rank + 1) * per


def sharded_gsm_update(eng, X_local, G_local, mu0, S0, group=None, rec_all=None, out=None):
    """(mu, S) of gsm_update for the union of all ranks' samples (gsmvi/gsm_numpy.py:27-55).

    X_local, G_local: this rank's (B/P, D) samples and scores; mu0, S0 replicated.  ``rec_all``
    (B, 2D+4) may be passed to avoid allocating the gather buffer every call."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rec_local = eng.gsm_local_stage(X_local, G_local, mu0, S0)
    if world == 1:
        rec = rec_local
    else:
        Bl, L = rec_local.shape
        if rec_all is None:
            rec_all = eng.empty(Bl * world, L)
        t_all, t_loc = _as_torch(rec_all), _as_torch(rec_local)
        dist.all_gather_into_tensor(t_all, t_loc.contiguous(), group=group)
        rec = rec_all if isinstance(rec_all, torch.Tensor) else t_all.numpy()
    return eng.gsm_apply(rec, mu0, S0, out=out)
