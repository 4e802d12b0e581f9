#!/usr/bin/env python3
"""Round-5 verdict, item 4 (second half): the 1 -> 8 GPU curve cannot be measured in this pool (one GPU per call), so this writes
the ANALYTIC MODEL SURVEY 8(e) asks for when only one GPU is reachable -- from quantities measured on the one GPU plus the xGMI
figures of the task description (7 links x ~153 GB/s per GPU, point to point):

  measured here   * the local stage on B/P rows and the combined apply on B rows, for P = 1, 2, 4, 8 (what each rank of a
                    batch-sharded update runs: gsm-vi_amd/dist.py)
                  * the built-in score and the sampler on B/P rows
                  * an RCCL all_gather_into_tensor at WORLD SIZE 1 on this GPU for every message size of the model: host
                    enqueue time and device completion time (the floor a real collective cannot beat)
  assumed         * wire time of the all-gather on fully connected xGMI: every rank sends its slice to its P - 1 peers over
                    P - 1 different links at once: m / (eff * 153 GB/s), eff = 0.7
                  * one-way hop latency alpha = 5 us added once ("direct") or P - 1 times ("ring")
  predicted       U(P) = 1 / (stage(B/P) + collective(P) + apply(B)),   F(P) = 1 / (iteration(1) - [score, sampler and stage
                  savings on B/P rows] + collective(P))

Every entry with P > 1 is marked "model, not measured".  usage: scaling_model.py out.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import gsmvi_amd  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/scaling_model.json"
torch.cuda.set_device(0)
eng = gsmvi_amd.get_engine()
LINK_GBS, EFF, ALPHA_US = 153.0, 0.7, 5.0


def b2b(fn, warm=10, n=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def state(D, B, seed):
    g = torch.Generator(device=eng.device)
    g.manual_seed(seed)
    kw = dict(dtype=torch.float64, device=eng.device, generator=g)
    m = torch.rand(D, **kw)
    pd = 0.5 + torch.rand(D, **kw)
    U = torch.randn(D, 8, **kw) / np.sqrt(D)
    P = (torch.diag(pd) + U @ U.T).contiguous()
    mu0 = torch.randn(D, **kw)
    A = torch.randn(D, D, **kw)
    S0 = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device=eng.device)
    S0 = (0.5 * (S0 + S0.T)).contiguous()
    F0 = torch.linalg.cholesky(S0).T.contiguous()
    Z = torch.randn(B, D, **kw)
    X = (mu0[None, :] + Z @ F0).contiguous()
    G = eng.gaussian_score(X, m, P)
    return dict(m=m, P=P, mu0=mu0, S0=S0, F0=F0, Z=Z, X=X, G=G)


res = {"what": __doc__, "assumptions": {"xgmi_link_GBs": LINK_GBS, "wire_efficiency": EFF, "hop_latency_us": ALPHA_US},
       "device": torch.cuda.get_device_name(0), "rccl_world1": {}, "configs": {}}

# ---- RCCL at world size 1: the collective's floor on this box -----------------------------------------------------------------
rccl_ok = True
try:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
except Exception as e:                                           # noqa: BLE001
    rccl_ok = False
    res["rccl_world1"]["error"] = f"{type(e).__name__}: {e}"[:300]


def rccl_floor(nbytes):
    key = str(nbytes)
    if key in res["rccl_world1"] or not rccl_ok:
        return res["rccl_world1"].get(key)
    n8 = max(1, nbytes // 8)
    src, dst = torch.zeros(n8, dtype=torch.float64, device=eng.device), torch.zeros(n8, dtype=torch.float64, device=eng.device)
    for _ in range(20):
        dist.all_gather_into_tensor(dst, src)
    torch.cuda.synchronize()
    n = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        dist.all_gather_into_tensor(dst, src)
    e1.record()
    t_enq = (time.perf_counter() - t0) / n * 1e6
    e1.synchronize()
    res["rccl_world1"][key] = {"bytes": nbytes, "host_enqueue_us": t_enq, "device_us_back_to_back": e0.elapsed_time(e1) * 1e3 / n}
    return res["rccl_world1"][key]


def collective(P, m_bytes):
    """(direct, ring) microseconds for an all-gather of m_bytes per rank among P ranks"""
    if P == 1:
        return 0.0, 0.0
    fl = rccl_floor(m_bytes)
    floor = max(fl["host_enqueue_us"], fl["device_us_back_to_back"]) if fl else 10.0
    wire = m_bytes / (EFF * LINK_GBS * 1e3)                       # us
    return floor + ALPHA_US + wire, floor + (P - 1) * (ALPHA_US + wire)


def allreduce(P, m_bytes):
    """(direct, ring) microseconds for a SUM all-reduce of m_bytes among P ranks: a reduce-scatter and an all-gather of m / P
    per rank, over P - 1 links at once (direct) or around the ring"""
    if P == 1:
        return 0.0, 0.0
    fl = rccl_floor(m_bytes)
    floor = max(fl["host_enqueue_us"], fl["device_us_back_to_back"]) if fl else 10.0
    wire = (m_bytes / P) / (EFF * LINK_GBS * 1e3)
    return floor + 2.0 * (ALPHA_US + wire), floor + 2.0 * (P - 1) * (ALPHA_US + wire)


# ---- the COLUMN-sharded factor form (gsm-vi_amd/dist.py: col_sharded_gsm_factor_update): every stage a rank
# runs is measured here on a D x D / P block; the all-gather of the sample slices and the all-reduce of W are modelled ----------
for name, D, B in (("c3_cols", 1024, 32), ("c5_cols", 4096, 64)):
    st = state(D, B, 7)
    flag = eng.new_flag()
    rows = {}
    for Pn in (1, 2, 4, 8):
        nc = D // Pn
        Fc = st["F0"][:, :nc].contiguous()
        Fo, mu = eng.empty(D, nc), eng.empty(D)
        Xc = eng.empty(B, nc)
        W = eng.gsm_factor_w_partial(st["G"], 0, Fc)
        t_sample = b2b(lambda: eng.sample_cols(st["Z"], st["mu0"][:nc], Fc, out=Xc), 5, 50)
        Wp = eng.empty(B, D)
        t_w = b2b(lambda: eng.gsm_factor_w_partial(st["G"], 0, Fc, out=Wp), 5, 50)
        t_apply = b2b(lambda: eng.gsm_factor_apply_cols(st["Z"], W, st["X"], st["mu0"], Fc, 0, out=(mu, Fo), flag=flag), 5, 40)
        Go = eng.empty(B, D)
        t_score = b2b(lambda: eng.gaussian_score(st["X"], st["m"], st["P"], out=Go))      # replicated: every rank scores all B samples
        ag_d, ag_r = collective(Pn, B * nc * 8)
        ar_d, ar_r = allreduce(Pn, B * D * 8)
        upd = t_w + t_apply
        it = t_sample + t_score + upd
        rows[str(Pn)] = {
            "columns_per_rank": nc, "factor_block_bytes": D * nc * 8, "all_gather_bytes_per_rank": B * nc * 8 if Pn > 1 else 0,
            "all_reduce_bytes": B * D * 8 if Pn > 1 else 0, "collectives_per_update": 0 if Pn == 1 else 2,
            "measured_us": {"sample_slice": t_sample, "score_all_rows": t_score, "partial_w": t_w, "apply_owned_block": t_apply},
            "collective_us": {"direct": ag_d + ar_d, "ring": ag_r + ar_r},
            "U_updates_per_s": {"direct": 1e6 / (upd + ar_d), "ring": 1e6 / (upd + ar_r)},
            "F_iterations_per_s": {"direct": 1e6 / (it + ag_d + ar_d), "ring": 1e6 / (it + ag_r + ar_r)},
            "status": "measured on one GPU (no collective)" if Pn == 1 else "model, not measured",
        }
        del Fc, Fo, W, Wp
        torch.cuda.empty_cache()
    res["configs"][name] = {"D": D, "B": B, "kind": "gsm_factor, column-sharded (GSM.fit(shard='cols'))", "by_world_size": rows}
    print(name, json.dumps(rows), flush=True)
    del st
    torch.cuda.empty_cache()

for name, D, B, kind in (("c3", 1024, 32, "gsm_dense"), ("c4", 1024, 128, "bam_dense"), ("c5", 4096, 64, "gsm_factor")):
    st = state(D, B, 7)
    mu, S, Fo, flag = eng.empty(D), eng.empty(D, D), eng.empty(D, D), eng.new_flag()
    tgt = gsmvi_amd.GaussianTarget(st["m"].cpu().numpy(), precision=st["P"].cpu().numpy())
    rows = {}
    it1 = None
    for Pn in (1, 2, 4, 8):
        Bl = B // Pn
        Xl, Gl, Zl = st["X"][:Bl], st["G"][:Bl], st["Z"][:Bl]
        Xo, Go = eng.empty(Bl, D), eng.empty(Bl, D)
        t_score = b2b(lambda: eng.gaussian_score(Xl, st["m"], st["P"], out=Go))
        t_sample = b2b(lambda: eng.sample(Zl, st["mu0"], st["F0"], out=Xo))
        if kind == "gsm_dense":
            rec = eng.empty(B, eng.record_len(D))
            t_stage = b2b(lambda: eng.gsm_local_stage(Xl, Gl, st["mu0"], st["S0"], out=rec[:Bl]))
            eng.gsm_local_stage(st["X"], st["G"], st["mu0"], st["S0"], out=rec)
            t_apply = b2b(lambda: eng.gsm_apply(rec, st["mu0"], st["S0"], out=(mu, S)))
            t_check = b2b(lambda: eng.potrf(S, out=Fo, flag=flag), 3, 20)
            msg = Bl * eng.record_len(D) * 8
        elif kind == "gsm_factor":
            rec = eng.empty(B, eng.record_len(D))
            t_stage = b2b(lambda: eng.gsm_factor_local_stage(Zl, Xl, Gl, st["mu0"], st["F0"], out=rec[:Bl]), 5, 40)
            eng.gsm_factor_local_stage(st["Z"], st["X"], st["G"], st["mu0"], st["F0"], out=rec)
            t_apply = b2b(lambda: eng.gsm_factor_apply(st["Z"], rec, st["mu0"], st["F0"], out=(mu, Fo), flag=flag), 5, 40)
            t_check = 0.0
            msg = Bl * eng.record_len(D) * 8
        else:                                                     # BaM: the (x_b, g_b) rows are gathered, the update is replicated
            t_stage = 0.0
            t_apply = b2b(lambda: eng.bam_update(st["X"], st["G"], st["mu0"], st["S0"], 1.0, 1e-6, out=(mu, S), flag=flag), 5, 40)
            t_check = b2b(lambda: eng.potrf(S, out=Fo, flag=flag), 3, 20)
            msg = 2 * Bl * D * 8
        direct, ring = collective(Pn, msg)
        upd = t_stage + t_apply
        it = t_sample + t_score + upd + t_check
        if Pn == 1:
            it1 = it
        rows[str(Pn)] = {
            "rows_per_rank": Bl, "message_bytes_per_rank": msg, "collectives_per_update": 0 if Pn == 1 else 1,
            "measured_us": {"sampler": t_sample, "score": t_score, "local_stage": t_stage, "apply_all_rows": t_apply,
                            "accept_test_potrf": t_check},
            "collective_us": {"direct": direct, "ring": ring},
            "U_updates_per_s": {"direct": 1e6 / (upd + direct), "ring": 1e6 / (upd + ring)},
            "F_iterations_per_s": {"direct": 1e6 / (it + direct), "ring": 1e6 / (it + ring)},
            "status": "measured on one GPU (no collective)" if Pn == 1 else "model, not measured",
        }
    res["configs"][name] = {"D": D, "B": B, "kind": kind, "by_world_size": rows,
                            "note": "sum of back-to-back kernel times; the fit loops of configs.json overlap a little more"}
    print(name, json.dumps(rows), flush=True)
    del st, tgt
    torch.cuda.empty_cache()
if rccl_ok:
    dist.destroy_process_group()
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
