#!/usr/bin/env python3
"""Off-grid shapes next to their padded grid neighbours: update (U) and fit-iteration (F) rates.

For every (D, B) of SHAPES and its neighbour (Dp, Bp) = (D rounded up to 64, B rounded up to the next of 8/16/32/64/128):
  U  one update on device-resident inputs, four kinds: GSM dense, GSM factor form, BaM dense, BaM factor form
       direct  : the engine call on plain contiguous (D, B) tensors -- what a C caller with its own arrays gets
       padded  : the same call on D-padded state (Dp columns / rows, zero / identity border; B rows as given) -- what the fit
                 loops keep resident
       oneshot : gsmvi_amd.gsm_update / bam_update on CUDA tensors (includes whatever padding copies the package makes)
  F  GSM.fit (auto, dense) and BaM.fit (dense, factor) marginal iteration rates with the built-in Gaussian score
Timing: HIP events around single calls after warm-up (median), and a replayed hipGraph of back-to-back calls.
Usage: offgrid_bench.py <label> <out.json>; label "before" = the library of round 4 (GSMVI_HIP_LIB_VARIANT=r04 or a checkout
of that commit), "after" = this tree.  scripts/offgrid_report.py merges the two documents into profiles/r05/offgrid.json.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gsmvi_amd  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "after"
out_path = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/offgrid_{label}.json"
quick = os.environ.get("OFFGRID_QUICK", "0") != "0"
eng = gsmvi_amd.get_engine()
dev = eng.device
SHAPES = [(1000, 32), (1024, 20), (1000, 30), (784, 50), (500, 10), (2000, 24)]
if os.environ.get("OFFGRID_SHAPES"):                      # e.g. OFFGRID_SHAPES=2000x64,3000x64 (ad-hoc runs)
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["OFFGRID_SHAPES"].split(",")]


def up64(D):
    return (D + 63) // 64 * 64


def upB(B):
    for b in (8, 16, 32, 64, 128):
        if B <= b:
            return b
    return B


def ev_time(fn, warm=10, n=200):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def graph_time(fn, reps=8, nrep=10):
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):                               # shortest of three timed regions (~1 ms each)
        t0 = time.perf_counter()
        for _ in range(nrep):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / (nrep * reps) * 1e6


def timed(fn):
    try:
        r = {"eager_us": ev_time(fn, 5 if quick else 10, 40 if quick else 200)}
        try:
            r["graph_us"] = graph_time(fn, 4 if quick else 8, 4 if quick else 10)
        except Exception as e:  # noqa: BLE001
            r["graph_error"] = f"{type(e).__name__}: {e}"[:200]
            torch.cuda.synchronize()
        return r
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def make_state(D, B, seed=0):
    """SURVEY 8(d) synthetic inputs on the device; everything contiguous (D, B)."""
    g = torch.Generator(device=dev)
    g.manual_seed(100 + seed)
    kw = dict(dtype=torch.float64, device=dev, generator=g)
    m = torch.rand(D, **kw)
    L = torch.randn(D, D, **kw)
    cov_t = L @ L.T + 1e-3 * torch.eye(D, dtype=torch.float64, device=dev)
    P = torch.linalg.inv(cov_t)
    P = (0.5 * (P + P.T)).contiguous()
    mu0 = torch.randn(D, **kw)
    A = torch.randn(D, D, **kw)
    S0 = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device=dev)
    S0 = (0.5 * (S0 + S0.T)).contiguous()
    F0 = torch.linalg.cholesky(S0).T.contiguous()
    Z = torch.randn(B, D, **kw)
    X = (mu0[None, :] + Z @ F0).contiguous()
    G = (-(X - m[None, :]) @ P).contiguous()
    return dict(m=m, P=P, mu0=mu0, S0=S0, F0=F0, Z=Z, X=X, G=G)


def pad_state(st, D, B):
    """The same state with D padded to Dp: vectors and row panels zero-padded, S0 with an identity border, F0 with a zero
    border (x = mu + z F stays zero in the padding for any z)."""
    Dp = up64(D)
    def vec(v):
        o = torch.zeros(Dp, dtype=torch.float64, device=dev)
        o[:D] = v
        return o
    def rows(M):
        o = torch.zeros(M.shape[0], Dp, dtype=torch.float64, device=dev)
        o[:, :D] = M
        return o
    def mat(M, eye):
        o = torch.eye(Dp, dtype=torch.float64, device=dev) if eye else torch.zeros(Dp, Dp, dtype=torch.float64, device=dev)
        o[:D, :D] = M
        return o
    return dict(mu0=vec(st["mu0"]), S0=mat(st["S0"], True), F0=mat(st["F0"], False), Z=rows(st["Z"]), X=rows(st["X"]),
                G=rows(st["G"]))


def update_kinds(s, D, B):
    """closures of the four update kinds on the tensors of s (plain or padded)"""
    Dd = s["S0"].shape[0]
    mu, S, Fo = eng.empty(Dd), eng.empty(Dd, Dd), eng.empty(Dd, Dd)
    flag = eng.new_flag()
    kinds = {
        "gsm_dense": lambda: eng.gsm_update(s["X"], s["G"], s["mu0"], s["S0"], out=(mu, S)),
        "bam_dense": lambda: eng.bam_update(s["X"], s["G"], s["mu0"], s["S0"], 1.0, 1e-6, out=(mu, S), flag=flag),
    }
    if 2 * B <= min(D, 256):
        kinds["gsm_factor"] = lambda: eng.gsm_factor_update(s["Z"], s["X"], s["G"], s["mu0"], s["F0"], out=(mu, Fo), flag=flag)
        kinds["bam_factor"] = lambda: eng.bam_factor_update(s["Z"], s["X"], s["G"], s["mu0"], s["F0"], 1.0, out=(mu, Fo), flag=flag)
    return kinds


def fit_rates(D, B, st):
    tgt = gsmvi_amd.GaussianTarget(st["m"].cpu().numpy(), precision=st["P"].cpu().numpy())
    res = {}

    def marginal(run, n):
        run(3 * n - 1)                               # full-length warm-up (lazy buffers, graph capture of the iteration)
        torch.cuda.synchronize()
        t1s, t3s = [], []
        for _ in range(3):                           # one host hiccup (~25 ms seen) in a 30 - 100 ms region moves a single pass by
            t0 = time.perf_counter()                 # half: the SHORTEST of three per leg, then the difference of the legs (a best-
            run(n - 1)                               # of-passes on the rate itself is biased upwards when the short leg stalls)
            torch.cuda.synchronize()
            t1s.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            run(3 * n - 1)
            torch.cuda.synchronize()
            t3s.append(time.perf_counter() - t0)
        t1, t3 = min(t1s), min(t3s)
        return {"it_per_s": n / t1, "it_per_s_marginal": 2 * n / (t3 - t1), "n": n, "t1_ms": [t * 1e3 for t in t1s],
                "t3_ms": [t * 1e3 for t in t3s]}

    n = 60 if quick else 200
    for method in ("auto", "dense"):
        gsm = gsmvi_amd.GSM(D, tgt.lp, tgt.lp_g)
        try:
            res["gsm_" + method] = marginal(
                lambda k: gsm.fit(1, niter=k, batch_size=B, verbose=False, rng="device", method=method), n)
            res["gsm_" + method]["method_used"] = getattr(gsm, "method_used", None)
        except Exception as e:  # noqa: BLE001
            res["gsm_" + method] = {"error": f"{type(e).__name__}: {e}"[:300]}
    sched = lambda i: 100.0 / (1 + i)  # noqa: E731   (examples/example_bam.py:58)
    for method in ("dense", "factor"):
        if method == "factor" and 2 * B > min(D, 256):
            continue
        bam = gsmvi_amd.BaM(D, tgt.lp, tgt.lp_g)
        try:
            res["bam_" + method] = marginal(
                lambda k: bam.fit(1, sched, niter=k, batch_size=B, verbose=False, rng="device", method=method), max(30, n // 2))
        except Exception as e:  # noqa: BLE001
            res["bam_" + method] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return res


doc = {"label": label, "device": torch.cuda.get_device_name(0), "dtype": "f64", "shapes": {}}
try:
    import subprocess
    doc["git"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:  # noqa: BLE001
    doc["git"] = None
doc["library"] = os.path.basename(gsmvi_amd.library_path())
done = set()
for (D, B) in SHAPES:
    for (d, b, role) in ((D, B, "offgrid"), (up64(D), upB(B), "neighbour")):
        if (d, b) in done:
            continue
        done.add((d, b))
        st = make_state(d, b)
        ent = {"D": d, "B": b, "role": role, "neighbour": [up64(d), upB(b)], "U": {}, "F": {}}
        for kind, fn in update_kinds(st, d, b).items():
            ent["U"][kind] = {"direct": timed(fn)}
        if d % 64:
            sp = pad_state(st, d, b)
            for kind, fn in update_kinds(sp, d, b).items():
                ent["U"][kind]["padded"] = timed(fn)
            del sp
        mu_o = lambda: gsmvi_amd.gsm_update(st["X"], st["G"], st["mu0"], st["S0"])  # noqa: E731
        ent["U"]["gsm_dense"]["oneshot"] = {"eager_us": ev_time(mu_o, 5, 40 if quick else 100)}
        bo = lambda: gsmvi_amd.bam_update(st["X"], st["G"], st["mu0"], st["S0"], 1.0)  # noqa: E731
        ent["U"]["bam_dense"]["oneshot"] = {"eager_us": ev_time(bo, 5, 40 if quick else 100)}
        ent["F"] = fit_rates(d, b, st)
        doc["shapes"][f"{d}x{b}"] = ent
        print(f"{d}x{b}", json.dumps(ent), flush=True)
        del st
        torch.cuda.empty_cache()

os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(doc, open(out_path, "w"), indent=1)
