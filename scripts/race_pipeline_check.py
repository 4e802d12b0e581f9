#!/usr/bin/env python3
"""The chol64_blk replica race INSIDE the real pipeline (DESIGN section 8, round-4 verdict item 2b).

Four builds of the whole library run the same calls in separate processes:
    product       libgsmvi_hip.so                                   the shipped header
    delay         + CHOLB_TEST_REPLICA_DELAY=1   (make variant ...)  replica waves ~3.5 us late in every panel, shipped header
    oldwb         + CHOLB_TEST_OLD_WRITEBACK                        the pre-fix in-place write-back, natural schedule
    oldwb_delay   both                                              the pre-fix write-back under the late-replica schedule
Every kernel that calls chol64_blk with replicas takes part: k_chol128w / k_chol128 (factor update, 2B = 128), the one-workgroup
chain (2B = 64), k_potrf_step8 (dense Cholesky), k_bam_cholw (dense BaM).  Expected: delay == product bit for bit (the shipped
header does not care when the replicas run); oldwb == product in a quiet pipeline (the race needs a late replica);
oldwb_delay deviates at O(1) or flags a failure -- the pre-fix header fails ON DEMAND in the real kernels.
usage: race_pipeline_check.py            (parent: runs the four children and compares)"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = ["", "delay", "oldwb", "oldwb_delay"]


def child(out_path):
    import numpy as np, torch, gsmvi_amd
    import _inputs as orc
    eng = gsmvi_amd.get_engine()
    res = {}
    for D, B in ((4096, 64), (1024, 64), (1024, 32)):
        st = orc.make_update_state(D, B, D + B)
        F0 = st["L"].T.copy()
        dv = [eng.asarray(st[k]) for k in ("Z", "samples", "vs", "mu0")] + [eng.asarray(F0)]
        mu, F, flag = eng.gsm_factor_update(*dv)
        res[f"gsmf_{D}_{B}_F"] = F.cpu().numpy(); res[f"gsmf_{D}_{B}_flag"] = np.array([eng.read_flag(flag)])
        if 2 * B <= 128 or True:
            mub, Fb, fb = eng.bam_factor_update(*dv, 1.0)
            res[f"bamf_{D}_{B}_F"] = Fb.cpu().numpy(); res[f"bamf_{D}_{B}_flag"] = np.array([eng.read_flag(fb)])
    st = orc.make_update_state(1024, 128, 3)
    S0 = eng.asarray(st["S0"])
    R, fl = eng.potrf(S0)
    res["potrf_1024_R"] = R.cpu().numpy(); res["potrf_1024_flag"] = np.array([eng.read_flag(fl)])
    mu, S, f = eng.bam_update(eng.asarray(st["samples"]), eng.asarray(st["vs"]), eng.asarray(st["mu0"]), S0, 1.0, 0.0)
    res["bam_1024_128_S"] = S.cpu().numpy(); res["bam_1024_128_flag"] = np.array([eng.read_flag(f)])
    np.savez(out_path, **res)


if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(sys.argv[2])
    sys.exit(0)

import numpy as np
outs = {}
for v in VARIANTS:
    path = f"/tmp/race_{v or 'product'}.npz"
    env = dict(os.environ)
    if v:
        env["GSMVI_HIP_LIB_VARIANT"] = v
    else:
        env.pop("GSMVI_HIP_LIB_VARIANT", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        print(f"[{v or 'product'}] child failed:\n{r.stderr[-1500:]}")
        sys.exit(1)
    outs[v or "product"] = dict(np.load(path))
ref = outs["product"]
print(f"{'result':22s} " + " ".join(f"{k:>24s}" for k in ("delay", "oldwb", "oldwb_delay")))
summary = {k: [] for k in ("delay", "oldwb", "oldwb_delay")}
for key in sorted(ref):
    if key.endswith("_flag"):
        continue
    row = []
    for v in ("delay", "oldwb", "oldwb_delay"):
        a, b = outs[v][key], ref[key]
        fl = int(outs[v][key.rsplit("_", 1)[0] + "_flag"][0])
        if np.array_equal(a, b) and fl == int(ref[key.rsplit("_", 1)[0] + "_flag"][0]):
            row.append("bit-identical")
            summary[v].append(0.0)
        else:
            d = float(np.nanmax(np.abs(a - b)) / max(float(np.abs(b).max()), 1e-300)) if np.isfinite(a).any() else float("inf")
            row.append(f"dev {d:.1e} flag {fl}")
            summary[v].append(d if d == d else float("inf"))
    print(f"{key:22s} " + " ".join(f"{x:>24s}" for x in row))
ok = all(x == 0.0 for x in summary["delay"]) and any(x > 1e-6 for x in summary["oldwb_delay"])
print("delay (shipped header, late replicas): " + ("bit-identical everywhere" if all(x == 0.0 for x in summary["delay"]) else "DEVIATES"))
print("oldwb_delay (pre-fix header, late replicas): " + ("fails on demand" if any(x > 1e-6 for x in summary["oldwb_delay"]) else "did not deviate"))
print("VERDICT " + ("OK" if ok else "UNEXPECTED"))
