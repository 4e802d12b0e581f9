#!/usr/bin/env python3
"""A/B of the dense update's three kernels between this tree's library and a previous round's (gsm-vi_amd/libgsmvi_hip_<tag>.so,
GSMVI_HIP_LIB_VARIANT=<tag>), same box, same process order: dispatch-event averages over HBM-cold ring instances (bench.py's
own method).  usage: cov_ab_rounds.py   (spawns itself once per library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, gsmvi_amd, bench
    eng = gsmvi_amd.get_engine()
    out = {}
    for D, B, n_inst in ((1024, 32, 24), (4096, 32, 3), (2048, 32, 8)):
        inst, m, P = bench.make_instances(eng, D, B, n_inst, seed0=3)
        eng.set_profiling(True)
        kt = {"panel": [], "scalars": [], "cov_update": []}
        for k in range(6 * n_inst):
            it = inst[k % n_inst]
            eng.gsm_update(it["X"], it["G"], it["mu0"], it["S0"], out=(it["mu"], it["S"]))
            pr = eng.get_profile()
            if k >= n_inst:
                for key in kt:
                    kt[key].append(pr[key] * 1e3)
        eng.set_profiling(False)
        out[f"{D}x{B}"] = {k: [round(float(np.mean(v)), 3), round(float(np.min(v)), 3)] for k, v in kt.items()}
        del inst
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out))
    sys.exit(0)
for rep in range(2):
    for tag in ("r04", ""):
        env = dict(os.environ)
        if tag:
            env["GSMVI_HIP_LIB_VARIANT"] = tag
        else:
            env.pop("GSMVI_HIP_LIB_VARIANT", None)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, env=env)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        print(f"{tag or 'this tree':10s}", line[0][7:] if line else p.stderr[-500:], flush=True)
