#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
python -m pytest tests -q -m gpu --maxfail=15 2>&1 | tail -80 > $OUT/tests_full.log
python scripts/race_pipeline_check.py > $OUT/race_pipeline_check.txt 2>&1
python scripts/c4_update_bench.py > $OUT/c4_ab.txt 2>&1
python bench.py --steps 100 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
python - > $OUT/cov_p_ab.txt 2>&1 <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gsmvi_amd
from bench import make_instances
eng = gsmvi_amd.get_engine()
for D, B in ((4096, 32), (4096, 64), (2048, 32), (8192, 32)):
    li, _, _ = make_instances(eng, D, B, 3, seed0=7)
    for tag, dbg in (("persistent", 0), ("one item per workgroup", 512)):
        eng.set_tuning("cov_dbg", dbg)
        eng.set_profiling(True)
        tl = []
        outs = []
        for kk in range(15):
            it = li[kk % 3]
            eng.gsm_update(it["X"], it["G"], it["mu0"], it["S0"], out=(it["mu"], it["S"]))
            if kk >= 3: tl.append(eng.get_profile()["cov_update"])
        eng.set_profiling(False)
        torch.cuda.synchronize()
        outs = [li[0]["S"].clone(), li[0]["mu"].clone()]
        if dbg == 0: ref = outs
        else: print("   bit-identical to the persistent kernel:", torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1]))
        t = float(np.mean(tl)) * 1e3
        alg = 16.0 * D * D + 16.0 * B * D
        nt = D // 32
        moved = (nt * (nt + 1) // 2) * 32 * 32 * 8.0 + 8.0 * D * D + 16.0 * B * D
        print(f"D={D} B={B} {tag:24s}: {t:7.1f} us  algorithmic {alg / t / 1e3:7.1f} GB/s ({alg / t / 1e3 / 8000:.3f} of 8 TB/s)  moved {moved / t / 1e3:7.1f} GB/s", flush=True)
    eng.set_tuning("cov_dbg", 0)
    del li
    torch.cuda.empty_cache()
PY
cat $OUT/tests_full.log | tail -30; cat $OUT/race_pipeline_check.txt $OUT/c4_ab.txt $OUT/cov_p_ab.txt; cat $OUT/bench.json | head -c 3000
