#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats and, in separate passes, the HBM
# traffic counters for the bench workload.  Output: gpurun_out/prof_$1/ (copy summaries to profiles/).
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
# evidence chain (round-4 verdict, weak 7): which build, which box -- written FIRST, published by scripts/publish_profiles.py
cp $ROOT/gsm-vi_amd/BUILD_INFO.json $OUT/BUILD_INFO.json 2>/dev/null || echo '{"error": "no BUILD_INFO.json: run scripts/stamp_build.py before gpurun"}' > $OUT/BUILD_INFO.json
python3 - "$OUT" "$ROOT" <<'PYBOX'
import hashlib, json, os, platform, subprocess, sys, time
out, root = sys.argv[1], sys.argv[2]
def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=60).stdout.strip()
    except Exception as e:
        return f"failed: {e}"
lib = os.path.join(root, "gsm-vi_amd", "libgsmvi_hip.so")
box = {"hostname": platform.node(), "utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
       "gpu_unique_id": sh("/opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | grep 'GPU\\[' | head -2 | tr -s ' \\t' ' '"),
       "gpu_serial": sh("/opt/rocm/bin/rocm-smi --showserial 2>/dev/null | grep 'GPU\\[' | head -2 | tr -s ' \\t' ' '"),
       "rocm": sh("cat /opt/rocm/.info/version 2>/dev/null"), "kernel": platform.release(), "cpus": os.cpu_count(),
       "library_sha256_on_box": hashlib.sha256(open(lib, "rb").read()).hexdigest()}
# (round-5 advice) name and uuid of the GPU AGENT: rocminfo lists the CPU agents first ("Uuid: CPU-XX") and the marketing name
# differs by box ("AMD Instinct MI355X", "AMD Radeon Graphics"), so the fields are taken from the agent whose Name is gfx*
agent, gpu = {}, None
for line in sh("/opt/rocm/bin/rocminfo 2>/dev/null").splitlines():
    if line.startswith("Agent ") or line.startswith("*******"):
        if agent.get("Name", "").startswith("gfx") and gpu is None:
            gpu = agent
        agent = {} if line.startswith("Agent ") else agent
    elif ":" in line:
        k, v = line.split(":", 1)
        agent.setdefault(k.strip(), v.strip())
if gpu is None and agent.get("Name", "").startswith("gfx"):
    gpu = agent
box["gpu_name"] = (gpu or {}).get("Marketing Name", "")
box["gpu_uuid"] = (gpu or {}).get("Uuid", "")
box["gpu_arch"] = (gpu or {}).get("Name", "")
if not box["gpu_name"]:                                  # (some boxes of the pool print an empty marketing name for the GPU agent)
    box["gpu_name"] = sh("/opt/rocm/bin/rocm-smi --showproductname 2>/dev/null | grep -i -m1 'card series\\|card model' | tr -s ' \\t' ' '") or box["gpu_arch"]
if not box["gpu_uuid"].startswith("GPU-") or not box["gpu_arch"].startswith("gfx"):
    print("collect_profiles: no GPU agent found in rocminfo", file=sys.stderr)
    json.dump(box, open(os.path.join(out, "box.json"), "w"), indent=1)
    sys.exit(3)
json.dump(box, open(os.path.join(out, "box.json"), "w"), indent=1)
print(json.dumps(box))
PYBOX
[ $? -eq 0 ] || { echo "collect_profiles: box identification failed"; exit 3; }
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 420 --warmup 42 --no-cpu-baseline --no-large-point --no-callpath"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
# (each counter pass up to three times: the profiler's counter tool segfaulted inside a kernel launch once in ~15 passes of
# this round -- a pass that leaves no counter file is repeated, its first log kept beside the final one)
pmc_pass() {   # name, counters...
  local name=$1; shift
  for try in 1 2 3; do
    rm -rf $OUT/$name
    timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS --no-graph > $OUT/$name.log 2>&1
    if ls $OUT/$name/*/*counter_collection.csv > /dev/null 2>&1; then break; fi
    cp $OUT/$name.log $OUT/$name.failed_try$try.log
  done
}
pmc_pass pmc_fetch FETCH_SIZE
pmc_pass pmc_write WRITE_SIZE
# MFMA pipe occupancy (north_star: "MFMA utilisation against gfx950 peak"): busy cycles of the matrix pipe and the
# fp64 MFMA op count, against the time the GPU was active during the dispatch
pmc_pass pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE
# the >= 0.50 HBM-roofline point of the covariance kernel (D=4096, B=32) and the fit-iteration kernel tables
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_d4096 -- python3 $ROOT/bench.py --D 4096 --B 32 --steps 60 --warmup 12 --no-cpu-baseline --no-callpath > $OUT/trace_d4096.log 2>&1
for cfg in "1024 32 factor" "1024 32 dense" "4096 64 factor" "256 8 factor" "1024 32 bam" "1024 32 bamf" "1024 128 bam" "1024 128 bamf"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fit_$tag -- python3 $ROOT/scripts/factor_prof.py $cfg > $OUT/fit_$tag.log 2>&1
  echo "== fit iteration kernels, D B method = $cfg (41 iterations; name, calls, avg ns, % of GPU time)" >> $OUT/fit_iteration_kernels.txt
  python3 $ROOT/scripts/prof_top.py $OUT/fit_$tag 18 >> $OUT/fit_iteration_kernels.txt 2>&1
done
# BASELINE config 4 (BaM update, D=1024, B=128): per-kernel table of 60 plain updates
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_update -- python3 $ROOT/scripts/c4_update_bench.py prof > $OUT/c4_update.log 2>&1
echo "== scripts/c4_update_bench.py prof: 60 dense BaM updates at D=1024, B=128 (name, calls, avg ns, % of GPU time)" > $OUT/c4_update_kernels.txt
python3 $ROOT/scripts/prof_top.py $OUT/c4_update 24 >> $OUT/c4_update_kernels.txt 2>&1
echo "== bench.py --D 4096 --B 32 (kernel-trace stats)" > $OUT/d4096_b32_kernels.txt
python3 $ROOT/scripts/prof_top.py $OUT/trace_d4096 14 >> $OUT/d4096_b32_kernels.txt 2>&1
cd $ROOT
# un-profiled figures of the same commit: every BASELINE config, the c4 update, the persistent-kernel A/B, the chol64_blk
# harness variants and the in-pipeline race check, bench.py at its default flags
python3 scripts/configs_bench.py $OUT/configs.json > $OUT/configs.log 2>&1
python3 scripts/c4_update_bench.py > $OUT/c4_update.txt 2>&1
python3 scripts/cov_p_ab.py > $OUT/cov_persistent_ab.txt 2>&1
python3 scripts/offgrid_bench.py after $OUT/offgrid_after.json > $OUT/offgrid_after.log 2>&1
python3 scripts/callpath_bench.py $OUT/callpath.json > $OUT/callpath.log 2>&1
python3 scripts/fit_kc_ab.py 1024 32 2>&1 | grep -v amdgpu.ids > $OUT/fit_kc_ab.txt
python3 scripts/bam_basis_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/bam_basis_ab.txt
python3 scripts/cov_ab_rounds.py 2>&1 | grep -v amdgpu.ids > $OUT/cov_ab_rounds.txt
python3 scripts/bam_graph_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/bam_graph_ab.txt
python3 scripts/gsm_graph_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/gsm_graph_ab.txt
# factor-form BaM update, orthogonal basis (1) and round-4 basis (0): timed alone (median of event pairs) and back-to-back
for cfg in "1024 128" "1024 64" "1024 32" "256 8" "4096 64"; do
  for b in 1 0; do
    echo "== D B = $cfg, bam_basis = $b (inputs of scripts/configs_bench.py)" >> $OUT/bamf_backtoback.txt
    python3 scripts/bamf_trace.py $cfg $b cfg 2>&1 | grep -E "eager|host" >> $OUT/bamf_backtoback.txt
  done
done
# round 6: the jitter-period table of the factor-form BaM fit, batch sizes beyond the old ceilings, the Cholesky as one persistent
# launch against one launch per block step, 64-column panel strips at large D, the analytic multi-GPU model
python3 scripts/jitter_period.py $OUT/jitter_period.json > $OUT/jitter_period.log 2>&1
python3 scripts/bigbatch_bench.py $OUT/bigbatch.json > $OUT/bigbatch.log 2>&1
for d in 1 0; do POTRF_DAG=$d python3 scripts/potrf_rate.py 256 512 1024 2048 4096 6144 8192 12288 2>&1 | grep "^potrf" >> $OUT/potrf_rate.txt; done
for d in 1024 4096; do python3 scripts/potrf_dag_timeline.py $d 2>&1 | grep -v amdgpu.ids >> $OUT/potrf_dag_timeline.txt; done
python3 scripts/panel_w4_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/panel_w4_ab.txt
timeout 600 python3 scripts/scaling_model.py $OUT/scaling_model.json > $OUT/scaling_model.log 2>&1
python3 scripts/soak_round3.py 120 > $OUT/soak.txt 2>&1
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_driver_flags.json 2>> $OUT/bench.err
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
st = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if st:
    rows = [r for r in csv.DictReader(open(st[0])) if r["Name"].lstrip("void ").startswith("k_")]
    res["kernel_stats"] = {r["Name"].split("(")[0]: {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                    "min_ns": int(r["MinNs"]), "max_ns": int(r["MaxNs"])} for r in rows}
for name in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64",
             "GRBM_GUI_ACTIVE"):
    sub = 'fetch' if name == 'FETCH_SIZE' else ('write' if name == 'WRITE_SIZE' else 'mfma')
    files = glob.glob(out + f"/pmc_{sub}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == name:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    res[name] = {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in acc.items() if "k_" in k}
# HBM-side bytes per launch (MI355X_MICROARCH.md, HBM / rocprofv3 section): counters are in KiB; FETCH_SIZE is
# doubled on gfx950 (it reports half of a coalesced streaming read); WRITE_SIZE is taken as is.
def per_launch(kern):
    f = next((v["mean"] for k, v in res.get("FETCH_SIZE", {}).items() if kern in k), None)
    w = next((v["mean"] for k, v in res.get("WRITE_SIZE", {}).items() if kern in k), None)
    return None if f is None or w is None else {"fetch_KiB_raw": f, "write_KiB": w, "bytes": (2.0 * f + w) * 1024.0}
# MFMA pipe utilisation per kernel.  SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 matrix pipes (64 per
# v_mfma_f64_16x16x4_f64: checked against the op count); GRBM_GUI_ACTIVE is useless under per-dispatch counter
# collection (it spans the collection overhead), so the busy time per SIMD at the 2.4 GHz peak clock is set against the
# kernel's average duration from the kernel-trace pass.
mf = {}
for k, v in res.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).items():
    dur = next((d["avg_ns"] for n_, d in res.get("kernel_stats", {}).items() if n_ == k), None)
    busy_us = v["mean"] / 1024.0 / 2400.0
    mf[k] = {"mfma_busy_cycles_all_simds": v["mean"], "mfma_busy_us_per_simd_at_2.4GHz": busy_us,
             "kernel_avg_us": dur / 1e3 if dur else None, "mfma_util": busy_us / (dur / 1e3) if dur else None,
             "mfma_f64_flop": 512.0 * res.get("SQ_INSTS_VALU_MFMA_MOPS_F64", {}).get(k, {}).get("mean", 0.0)}
res["mfma_util"] = {k: v for k, v in mf.items() if v["mfma_busy_cycles_all_simds"] > 0}
cov, pan, sca = per_launch("k_gsm_cov_sym"), per_launch("k_panel_fast"), per_launch("k_gsm_scalars_fast")
if cov:
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 420 "
                         "--warmup 42 --no-cpu-baseline --no-graph` (scripts/collect_profiles.sh), D=1024 B=32",
               "units": "counters in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a coalesced "
                        "streaming read); WRITE_SIZE taken as is.  The x2 was calibrated on this box for the 8-B/lane "
                        "128-byte-segment loads of these kernels too (scripts/fetchcal.hip, profiles/r02/fetchcal.txt: "
                        "ratio 0.500 for 16-B/lane, 8-B/lane contiguous and 8-B/lane row segments).  The fetch side of "
                        "k_gsm_cov_sym is the upper triangle of S0 (4.33 MB at D=1024) plus the records re-fetched through "
                        "the fabric by each of the 8 XCD L2s (8 x 0.52 MB, served by the Infinity Cache)",
               "k_gsm_cov_update_fetch_KiB_raw": cov["fetch_KiB_raw"], "k_gsm_cov_update_write_KiB": cov["write_KiB"],
               "k_gsm_cov_update_bytes_per_launch": cov["bytes"],
               "k_panel_fast_bytes_per_launch": pan["bytes"] if pan else None,
               "k_gsm_scalars_bytes_per_launch": sca["bytes"] if sca else None,
               "mfma_util": {k.replace("void ", ""): {"busy_us_per_simd_at_2.4GHz": v["mfma_busy_us_per_simd_at_2.4GHz"],
                                                       "kernel_avg_us": v["kernel_avg_us"], "util": v["mfma_util"],
                                                       "f64_flop": v["mfma_f64_flop"]}
                             for k, v in res.get("mfma_util", {}).items()
                             if any(t in k for t in ("k_gsm_cov_sym", "k_panel_fast<2, false", "k_gsmf_update_f"))}},
              open(out + "/traffic.json", "w"), indent=1)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
# the whole GPU suite in ONE process on the same box, same library (GSMVI_PASS_PYTEST=1: the round's published pass)
if [ "${GSMVI_PASS_PYTEST:-0}" = "1" ]; then
  (cd $ROOT && timeout 1800 python3 -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -15 > $OUT/pytest_gpu.txt)
fi
# what THIS pass produced, with hashes: scripts/publish_profiles.py publishes only files listed here (the local gpurun_out/ keeps
# files of earlier passes: gpurun merges, it does not mirror)
python3 - "$OUT" <<'PYLIST'
import hashlib, json, os, sys
out = sys.argv[1]
files = {}
for root, _, names in os.walk(out):
    for n in names:
        if n == "files.json":
            continue
        p = os.path.join(root, n)
        files[os.path.relpath(p, out)] = {"sha256": hashlib.sha256(open(p, "rb").read()).hexdigest(), "bytes": os.path.getsize(p)}
json.dump(files, open(os.path.join(out, "files.json"), "w"), indent=1)
print(len(files), "files listed")
PYLIST
