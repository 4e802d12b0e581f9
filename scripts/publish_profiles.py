#!/usr/bin/env python3
"""Run HERE after `gpurun -- bash scripts/collect_profiles.sh <tag>`: copies the summaries of gpurun_out/prof_<tag>/ that are
meant to be judged into profiles/<tag>/ and writes profiles/<tag>/MANIFEST.json -- for every published file its source (the
rocprofv3 output file name included), size and sha256, plus the build stamp (git commit, library hashes) and the box the pass
ran on.  Nothing is copied by hand (round-4 verdict, weak 7: a stale kernel_stats.csv and a duplicate had slipped in that way);
the script refuses duplicate kernel-stats files and a library hash that differs from the stamp.
usage: publish_profiles.py <tag> [--also-root-traffic]"""
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
final = os.path.join(ROOT, "profiles", tag)
# (round-5 advice) everything is STAGED in a temporary directory and swapped in after every check has passed: a pass that fails
# a hash check midway leaves profiles/<tag>/ as it was
dst = os.path.join(ROOT, "profiles", f".{tag}.staging")
if os.path.isdir(dst):
    shutil.rmtree(dst)
os.makedirs(dst)


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


listed = json.load(open(os.path.join(src, "files.json")))      # what the LAST pass on the box wrote (collect_profiles.sh), with hashes
box = json.load(open(os.path.join(src, "box.json")))
build = json.load(open(os.path.join(src, "BUILD_INFO.json")))
if box.get("library_sha256_on_box") != build["libraries"].get("libgsmvi_hip.so"):
    sys.exit("the library the box loaded is not the stamped one")
manifest = {"tag": tag, "build": build, "box": box, "files": {}}


def publish(src_rel, dst_name, note):
    sp = os.path.join(src, src_rel)
    if src_rel not in listed or not os.path.exists(sp):
        return False                             # not part of the last pass (a leftover of an earlier one in the local scratch)
    if sha256(sp) != listed[src_rel]["sha256"]:
        sys.exit(f"{src_rel}: local copy differs from what the box wrote")
    shutil.copyfile(sp, os.path.join(dst, dst_name))
    manifest["files"][dst_name] = {"source": os.path.join(f"gpurun_out/prof_{tag}", src_rel), "bytes": os.path.getsize(sp),
                                   "sha256": sha256(sp), "what": note}
    return True


def first(pattern):
    hits = sorted(h for h in glob.glob(os.path.join(src, pattern), recursive=True) if os.path.relpath(h, src) in listed)
    if len(hits) > 1:
        sys.exit(f"{pattern}: more than one file of the last pass matches: {hits}")
    return os.path.relpath(hits[0], src) if hits else None


plain = {
    "bench.json": "python bench.py (default flags), the JSON line",
    "bench_driver_flags.json": "python bench.py --steps 20 --warmup 3 (the driver's flags)",
    "configs.json": "scripts/configs_bench.py: U and F for every BASELINE config that fits one GPU",
    "offgrid_after.json": "scripts/offgrid_bench.py after (this tree)",
    "callpath.json": "scripts/callpath_bench.py: host / autograd score call path at c2 and c3, both fit methods",
    "fit_iteration_kernels.txt": "kernel tables of 41 fit iterations per (D, B, method), rocprofv3 --kernel-trace --stats",
    "c4_update_kernels.txt": "kernel table of 60 dense BaM updates at D=1024, B=128",
    "c4_update.txt": "scripts/c4_update_bench.py (un-profiled), incl. the low-rank staging A/B",
    "d4096_b32_kernels.txt": "kernel table of bench.py --D 4096 --B 32",
    "fit_kc_ab.txt": "scripts/fit_kc_ab.py: factor-form fit rate by the split-K count of the panel products",
    "cov_persistent_ab.txt": "scripts/cov_p_ab.py",
    "bam_basis_ab.txt": "scripts/bam_basis_ab.py: factor-form BaM update in the orthogonal basis (default) against the round-4 basis",
    "bam_graph_ab.txt": "scripts/bam_graph_ab.py: BaM.fit eager against graph=True (marginal iteration, capture cost)",
    "gsm_graph_ab.txt": "scripts/gsm_graph_ab.py: GSM.fit eager against graph=True by shape (marginal iteration)",
    "bamf_backtoback.txt": "scripts/bamf_trace.py: factor-form BaM update timed alone and back-to-back, both bases",
    "cov_ab_rounds.txt": "scripts/cov_ab_rounds.py: the dense update's three kernels, this tree against the round-4 library, same box",
    "traffic.json": "HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes + MFMA pipe utilisation",
    "rocprof_summary.json": "per-kernel stats and raw counters of the bench passes (summary.json of the collection)",
    "soak.txt": "run-to-run bit-identity soak over the (case, kind) pairs, library of this pass",
    "jitter_period.json": "scripts/jitter_period.py: factor-form BaM fit absorbing its jitter every K updates against the reference's loop (deviation and rate by K)",
    "bigbatch.json": "scripts/bigbatch_bench.py: U and F at D = 1024 for B = 128 ... 1024 (GSM dense, BaM dense), per-sample cost relative to B = 128",
    "potrf_rate.txt": "scripts/potrf_rate.py: gsmvi_potrf_f64 by size, k_potrf_dag (dag=1) against one launch per block step (dag=0)",
    "potrf_dag_timeline.txt": "scripts/potrf_dag_timeline.py: phases of k_potrf_dag's chain workgroup per iteration (s_memrealtime stamps), D = 1024 and 4096",
    "panel_w4_ab.txt": "scripts/panel_w4_ab.py: panel product at D >= 2048 with 64-column strips against 16-column strips",
    "scaling_model.json": "scripts/scaling_model.py: per-rank stages measured on one GPU + RCCL world-1 floor + xGMI wire model; N > 1 entries are a model, not measured",
    "pytest_gpu.txt": "python -m pytest tests -m gpu -q (tail)",
}
# the directory is rebuilt from the last pass alone; files listed in KEEP (merged reports written by other scripts of the
# round) are carried over from the published directory
KEEP = ("offgrid.json", "potrf_soak.txt")
for keep in KEEP:
    if os.path.exists(os.path.join(final, keep)):
        shutil.copyfile(os.path.join(final, keep), os.path.join(dst, keep))
for name, note in plain.items():
    publish("summary.json" if name == "rocprof_summary.json" else name, name, note)
ks = first("trace/**/*kernel_stats.csv")
if ks:
    publish(ks, "kernel_stats.csv", "rocprofv3 --kernel-trace --stats of `bench.py --steps 420 --warmup 42 --no-cpu-baseline "
                                    "--no-large-point --no-callpath` (the headline kernel's average duration comes from here)")
ks = first("trace_d4096/**/*kernel_stats.csv")
if ks:
    publish(ks, "d4096_b32_kernel_stats.csv", "rocprofv3 --kernel-trace --stats of `bench.py --D 4096 --B 32 --steps 60 --warmup 12`")
ks = first("c4_update/**/*kernel_stats.csv")
if ks:
    publish(ks, "c4_update_kernel_stats.csv", "rocprofv3 --kernel-trace --stats of scripts/c4_update_bench.py prof")
# no two kernel-stats files may be the same bytes (that is how a mislabelled copy was caught in round 4)
seen = {}
for n, e in manifest["files"].items():
    if "kernel_stats" in n:
        if e["sha256"] in seen:
            sys.exit(f"{n} and {seen[e['sha256']]} are identical")
        seen[e["sha256"]] = n
if "--also-root-traffic" in sys.argv and os.path.exists(os.path.join(dst, "traffic.json")):
    shutil.copyfile(os.path.join(dst, "traffic.json"), os.path.join(ROOT, "profiles", "traffic.json"))   # bench.py's labelled source
json.dump(manifest, open(os.path.join(dst, "MANIFEST.json"), "w"), indent=1)
if os.path.isdir(final):
    shutil.rmtree(final)
os.rename(dst, final)
print(f"published {len(manifest['files'])} files to profiles/{tag}/ (build {build['git_head_short']}, box {box.get('hostname')})")
