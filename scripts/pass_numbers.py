#!/usr/bin/env python3
"""Prints the numbers README / DESIGN quote from a published profile directory (profiles/<tag>): bench value at both flag sets,
the headline kernel's profiler average and roofline fraction, Cholesky by size, fit rates per config, the call-path overheads.
usage: pass_numbers.py [tag]"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(ROOT, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r06")
J = lambda n: json.load(open(os.path.join(d, n)))
m = J("MANIFEST.json")
print("build", m["build"].get("git_head_short"), "library", m["build"]["libraries"]["libgsmvi_hip.so"][:16], "on box", m["box"].get("library_sha256_on_box", "")[:16])
b = J("bench.json"); r = b["roofline"]
print(f"value {b['value']:.0f} (driver flags {J('bench_driver_flags.json')['value']:.0f}); bench.json frac {r['frac']:.4f} from {r['avg_kernel_us_rocprof']:.3f} us, events {r['avg_kernel_us_events']:.3f} us")
for f in ("kernel_stats.csv", "d4096_b32_kernel_stats.csv"):
    for row in csv.DictReader(open(os.path.join(d, f))):
        n = row["Name"]
        if n.startswith("k_potrf_dag(") or "k_gsm_cov_sym<32" in n or "cov_sym_p<32" in n or "k_panel_fast_p<2, false" in n:
            us = float(row["AverageNs"]) / 1e3
            extra = ""
            if "k_gsm_cov_sym<32" in n:
                extra = f" frac {17301504.0 / (us * 1e-6) / 8e12:.4f}"
            if "cov_sym_p<32" in n:
                extra = f" frac {270532608.0 / (us * 1e-6) / 8e12:.4f}"
            print(f"  {f}: {n[:36]} x{row['Calls']} avg {us:.3f} us (min {float(row['MinNs']) / 1e3:.2f}){extra}")
print(open(os.path.join(d, "potrf_rate.txt")).read().strip())
fi = b["fit_iterations_per_s"]
print({k: (round(v, 1) if isinstance(v, (int, float)) else v) for k, v in fi.items() if k != "call_path"})
for cfg, dd in J("configs.json")["configs"].items():
    out = {}
    for k, v in dd.items():
        if isinstance(v, dict):
            for kk in ("it_per_s", "it_per_s_marginal", "median_us", "graph_us"):
                if kk in v:
                    out[k + "." + kk] = round(v[kk], 1)
    print(cfg, out)
for cfg, dd in J("callpath.json")["configs"].items():
    for meth, t in dd.items():
        if not isinstance(t, dict) or "native" not in t:
            continue
        print(cfg, meth, "native", round(t["native"]["iteration_us"], 1), {k: (round(v.get("host_fn_us", v.get("idle_gap_us", 0)), 1), round(v["overhead_us"], 1))
                                                                             for k, v in t.items() if isinstance(v, dict) and "overhead_us" in v})
for B, v in J("bigbatch.json")["rows"].items():
    print("B", B, "U_gsm %.1f" % v["U_gsm_dense_us"], "U_bam %.1f" % v["U_bam_dense_us"], {k: round(x, 2) for k, x in v["per_sample_vs_B128"].items()},
          "F_gsm %.0f F_bam %.0f" % (v["F_gsm"]["it_per_s"], v["F_bam"]["it_per_s"]))
