#!/usr/bin/env python3
"""Merges gpurun_out/offgrid_before.json (round-4 library) and gpurun_out/offgrid_after.json (this tree), both written by
scripts/offgrid_bench.py, into ONE document (argv[1], default profiles/r05/offgrid.json) with a summary table: every off-grid
shape next to its padded grid neighbour, update (replayed-graph microseconds) and fit (marginal iterations / s), the ratio to
the neighbour, and the pre-change cliff."""
import json
import sys

out = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05/offgrid.json"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
before = json.load(open(f"{src}/offgrid_before.json"))
after = json.load(open(f"{src}/offgrid_after.json"))


def u(e, kind, mode="direct"):
    try:
        return e["U"][kind][mode]["graph_us"]
    except (KeyError, TypeError):
        return None


def f(e, kind):
    try:
        return e["F"][kind]["it_per_s_marginal"]
    except (KeyError, TypeError):
        return None


rows, worst = [], 0.0
for key, e in after["shapes"].items():
    if e["role"] != "offgrid":
        continue
    nkey = "%dx%d" % tuple(e["neighbour"])
    nb, bf = after["shapes"][nkey], before["shapes"].get(key, {})
    row = {"shape": key, "neighbour": nkey, "U_graph_us": {}, "F_it_per_s": {}}
    for kind in ("gsm_dense", "gsm_factor", "bam_dense", "bam_factor"):
        a, n, b = u(e, kind), u(nb, kind), u(bf, kind)
        if a is None or n is None:
            continue
        row["U_graph_us"][kind] = {"before": b, "after": a, "neighbour": n, "after_over_neighbour": a / n,
                                   "before_over_neighbour": (b / n) if b else None}
        worst = max(worst, a / n)
    for kind in ("gsm_auto", "gsm_dense", "bam_dense", "bam_factor"):
        a, n, b = f(e, kind), f(nb, kind), f(bf, kind)
        if a is None or n is None:
            continue
        row["F_it_per_s"][kind] = {"before": b, "after": a, "neighbour": n, "neighbour_over_after": n / a,
                                   "neighbour_over_before": (n / b) if b else None}
        worst = max(worst, n / a)
    rows.append(row)
doc = {"what": "off-grid (D, B) next to the padded grid neighbour (D up to 64, B up to 8/16/32/64/128); update = replayed hipGraph "
               "of back-to-back engine calls on plain contiguous tensors, fit = marginal iterations/s with the built-in Gaussian "
               "score; 'before' = the round-4 library (guarded kernels off the grid), 'after' = this tree",
       "worst_ratio_to_neighbour_after": worst, "summary": rows, "before": before, "after": after}
json.dump(doc, open(out, "w"), indent=1)
for r in rows:
    print(r["shape"], "vs", r["neighbour"])
    for k, v in r["U_graph_us"].items():
        print("   U %-11s before %7.1f  after %7.1f  neighbour %7.1f us   after/neighbour %.2f" %
              (k, v["before"] or float("nan"), v["after"], v["neighbour"], v["after_over_neighbour"]))
    for k, v in r["F_it_per_s"].items():
        print("   F %-11s before %7.0f  after %7.0f  neighbour %7.0f it/s neighbour/after %.2f" %
              (k, v["before"] or float("nan"), v["after"], v["neighbour"], v["neighbour_over_after"]))
print("worst ratio to the neighbour:", round(worst, 3))
