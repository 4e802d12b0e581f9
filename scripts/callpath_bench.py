#!/usr/bin/env python3
"""The drop-in CALL PATH (round-4 verdict, item 4) at BASELINE configs c2 and c3: fit-iteration rates with a host numpy score
(examples/example_gsm_numpy.py:24-29) and a torch-autograd score (examples/example_gsm.py:34-35) beside the built-in device
score; both fit methods.  The measurement itself is bench.callpath_rates (bench.py carries the c3 / c2 "auto" rows in its JSON
line).  Usage: callpath_bench.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/callpath.json"
res = {"device": torch.cuda.get_device_name(0), "host_cpus": os.cpu_count(), "configs": {}}
for name, D, B in (("c2", 256, 8), ("c3", 1024, 32)):
    res["configs"][name] = bench.callpath_rates(D, B, methods=("auto", "dense"), loop_variant=(name == "c2"))
    print(name, json.dumps(res["configs"][name]), flush=True)
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
