#!/usr/bin/env python3
"""Run HERE (build container), right before a profile collection: writes gsm-vi_amd/BUILD_INFO.json -- the git commit the
library was built from, whether the tree was dirty, and the sha256 of every shipped library.  The file travels to the GPU box
with the snapshot (.git does not); scripts/collect_profiles.sh copies it beside what it measures and checks that the library it
loaded has the recorded hash; scripts/publish_profiles.py puts it into profiles/<tag>/MANIFEST.json."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def git(*a):
    return subprocess.run(["git", "-C", ROOT, *a], capture_output=True, text=True).stdout.strip()


csrc = os.path.join(ROOT, "gsm-vi_amd", "csrc")
stale = subprocess.run(["make", "-C", csrc, "-q"], capture_output=True).returncode != 0
if stale:
    sys.exit("the library is older than its sources: run `make -C gsm-vi_amd/csrc` first")
info = {"git_head": git("rev-parse", "HEAD"), "git_head_short": git("rev-parse", "--short", "HEAD"),
        "git_dirty_files": [ln for ln in git("status", "--porcelain").splitlines() if ln.strip()],
        "stamped_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
        "libraries": {n: sha256(os.path.join(ROOT, "gsm-vi_amd", n)) for n in ("libgsmvi_hip.so", "libgsmvi_hip_debug.so")
                      if os.path.exists(os.path.join(ROOT, "gsm-vi_amd", n))},
        "csrc_sha256": {n: sha256(os.path.join(csrc, n)) for n in sorted(os.listdir(csrc))
                        if n.endswith((".hip", ".h", ".map")) or n == "Makefile"}}
json.dump(info, open(os.path.join(ROOT, "gsm-vi_amd", "BUILD_INFO.json"), "w"), indent=1)
print("stamped", info["git_head_short"], "dirty:" if info["git_dirty_files"] else "clean", *info["git_dirty_files"][:5])
