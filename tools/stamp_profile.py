#!/usr/bin/env python3
"""Copy a bench.py result line (gpurun_out/*.json) into profiles/ with the commit and kernel-source hash it was taken at.

usage: stamp_profile.py <gpurun_out/line.json> <profiles/rNN_name.json> [commit]
Run in the build container right after the GPU call, before the sources move on (bench.py replays cpu_baseline of the newest
profiles/rNN_bench_config2.json at N > 1 and names this commit)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha256  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
text = open(src).read().strip().splitlines()
line = next(ln for ln in reversed(text) if ln.lstrip().startswith("{"))
j = json.loads(line)
j["commit"] = sys.argv[3] if len(sys.argv) > 3 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
j["kernel_source_sha256"] = kernel_source_sha256()
json.dump(j, open(dst, "w"), indent=1)
print("wrote", dst, "value", j.get("value"), j.get("unit"), "ms_per_step", j.get("ms_per_step"))
