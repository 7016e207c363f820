#!/usr/bin/env python3
"""Delivered shader clock of a kernel class under load, from one `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace
--output-format csv` pass: GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / 8 / launch duration.
MI355X_MICROARCH.md: the quotient reads high on dispatches shorter than ~0.3 ms, so only launches of at least
--min-ms are averaged.

usage: pmc_clock.py <counter_collection.csv> <kernel_trace.csv> <kernel substring> <out.json> [command line of the pass]
"""
import csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cc, kt, sub, out = sys.argv[1:5]
min_ms = 0.3
dur = {}
for r in csv.DictReader(open(kt)):
    if sub in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
cyc = {}
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
        cyc[r["Dispatch_Id"]] = cyc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
rows = [(dur[d], cyc[d] / 8.0) for d in cyc if dur[d] >= min_ms * 1e-3]
assert rows, "no launches of at least %.1f ms" % min_ms
t = sum(a for a, _ in rows)
c = sum(b for _, b in rows)
try:
    if os.environ.get("PCC_PROFILE_COMMIT"):          # the commit the GPU passes ran at, when HEAD has moved on since
        raise KeyError
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except KeyError:
    commit = os.environ["PCC_PROFILE_COMMIT"]
except Exception:
    commit = None
from bench import kernel_source_sha256
key = sub.replace(" ", "")
res = {"what": "delivered shader clock during the dominant convolution class: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / launch "
               "duration, launches >= %.1f ms, one rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace pass" % min_ms,
       "commit": commit, "kernel_source_sha256": kernel_source_sha256(), "command": sys.argv[5] if len(sys.argv) > 5 else None,
       "bench_config2_frame": {key: {"launches": len(rows), "total_ms": t * 1e3, "Mcycles": c / 1e6, "mean_clock_GHz": c / t / 1e9}}}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["bench_config2_frame"]))
