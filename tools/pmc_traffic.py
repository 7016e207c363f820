#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes) into the per-launch HBM traffic of the dominant kernel.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> <out.json> [command line of the passes]

Run in the build container after gpurun has merged the CSVs back (git is available there): the output records the
commit and the hash of the kernel sources, so that bench.py can say whether the replayed figure is stale.

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE is in KB and reports half the bytes of
wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is in KB and exact for streaming stores.
"""
import csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def per_dispatch(path, counter, substr):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and substr in r["Kernel_Name"]:
            out[r["Dispatch_Id"]] = out.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return out

fetch = per_dispatch(sys.argv[1], "FETCH_SIZE", sys.argv[3])
write = per_dispatch(sys.argv[2], "WRITE_SIZE", sys.argv[3])
n = min(len(fetch), len(write))
assert n > 0, "no matching dispatches"
fb = sum(fetch.values()) / len(fetch) * 1024 * 2
wb = sum(write.values()) / len(write) * 1024
try:
    if os.environ.get("PCC_PROFILE_COMMIT"):          # the commit the GPU passes ran at, when HEAD has moved on since
        raise KeyError
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "learned-compression-of-point-cloud-geometry-and-attributes_amd/csrc"], text=True).strip())
    commit += "+uncommitted csrc changes" if dirty else ""
except KeyError:
    commit = os.environ["PCC_PROFILE_COMMIT"]
except Exception:
    commit = None
from bench import kernel_source_sha256
json.dump({"kernel": sys.argv[3], "commit": commit, "kernel_source_sha256": kernel_source_sha256(),
           "command": sys.argv[5] if len(sys.argv) > 5 else None, "launches_fetch_pass": len(fetch), "launches_write_pass": len(write),
           "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb,
           "note": "FETCH_SIZE x2 (gfx950 wide-read correction), KB -> bytes; averages over all launches of the kernel in one bench step"},
          open(sys.argv[4], "w"), indent=1)
print(open(sys.argv[4]).read())
