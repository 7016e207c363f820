#!/usr/bin/env python3
"""Matrix-pipe utilisation of the convolution kernels from counters, not from a model.

Input: the MFMA_BUSY / MFMA_OPS / WAVE passes of tools/gpu_pmc_job.sh (separate `rocprofv3 --pmc ... --kernel-trace
--output-format csv` runs of the bench command).  Per kernel template (the top templates by time):
  mfma_busy_frac   = sum over dispatches of SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / XCDs x SIMDs)
                     — rocprofiler-sdk's MfmaUtil quotient (counter_defs.yaml): the share of SIMD-cycles, AT THE DELIVERED
                     CLOCK, in which the matrix pipe is busy.  rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs.
  mfma_mops_f32    = SQ_INSTS_VALU_MFMA_MOPS_F32 per dispatch (512 FLOP each on this counter's scale) and the FLOP count
                     they imply, beside the issued-row model of bench.py;
  wave-time split  = SQ_WAIT_ANY (parked at a wait / barrier), SQ_WAIT_INST_ANY (issue stall), SQ_ACTIVE_INST_ANY over
                     SQ_WAVE_CYCLES.
usage: pmc_mfma.py <dir with <tag>_pmc_*/> <tag> <out.json> [command]"""
import csv, glob, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
base, tag, out = sys.argv[1:4]
SIMDS = 256 * 4
XCDS = 8


def load(name):
    d = os.path.join(base, f"{tag}_pmc_{name}")
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert cc and kt, (d, cc, kt)
    dur, kname = {}, {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        kname[r["Dispatch_Id"]] = r["Kernel_Name"]
    vals = {}
    for r in csv.DictReader(open(cc[0])):
        vals.setdefault(r["Dispatch_Id"], {}).setdefault(r["Counter_Name"], 0.0)
        vals[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    return dur, kname, vals


def template(name):
    m = re.search(r"(conv_[a-z_0-9]+<[^>]*>)", name) or re.search(r"(conv_[a-z_0-9]+)\(", name)
    return m.group(1).replace(" ", "") if m else None


def per_template(name, min_ms=0.0):
    dur, kname, vals = load(name)
    agg = {}
    for d, v in vals.items():
        t = template(kname.get(d, ""))
        if t is None or dur.get(d, 0.0) < min_ms * 1e-3:
            continue
        a = agg.setdefault(t, {"launches": 0, "seconds": 0.0})
        a["launches"] += 1
        a["seconds"] += dur[d]
        for k, x in v.items():
            a[k] = a.get(k, 0.0) + x
    return agg

def maybe(name):
    try:
        return per_template(name)
    except AssertionError:
        return {}

busy = per_template("MFMA_BUSY")
ops = maybe("MFMA_OPS")
wave = maybe("WAVE")
top = sorted(busy, key=lambda t: -busy[t]["seconds"])[:6]
res = {}
for t in top:
    b = busy[t]
    gui = b["GRBM_GUI_ACTIVE"] / XCDS
    e = {"launches": b["launches"], "total_ms": b["seconds"] * 1e3,
         "mfma_busy_frac": b["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * SIMDS),
         "delivered_clock_ghz": gui / b["seconds"] / 1e9,
         "mfma_busy_cycles_per_simd_per_second_of_launch": b["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / b["seconds"]}
    # the same busy cycles against the 2.4 GHz the roofline peak assumes
    e["mfma_busy_frac_of_2p4ghz"] = b["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / (b["seconds"] * 2.4e9)
    if t in ops:
        o = ops[t]
        e["mfma_mops_f32_per_launch"] = o["SQ_INSTS_VALU_MFMA_MOPS_F32"] / o["launches"]
        e["sq_busy_cycles_per_launch"] = o.get("SQ_BUSY_CYCLES", 0.0) / o["launches"]
    if t in wave:
        w = wave[t]
        wc = max(w.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        e["wave_time_split"] = {"wait_any": w.get("SQ_WAIT_ANY", 0.0) / wc, "wait_inst_any": w.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                                "active_inst_any": w.get("SQ_ACTIVE_INST_ANY", 0.0) / wc}
    res[t] = e
try:
    commit = os.environ.get("PCC_PROFILE_COMMIT") or subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = None
from bench import kernel_source_sha256
doc = {"what": "matrix-pipe utilisation of the convolution kernel templates of one config-2 bench run from rocprofv3 --pmc passes "
               "(SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; SQ_INSTS_VALU_MFMA_MOPS_F32 + SQ_BUSY_CYCLES; SQ wave-time counters), "
               "one pass per counter group, --kernel-trace beside them",
       "commit": commit, "kernel_source_sha256": kernel_source_sha256(), "command": sys.argv[4] if len(sys.argv) > 4 else None,
       "simds": SIMDS, "kernels": res}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc, indent=1))
