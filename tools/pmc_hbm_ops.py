#!/usr/bin/env python3
"""HBM traffic of the non-matrix (HBM-bound) kernels of a bench step, per operator class, from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only beside them — MI355X_MICROARCH.md) of the bench command.

usage: pmc_hbm_ops.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps in the run (warm-up + timed)> <out.json> [command]

Output (profiles/rNN_hbm_kernel_traffic.json, replayed by bench.py's `roofline_hbm` record): per operator class the HBM bytes
per step and the kernels behind them.  Units and corrections as the guide prescribes for gfx950: both counters in KB;
FETCH_SIZE doubled (128-byte requests tallied at 64 B) — calibrated there for wide coalesced reads; the 4- and 8-byte
probes and scattered row reads of these kernels are outside that calibration, so their absolute figures are upper-bound
estimates (ratios between variants of one kernel are unaffected).  Infinity-Cache hits are counted, not excluded.
"""
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

OPERATOR_OF = [            # first match wins; names as pcc_amd/sparse.py logs them (bench.py roofline_hbm)
    (r"kernel_map27_kernel|pcc::kernel_map_kernel", "kernel_map"),
    (r"small_map_kernel", "small_map+order"),
    (r"mask_bit_counts_kernel|order_keys32|order_keys64|group_masks_kernel|order_small_kernel|radix_\w+_kernel<unsigned int|radix_rowscan", "execution_order"),
    (r"coords_to_keys|radix_\w+_kernel<unsigned long|radix_sort_small_kernel<unsigned long", "canonical_sort"),
    (r"unique_insert<pcc::GenChildren|unique_finalize<pcc::GenChildren", "unique_children"),
    (r"unique_insert<pcc::GenStride|unique_finalize<pcc::GenStride", "unique_stride_map"),
    (r"unique_flag|unique_small_kernel|scan_block_sums|scan_of_block_sums|scan_apply|table_clear", "unique / prune (shared: flags, scans, table clear)"),
    (r"build_insert|build_count_dups", "hash_build"),
    (r"lookup_kernel", "hash_lookup"),
    (r"topk_", "top_k"),
    (r"compact_", "prune"),
    (r"gather_rows_kernel|scatter_rows_kernel", "gather_rows"),
    (r"im2col_thin_kernel", "im2col_thin"),
    (r"gather_sum", "gather_sum"),
    (r"eb_|gc_", "entropy model (quantise / index / dequantise)"),
    (r"conv_thin_kernel", "conv_thin (HBM-bound convolutions, cin <= 16)"),
]


def operator_of(kernel):
    for pat, op in OPERATOR_OF:
        if re.search(pat, kernel):
            return op
    return None


def per_kernel(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        d = out.setdefault(name, {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return {k: (len(v), sum(v.values())) for k, v in out.items()}


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("pcc::", "")[:70]


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    steps = int(sys.argv[3])
    ops = {}
    for name in sorted(set(fetch) | set(write)):
        op = operator_of(name)
        if op is None:
            continue
        nf, f = fetch.get(name, (0, 0.0))
        nw, w = write.get(name, (0, 0.0))
        o = ops.setdefault(op, {"hbm_bytes_per_step": 0.0, "fetch_bytes_per_step": 0.0, "write_bytes_per_step": 0.0, "kernels": {}})
        fb, wb = f * 1024 * 2 / steps, w * 1024 / steps
        o["fetch_bytes_per_step"] += fb
        o["write_bytes_per_step"] += wb
        o["hbm_bytes_per_step"] += fb + wb
        o["kernels"][short(name)] = {"launches_per_step": max(nf, nw) / steps, "fetch_bytes_per_step": fb, "write_bytes_per_step": wb}
    try:
        commit = os.environ.get("PCC_PROFILE_COMMIT") or subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = None
    from bench import kernel_source_sha256
    out = {"commit": commit, "kernel_source_sha256": kernel_source_sha256(), "command": sys.argv[5] if len(sys.argv) > 5 else None,
           "steps_in_run": steps, "operators": ops,
           "note": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B; calibrated for wide coalesced reads — narrow random accesses "
                   "are outside the calibration), WRITE_SIZE as read, KB -> bytes; sums over all launches of a kernel / steps in the run"}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    for op, o in sorted(ops.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"]):
        print(f"{op:55s} {o['hbm_bytes_per_step'] / 1e6:10.1f} MB/step  (fetch {o['fetch_bytes_per_step'] / 1e6:9.1f}  write {o['write_bytes_per_step'] / 1e6:9.1f})")


if __name__ == "__main__":
    main()
