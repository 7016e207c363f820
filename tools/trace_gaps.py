#!/usr/bin/env python3
"""Step anatomy from a rocprofv3 kernel trace of `PCC_BENCH_MARK=1 python3 bench.py ...`.

bench.py (with PCC_BENCH_MARK=1) launches a marker kernel (torch's logcumsumexp on 3 elements — nothing else in
the path uses it) before compress, between compress and decompress and after decompress.  This tool cuts the
trace at the markers and reports, per phase and averaged over the timed steps: wall time, GPU-busy time (union
of kernel intervals over all streams), idle time, the largest idle gaps with the kernels either side of them,
and busy time per kernel group (overlap-free attribution is not attempted: group times are plain sums).

usage: trace_gaps.py <..._kernel_trace.csv> [--skip-steps W] [--top 12] [--json out.json]
"""
import argparse
import csv
import json
import re
import sys


def group_of(name):
    if "conv_mfma" in name:
        return "conv_mfma"
    if "rocprim" in name or "hipcub" in name:
        return "rocprim/hipcub"
    if "at::native" in name or "at_cuda" in name:
        return "torch elementwise/copy"
    m = re.search(r"pcc::(\w+)", name)
    if m:
        return m.group(1)
    return name.split("(")[0][:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--skip-steps", type=int, default=2, help="steps (marker triplets) to drop at the front: warm-up")
    ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--json", default=None)
    ap.add_argument("--list", default=None, help="write the launch sequence of the last timed step here: phase, start (us from the "
                                                 "phase's start), gap before (us), duration (us), kernel")
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "logcumsumexp" in r[2].lower()]
    # one marker launch can be more than one kernel: merge markers closer than 20 us
    merged = []
    for i in marks:
        if merged and rows[i][0] - rows[merged[-1][-1]][1] < 20000:
            merged[-1].append(i)
        else:
            merged.append([i])
    assert len(merged) >= 3, f"found {len(merged)} markers: run bench.py with PCC_BENCH_MARK=1"
    steps = []
    for j in range(0, len(merged) - 2, 3):
        a, b, c = merged[j], merged[j + 1], merged[j + 2]
        steps.append(((rows[a[-1]][1], rows[b[0]][0]), (rows[b[-1]][1], rows[c[0]][0])))
    steps = steps[args.skip_steps:]
    assert steps, "no timed steps left after --skip-steps"
    mark_idx = {i for m in merged for i in m}
    if args.list:
        with open(args.list, "w") as f:
            for pi, phase in enumerate(("encode", "decode")):
                t0, t1 = steps[-1][pi]
                cur_end = t0
                for i, (s_, e_, n_) in enumerate(rows):
                    if i in mark_idx or s_ < t0 or e_ > t1:
                        continue
                    f.write(f"{phase} {(s_ - t0) / 1e3:10.1f} {max(0, s_ - cur_end) / 1e3:8.1f} {(e_ - s_) / 1e3:8.1f}  {n_[:110]}\n")
                    cur_end = max(cur_end, e_)
                f.write(f"{phase} {(t1 - t0) / 1e3:10.1f} {max(0, t1 - cur_end) / 1e3:8.1f} {0.0:8.1f}  <phase end>\n")
    out = {"steps": len(steps), "phases": {}}
    for pi, phase in enumerate(("encode", "decode")):
        wall = busy = 0.0
        groups, gaps = {}, []
        for st in steps:
            t0, t1 = st[pi]
            ks = [r for i, r in enumerate(rows) if i not in mark_idx and r[0] >= t0 and r[1] <= t1]
            wall += t1 - t0
            cur_end, prev_name = t0, "<phase start>"
            for s, e, n in ks:
                if s > cur_end:
                    gaps.append((s - cur_end, prev_name, n))
                    busy += 0
                if e > cur_end:
                    busy += e - max(s, cur_end)
                    cur_end, prev_name = e, n
                g = group_of(n)
                groups[g] = groups.get(g, 0.0) + (e - s)
            if t1 > cur_end:
                gaps.append((t1 - cur_end, prev_name, "<phase end>"))
        n = len(steps)
        gaps.sort(reverse=True)
        hist = {"<10us": 0, "10-50us": 0, "50-200us": 0, "0.2-1ms": 0, ">1ms": 0}
        hist_t = dict.fromkeys(hist, 0.0)
        for g, _, _ in gaps:
            k = "<10us" if g < 1e4 else "10-50us" if g < 5e4 else "50-200us" if g < 2e5 else "0.2-1ms" if g < 1e6 else ">1ms"
            hist[k] += 1
            hist_t[k] += g
        ph = {"wall_ms": wall / n / 1e6, "gpu_busy_ms": busy / n / 1e6, "gpu_idle_ms": (wall - busy) / n / 1e6,
              "kernel_sum_ms": sum(groups.values()) / n / 1e6,
              "groups_ms": {g: t / n / 1e6 for g, t in sorted(groups.items(), key=lambda kv: -kv[1])},
              "gap_histogram_per_step": {k: {"count": hist[k] / n, "ms": hist_t[k] / n / 1e6} for k in hist},
              "top_gaps": [{"ms": g / 1e6, "after": a[:90], "before": b[:90]} for g, a, b in gaps[: args.top]]}
        out["phases"][phase] = ph
        print(f"== {phase}: wall {ph['wall_ms']:.2f} ms  busy {ph['gpu_busy_ms']:.2f}  idle {ph['gpu_idle_ms']:.2f}  "
              f"(sum of kernel durations {ph['kernel_sum_ms']:.2f})")
        for g, t in list(ph["groups_ms"].items())[:24]:
            print(f"     {t:8.3f} ms  {g}")
        print("   idle gaps per step:", "  ".join(f"{k}: {v['count']:.0f} ({v['ms']:.2f} ms)" for k, v in ph["gap_histogram_per_step"].items()))
        for g in ph["top_gaps"]:
            print(f"     gap {g['ms']:7.3f} ms  after {g['after'][:60]:60s} before {g['before'][:60]}")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
