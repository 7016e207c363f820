#!/usr/bin/env python3
"""Reference-generated fixture for the Bjontegaard half of SURVEY.md §8f rank 3.

Runs ONLY in the build container: imports /root/reference/metrics/bjontegaard.py (numpy / scipy /
matplotlib, all present here) and evaluates it on rate-distortion rows the reference itself recorded in
/root/reference/results/Ours/test.csv (the 4-point q-grid of plot.py:31-32 for longdress frame 1300 and
soldier frame 690).  The output, tests/golden/bjontegaard_ref.json, holds inputs and the reference's
outputs only — it is the one fixture in this repository produced by reference code, and it pins
pcc_amd.metrics.Bjontegaard_Model / Bjontegaard_Delta, not the codec (whose operators live in
MinkowskiEngine / compressai, neither importable here).

    python tests/golden/make_bd_golden.py
"""
import csv
import json
import os
import sys

import matplotlib
matplotlib.use("Agg")
import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
# 1-based line numbers in results/Ours/test.csv: (q_a, q_g) = (0.1,0.05) (0.2,0.1) (0.4,0.2) (0.8,0.4)
LINES = {"longdress_1300": [486, 529, 615, 787], "soldier_690": [927, 970, 1056, 1228]}
METRICS = ["sym_p2p_psnr", "sym_y_psnr", "sym_yuv_psnr"]


def main():
    sys.path.insert(0, os.path.join(REF, "metrics"))
    import bjontegaard as ref                                   # the reference's own module

    with open(os.path.join(REF, "results", "Ours", "test.csv")) as f:
        rows = list(csv.reader(f))
    head = rows[0]
    col = {name: i for i, name in enumerate(head)}
    curves = {}
    for name, lines in LINES.items():
        pts = [rows[ln - 1] for ln in lines]
        seq, frame = name.split("_")
        for p in pts:
            assert p[col["sequence"]] == seq and p[col["frameIdx"]] == frame, (name, p[col["sequence"]], p[col["frameIdx"]])
        curves[name] = {"csv_lines": lines,
                        "q_a": [float(p[col["q_a"]]) for p in pts], "q_g": [float(p[col["q_g"]]) for p in pts],
                        "bpp": [float(p[col["bpp"]]) for p in pts],
                        **{m: [float(p[col[m]]) for p in pts] for m in METRICS}}

    out = {"source": "results/Ours/test.csv rows evaluated by metrics/bjontegaard.py of the reference (build container)",
           "curves": curves, "models": {}, "deltas": []}
    models = {}
    for name, c in curves.items():
        for m in METRICS:
            mod = ref.Bjontegaard_Model(np.array(c["bpp"]), np.array(c[m]))
            models[(name, m)] = mod
            probe_r = [c["bpp"][0], 0.5 * (c["bpp"][1] + c["bpp"][2]), c["bpp"][-1]]
            probe_d = [c[m][0], 0.5 * (c[m][1] + c[m][2]), c[m][-1]]
            xs = mod.get_plot_data()
            out["models"][f"{name}/{m}"] = {
                "parameters_PSNR": [float(v) for v in mod.parameters_PSNR],
                "parameters_Rate": [float(v) for v in mod.parameters_Rate],
                "probe_rates": probe_r, "evaluate": [float(mod.evaluate(r)) for r in probe_r],
                "probe_psnr": probe_d, "evaluate_rate": [float(mod.evaluate_rate(d)) for d in probe_d],
                "plot_x_first_last": [float(xs[2][0]), float(xs[2][-1])],
                "plot_y_first_last": [float(xs[3][0]), float(xs[3][-1])]}
    bd = ref.Bjontegaard_Delta()
    names = list(curves)
    for m in METRICS:
        for a in names:
            for b in names:
                out["deltas"].append({"metric": m, "model1": a, "model2": b,
                                      "BD_PSNR": float(bd.compute_BD_PSNR(models[(a, m)], models[(b, m)])),
                                      "BD_Rate": float(bd.compute_BD_Rate(models[(a, m)], models[(b, m)]))})
    # the reference module's own __main__ example (metrics/bjontegaard.py:82-99)
    r1, r2, d1 = [22.35, 12.93, 8.27, 4.53], [24.35, 13.93, 9.27, 6.53], [71.17, 69.54, 67.62, 65.77]
    m1, m2 = ref.Bjontegaard_Model(r1, d1), ref.Bjontegaard_Model(r2, d1)
    out["module_example"] = {"bitrates1": r1, "bitrates2": r2, "d1": d1,
                             "BD_PSNR_12": float(bd.compute_BD_PSNR(m1, m2)), "BD_Rate_12": float(bd.compute_BD_Rate(m1, m2)),
                             "BD_PSNR_21": float(bd.compute_BD_PSNR(m2, m1)), "BD_Rate_21": float(bd.compute_BD_Rate(m2, m1))}
    path = os.path.join(HERE, "bjontegaard_ref.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
