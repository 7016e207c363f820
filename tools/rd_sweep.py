#!/usr/bin/env python3
"""RD sweep of BASELINE config 3: 4 synthetic frames with the point counts of
redandblack / loot / longdress / soldier x the 4 (q_g, q_a) pairs of plot.py:31-32, each through
file-mode compress / decompress + GPU metrics; Bjontegaard deltas between the frames' curves.
Seeded random weights in their q-RESPONSIVE variant (synthetic.FILM_GAIN_Q_RESPONSIVE: FiLM heads with gain 1.0, so the rate
follows the quality map; PCC_SWEEP_FILM_GAIN overrides): the rate axis is a real sweep, the distortion axis is that of random
weights — not codec quality.

usage: rd_sweep.py [out.json] [weights.pt]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pcc_amd
from pcc_amd import synthetic as syn
from pcc_amd.harness import evaluate_frame
from pcc_amd.metrics import Bjontegaard_Delta, Bjontegaard_Model

dev = "cuda:0"
FRAMES = {"redandblack~": 247.0, "loot~": 255.0, "longdress~": 261.5, "soldier~": 294.5}   # shell radii -> ~0.76 / 0.81 / 0.86 / 1.09 M
QS = [(0.05, 0.1), (0.1, 0.2), (0.2, 0.4), (0.4, 0.8)]
film_gain = float(os.environ.get("PCC_SWEEP_FILM_GAIN", syn.FILM_GAIN_Q_RESPONSIVE))
model = syn.make_model(seed=0, device=dev, film_gain=film_gain)
if len(sys.argv) > 2:
    model.load_state_dict(torch.load(sys.argv[2], map_location=dev))
model.update()
rows = []
with tempfile.TemporaryDirectory() as td:
    for name, radius in FRAMES.items():
        pts = syn.sphere_shell(grid=1024, radius=radius, half_width=0.5, noise=0.02)
        data = {"src": {"points": torch.from_numpy(pts[None, :, :3]), "colors": torch.from_numpy(pts[None, :, 3:])}}
        for q_g, q_a in QS:
            t0 = time.time()
            row = evaluate_frame("sweep", model, data, q_a, q_g, dev, td)
            row["frame"] = name
            rows.append(row)
            print(f"{name:13s} N={row['n_source']:8d} q=({q_g},{q_a}) bpp {row['bpp']:.3f} D1 {row['sym_p2p_psnr']:.2f} Y {row['sym_y_psnr']:.2f} "
                  f"t_enc {row['t_compress']*1e3:.0f} ms t_dec {row['t_decompress']*1e3:.0f} ms (row {time.time()-t0:.1f} s)", flush=True)
bd = {}
names = list(FRAMES)
ref = [r for r in rows if r["frame"] == names[0]]
for nm in names[1:]:
    cur = [r for r in rows if r["frame"] == nm]
    try:
        m1 = Bjontegaard_Model([r["bpp"] for r in ref], [r["sym_y_psnr"] for r in ref])
        m2 = Bjontegaard_Model([r["bpp"] for r in cur], [r["sym_y_psnr"] for r in cur])
        bd[nm] = {"bd_psnr_y_vs_" + names[0]: float(Bjontegaard_Delta().compute_BD_PSNR(m1, m2))}
    except Exception as e:      # degenerate curves (seeded weights) must not lose the table
        bd[nm] = {"error": repr(e)}
out = {"rows": rows, "bjontegaard": bd,
       "note": ("weights from " + os.path.basename(sys.argv[2]) if len(sys.argv) > 2 else "seeded random weights") +
               f" (FiLM-head gain {film_gain}); synthetic shells sized like the 8iVFB frames",
       "rate_axis_monotone_in_q": {nm: [r["bpp"] for r in rows if r["frame"] == nm] == sorted(r["bpp"] for r in rows if r["frame"] == nm)
                                   for nm in names}}
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "rd_sweep.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
