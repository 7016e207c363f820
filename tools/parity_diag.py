#!/usr/bin/env python3
"""Development aid: HIP codec vs CPU oracle on one synthetic frame — are the streams byte-identical, where do the decoded
voxel sets part, how far apart are the metrics.   python tools/parity_diag.py GRID RADIUS Q_G Q_A [noise]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.set_num_threads(8)
import pcc_amd
from oracle.codec import Codec, count_bits
from oracle.metrics import pc_metrics
from _parity import voxel_flips
grid, radius, qg, qa = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
noise = float(sys.argv[5]) if len(sys.argv) > 5 else 0.02
dev = "cuda:0"
syn = pcc_amd.synthetic
model = syn.make_model(0, dev); model.update()
sd = {k: v.detach().cpu() for k, v in syn.make_model(0, "cpu").state_dict().items()}
codec = Codec(sd); codec.update()
pts = syn.sphere_shell(grid=grid, radius=radius, half_width=0.5, noise=noise)
N = pts.shape[0]
qc, qf = syn.uniform_qmap(pts[:, :3], qg, qa)
x = torch.from_numpy(pts).to(dev)
Q = pcc_amd.SparseTensor(coordinates=torch.from_numpy(qc).to(dev), features=torch.from_numpy(qf).to(dev), device=dev)
s, shape, k, coords = model.compress(x, Q)
os_, oshape, ok, ocoords = codec.compress(pts, qc, qf)
print("N", N, "k", k, ok, "y stream equal", s[0] == os_[0], "z stream equal", s[1] == os_[1], "bytes", len(s[0][0]), len(os_[0][0]))
rec = model.decompress(coordinates=coords, strings=s, shape=shape, k=k).cpu().numpy()
orec = codec.decompress(ocoords, os_, oshape, ok)
print("flips own streams", voxel_flips(rec, orec))
m, om = pc_metrics(pts, rec, resolution=grid - 1), pc_metrics(pts, orec, resolution=grid - 1)
print("D1", m["sym_psnr_mse"], om["sym_psnr_mse"], "Y", m["sym_y_psnr"], om["sym_y_psnr"])
# the oracle's stream through the HIP decoder (may desynchronise if h_s differs in a scale index)
try:
    rec2 = model.decompress(coordinates=torch.from_numpy(ocoords).to(dev), strings=os_, shape=oshape, k=ok).cpu().numpy()
    print("flips HIP decoder on oracle stream vs oracle decoder", voxel_flips(rec2, orec))
    m2 = pc_metrics(pts, rec2, resolution=grid - 1)
    print("D1", m2["sym_psnr_mse"], "Y", m2["sym_y_psnr"])
except Exception as e:
    print("cross decode failed:", repr(e)[:200])
