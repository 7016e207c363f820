#!/usr/bin/env python3
"""Development aid: time PointCloudMetric on BASELINE config 2 (source frame vs. the decoded frame)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pcc_amd
from pcc_amd import synthetic as syn
from pcc_amd.metrics import PointCloudMetric

dev = "cuda:0"
model = syn.make_model(seed=0, device=dev); model.update()
pts = syn.sphere_shell(**syn.CONFIG2, noise=0.02)
qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)
x = torch.from_numpy(pts).to(dev)
Q = pcc_amd.SparseTensor(coordinates=torch.from_numpy(qc).to(dev), features=torch.from_numpy(qf).to(dev), device=dev)
strings, shape, k, coords = model.compress(x, Q)
rec = model.decompress(coordinates=coords, strings=strings, shape=shape, k=k)
for name, other in (("decoded (seeded weights: geometry far from the source)", rec),
                    ("source jittered by +-1 voxel", torch.cat([torch.round(x[:, :3] + torch.randint(-1, 2, x[:, :3].shape, device=dev)), x[:, 3:]], 1))):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res, _ = PointCloudMetric(x, other, resolution=1023).compute_pointcloud_metrics(drop_duplicates=True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{name}: {1e3 * (t1 - t0):.1f} ms  D1 {res['sym_psnr_mse']:.2f} dB  Y {res['sym_y_psnr']:.2f} dB  N={x.shape[0]} vs {other.shape[0]}")
