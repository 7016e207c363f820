#!/usr/bin/env python3
"""Development aid: the thin convolutions (cin <= 16: q-map branches, input layer) in isolation, HIP events, median of 7."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pcc_amd
from pcc_amd import synthetic as syn

dev = "cuda:0"
pts = syn.sphere_shell(**syn.CONFIG2)[:, :3].astype(np.int32)
c1 = torch.from_numpy(np.concatenate([np.zeros((pts.shape[0], 1), np.int32), pts], axis=1)).to(dev)
m1 = pcc_amd.CoordMap(c1, 1, nbatch=1)
m2 = m1.down()
for name, m, cin, cout in (("stride2", m2, 2, 128), ("stride1", m1, 4, 64), ("stride1", m1, 2, 2), ("stride2", m2, 2, 16)):
    layer = pcc_amd.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, bias=True, dimension=3).to(dev)
    x = pcc_amd.SparseTensor(torch.randn(m.n, cin, device=dev), coordinate_map=m)
    with torch.no_grad():
        layer(x)
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); layer(x); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    t = statistics.median(ts)
    out_mb = m.n * cout * 4 / 1e6
    print(f"{name} rows {m.n:7d} {cin}->{cout:<3d}: {t * 1e3:7.1f} us   output {out_mb:6.1f} MB -> {out_mb / t / 1e3:5.2f} TB/s of writes")
