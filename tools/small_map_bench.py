#!/usr/bin/env python3
"""Development aid: time of building one kernel map + execution order (stride-1, kernel 3), the one-launch small-map kernel
(csrc/select.hip) against the separate launches, HIP events around 200 builds.   python tools/small_map_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pcc_amd as pcc
from pcc_amd import sparse as sp
dev = "cuda:0"
shell = pcc.synthetic.sphere_shell(96, 42.0, 0.9)[:, :3]
order = np.argsort(((shell - shell[0]) ** 2).sum(axis=1))
for n in (56, 300, 512, 1136, 4904, 12000):
    c = torch.from_numpy(np.concatenate([np.zeros((n, 1)), shell[order[:n]]], axis=1).astype(np.int32)).to(dev)
    m = pcc.CoordMap(c, 1)
    m.table()
    out = []
    for cap in (1 << 20, 0):
        sp.set_small_map_max(cap)
        def build():
            for k in [k for k in m._cache if k[0] in ("kmap", "okmap")]:
                del m._cache[k]
            m.ordered_kernel_map(m, 3)
        for _ in range(10): build()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): build()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 200 * 1e3)
    print(f"rows {n:5d}: one launch {out[0]:7.1f} us   separate launches {out[1]:7.1f} us")
