#!/usr/bin/env python3
"""Development aid: time of ONE small map-convolution launch (HIP events around 300 back-to-back launches), the small-launch
kernel (csrc/conv.hip conv_small_kernel, pipeline shape from PCC_CONV_SMALL_CFG) against the ordinary tile kernels.
  python tools/conv_small_bench.py            # rows 56 .. 8 k, 128 -> 128 and 64 -> 64"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pcc_amd as pcc
from pcc_amd import sparse as sp
torch.set_grad_enabled(False)
dev = "cuda:0"
shell = pcc.synthetic.sphere_shell(128, 55.0, 0.9)[:, :3]         # ~38 k voxels of a surface
rng = np.random.default_rng(0)
center = shell.mean(axis=0)
order = np.argsort(((shell - shell[0]) ** 2).sum(axis=1))          # compact patches of the surface
print("cfg", os.environ.get("PCC_CONV_SMALL_CFG", "default"))
for cin, cout in ((128, 128), (64, 64), (64, 128)):
    layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, bias=True, dimension=3).to(dev)
    for n in [int(v) for v in os.environ.get("PCC_SMALL_BENCH_ROWS", "56,300,1136,2500,4904,8000").split(",")]:
        p = shell[order[:n]]
        c = torch.from_numpy(np.concatenate([np.zeros((n, 1)), p], axis=1).astype(np.int32)).to(dev)
        x = pcc.SparseTensor(torch.randn(n, cin, device=dev), coordinate_map=pcc.CoordMap(c, 1))
        out = []
        for thr in (1 << 20, 0):
            sp.set_conv_small_max(thr)
            for _ in range(20): layer(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300): layer(x)
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 300 * 1e3)
        print(f"{cin:4d}->{cout:4d} rows {n:6d}: small {out[0]:7.1f} us   tiles {out[1]:7.1f} us")
