#!/usr/bin/env python3
"""Development aid: forward / backward-data / backward-weight rates of the sparse convolution on a
surface-like cloud (1024^3 shell, the config-2 geometry at stride 1 or 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pcc_amd
from pcc_amd import synthetic as syn

dev = "cuda:0"
pts = syn.sphere_shell(**syn.CONFIG2)
c = torch.from_numpy(np.concatenate([np.zeros((pts.shape[0], 1)), pts[:, :3]], 1).astype(np.int32)).to(dev)
m1 = pcc_amd.CoordMap(c, 1, nbatch=1)
maps = {"stride1": m1, "stride2": m1.down()}
for name, m in maps.items():
    for cin, cout in [(64, 64), (128, 128)]:
        layer = pcc_amd.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, bias=True, dimension=3).to(dev)
        x = torch.randn(m.n, cin, device=dev, requires_grad=True)
        g = torch.randn(m.n, cout, device=dev)
        nbr, order, gmask, pairs = m.ordered_kernel_map(m, 3)
        flops = 2.0 * float(pairs) * cin * cout
        def timed(fn, reps=5):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        with torch.no_grad():
            t_f = timed(lambda: layer(pcc_amd.SparseTensor(x.detach(), coordinate_map=m)))
        out = layer(pcc_amd.SparseTensor(x, coordinate_map=m)).F
        layer.kernel.requires_grad_(False); layer.bias.requires_grad_(False)
        out_x = layer(pcc_amd.SparseTensor(x, coordinate_map=m)).F
        t_dx = timed(lambda: torch.autograd.grad(out_x, x, g, retain_graph=True))
        layer.kernel.requires_grad_(True)
        out_w = layer(pcc_amd.SparseTensor(x.detach(), coordinate_map=m)).F
        t_dw = timed(lambda: torch.autograd.grad(out_w, layer.kernel, g, retain_graph=True))
        print(f"{name} rows {m.n:8d} {cin:3d}->{cout:<3d} nbrs/row {float(pairs)/m.n:4.1f}  fwd {t_f:7.3f} ms {flops/t_f/1e9:6.1f} TF/s | "
              f"dX {t_dx:7.3f} ms {flops/t_dx/1e9:6.1f} TF/s | dW {t_dw:7.3f} ms {flops/t_dw/1e9:6.1f} TF/s")
