#!/usr/bin/env python3
"""Development aid: fp32 vs bf16-input convolution on the config-2 geometry (850 k rows) and on its 5.2 M
k3-children (the decoder's largest candidate set)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pcc_amd
from pcc_amd import _lib, synthetic as syn
from pcc_amd._lib import check, ptr
L = pcc_amd.lib(); dev = "cuda:0"
pts = syn.sphere_shell(**syn.CONFIG2)
c = torch.from_numpy(np.concatenate([np.zeros((pts.shape[0], 1)), pts[:, :3]], 1).astype(np.int32)).to(dev)
m1 = pcc_amd.CoordMap(c, 1, nbatch=1)
maps = {"surface 850k": m1, "children of stride-2 (3.4M)": m1.down().up(3)}
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, m in maps.items():
    nbr, order, gmask, pairs = m.ordered_kernel_map(m, 3)
    for cin, cout in [(64, 64), (128, 128)]:
        n = m.n
        F = torch.randn(n, cin, device=dev); W = torch.randn(27, cin, cout, device=dev) * 0.05
        Wp = torch.empty(L.pcc_conv_packed_elems(27, cin, cout), device=dev); check(L.pcc_conv_pack_weights(ptr(W), 27, cin, cout, ptr(Wp), _lib.stream()))
        Wb = torch.empty(L.pcc_conv_packed_elems_bf16(27, cin, cout), dtype=torch.bfloat16, device=dev); check(L.pcc_conv_pack_weights_bf16(ptr(W), 27, cin, cout, ptr(Wb), _lib.stream()))
        Fb = F.to(torch.bfloat16); out = torch.empty(n, cout, device=dev)
        t32 = timed(lambda: check(L.pcc_conv_fwd(ptr(F), n, cin, ptr(W), ptr(Wp), None, ptr(nbr), ptr(order), ptr(gmask), 27, ptr(out), n, cout, 0, None, None, _lib.stream())))
        t16 = timed(lambda: check(L.pcc_conv_fwd_bf16(ptr(Fb), n, cin, ptr(Wb), None, ptr(nbr), ptr(order), ptr(gmask), 27, ptr(out), n, cout, 0, None, None, _lib.stream())))
        tcast = timed(lambda: F.to(torch.bfloat16))
        fl = 2.0 * float(pairs) * cin * cout
        gb = float(pairs) * cin * 2 / 1e9
        print(f"{name:28s} rows {n:8d} {cin:3d}->{cout:<3d} fp32 {t32:7.3f} ms {fl/t32/1e9:6.1f} TF/s | bf16 {t16:7.3f} ms {fl/t16/1e9:6.1f} TF/s  gather {gb/t16*1e3:5.2f} TB/s | cast {tcast:.3f} ms")
