#!/usr/bin/env python3
"""Development aid: wall-time breakdown of one compress+decompress with a device sync around every
wrapped operator (serialises everything, so totals exceed bench.py's; use for shares only)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pcc_amd
from pcc_amd import sparse as sp, entropy as en

T = collections.OrderedDict()
def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[label] = T.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, w)

wrap(sp.CoordMap, "_unique", "coord sets (stride map / children)")
wrap(sp.CoordMap, "kernel_map", "kernel_map")
wrap(sp.CoordMap, "ordered_kernel_map", "ordered_kernel_map (incl. kernel_map)")
wrap(sp.CoordMap, "table", "hash_build")
wrap(sp.CoordMap, "sort_permutation", "sort_permutation")
wrap(sp, "conv_forward", "conv_forward (incl. maps)")
wrap(sp, "topk_mask", "topk_mask")
wrap(sp, "compact_rows", "compact_rows")
wrap(en, "_rans_encode", "rans_encode (host)")
wrap(en, "_rans_decode", "rans_decode (host)")
wrap(en.EntropyBottleneck, "compress_features", "EB.compress_features (incl. rans)")
wrap(en.GaussianConditional, "compress_features", "GC.compress_features (incl. rans)")
wrap(en.EntropyBottleneck, "decompress_features", "EB.decompress_features (incl. rans)")
wrap(en.GaussianConditional, "decompress_features", "GC.decompress_features (incl. rans)")

dev = "cuda:0"
syn = pcc_amd.synthetic
model = syn.make_model(0, dev); model.update()
cfg = syn.CONFIG2 if len(sys.argv) < 2 else dict(grid=256, radius=100.0, half_width=0.5)
pts = syn.sphere_shell(**cfg)
qc, qf = syn.uniform_qmap(pts[:, :3])
x = torch.from_numpy(pts).to(dev)
for it in range(2):
    T.clear()
    Q = pcc_amd.SparseTensor(coordinates=torch.from_numpy(qc).to(dev), features=torch.from_numpy(qf).to(dev), device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    strings, shape, k, coords = model.compress(x, Q)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    enc = dict(T); T.clear()
    rec = model.decompress(coordinates=coords, strings=strings, shape=shape, k=k)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    dec = dict(T)
print(f"encode {1e3*(t1-t0):.1f} ms")
for k_, v in enc.items(): print(f"   {k_:50s} {1e3*v:8.2f} ms")
print(f"decode {1e3*(t2-t1):.1f} ms")
for k_, v in dec.items(): print(f"   {k_:50s} {1e3*v:8.2f} ms")
