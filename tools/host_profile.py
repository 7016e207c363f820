#!/usr/bin/env python3
"""Development aid: where the HOST time of a small frame goes (the ~10 ms per-frame floor that bounds small frames and the
per-rank time of the spatial-block mode): cProfile of compress + decompress on the config-1 sphere (or `mid`), after warm-up.
  python tools/host_profile.py [config1|mid] [frames]"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pcc_amd
from pcc_amd import synthetic as syn
dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "config1"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = syn.CONFIG1 if which == "config1" else dict(grid=256, radius=100.0, half_width=0.5)
model = syn.make_model(0, dev); model.update()
pts = syn.sphere_shell(**cfg)
qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)
x, qcd, qfd = torch.from_numpy(pts).to(dev), torch.from_numpy(qc).to(dev), torch.from_numpy(qf).to(dev)
def frame():
    Q = pcc_amd.SparseTensor(coordinates=qcd, features=qfd, device=dev)
    s, shape, k, c = model.compress(x, Q)
    return model.decompress(coordinates=c, strings=s, shape=shape, k=k)
for _ in range(3): frame()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(frames): frame()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(22)
