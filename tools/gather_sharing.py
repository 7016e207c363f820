#!/usr/bin/env python3
"""Development aid: how many DISTINCT input rows does a tile of the MFMA convolution gather, per offset (today),
per (dy, dz) pair (the three dx offsets sharing one LDS image) and per tile (all 27 offsets)?  Candidate set of the
last up block of the config-2 frame (stride 1) and its parent level, in the execution order the kernel uses."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pcc_amd
from pcc_amd import synthetic as syn

dev = "cuda:0"
pts = syn.sphere_shell(**syn.CONFIG2)[:, :3]
c = torch.from_numpy(np.concatenate([np.zeros((pts.shape[0], 1)), pts], 1).astype(np.int32)).to(dev)
m1 = pcc_amd.CoordMap(c, 1, nbatch=1)
for name, m in (("surface, stride 1", m1), ("candidates of the stride-2 surface (k3 children)", m1.down().up(3))):
    nbr, order, gmask, pairs = m.position_ordered_table(m, 3)        # rows in execution order
    n = m.n
    for BM in (64, 128):
        T = n // BM
        t = nbr[: T * BM].reshape(T, BM, 27).long()
        present = int((t >= 0).sum())

        def distinct(x):                      # x: [T, M] with -1 = absent -> number of distinct non-negative values per tile
            s, _ = torch.sort(x, dim=1)
            new = torch.ones_like(s, dtype=torch.bool)
            new[:, 1:] = s[:, 1:] != s[:, :-1]
            return int((new & (s >= 0)).sum())

        per_pair = sum(distinct(t[:, :, 3 * g: 3 * g + 3].reshape(T, -1)) for g in range(9))
        per_tile = distinct(t.reshape(T, -1))
        print(f"{name}: rows {n}, tile {BM}: gathers today {present / 1e6:.1f} M | distinct per (dy,dz) pair {per_pair / 1e6:.1f} M "
              f"({per_pair / present:.2f}) | distinct per tile {per_tile / 1e6:.1f} M ({per_tile / present:.2f})", flush=True)
