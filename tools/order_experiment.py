#!/usr/bin/env python3
"""Offline (CPU) experiment behind the execution-order key of csrc/select.hip: issued / useful MFMA rows of 32-row tiles
under different row orders, on a dense surface, random subsets of it (the sparse sets an untrained decoder keeps) and a
set of generative children."""
import numpy as np, sys
sys.path.insert(0, '/root/repo')
import pcc_amd
from oracle import coords as oc
rng = np.random.default_rng(0)
def masks_of(c):
    nbr = oc.kernel_map(c, c, 3, 1)
    m = ((nbr >= 0).astype(np.int64) << np.arange(27)).sum(axis=1)
    pc = np.array([bin(int(v)).count("1") for v in m])
    return m, pc
def issued(m, order):
    mm = m[order]
    pad = (-len(mm)) % 32
    mm = np.concatenate([mm, np.zeros(pad, np.int64)]).reshape(-1, 32)
    u = np.bitwise_or.reduce(mm, axis=1)
    return sum(bin(int(v)).count("1") for v in u) * 32
def keyperm(m, perm):
    key = np.zeros(len(m), np.int64)
    for rank, b in enumerate(perm):
        key |= ((m >> b) & 1) << (26 - rank)
    return key
def run(name, c):
    m, pc = masks_of(c)
    alg = pc.sum()
    freq = ((m[:, None] >> np.arange(27)) & 1).mean(axis=0)
    asc = np.argsort(freq)
    res = {
      "current (27-popc, mask)": np.argsort(((27 - pc).astype(np.int64) << 27) | m, kind="stable"),
      "mask only": np.argsort(m, kind="stable"),
      "rare bits first": np.argsort(keyperm(m, asc), kind="stable"),
      "rare bits first, descending": np.argsort(-keyperm(m, asc), kind="stable"),
      "popc>>2 then rare-first": np.argsort((((27 - pc) >> 2).astype(np.int64) << 27) | keyperm(m, asc), kind="stable"),
    }
    print(f"== {name}: rows {len(c)} nbrs/row {pc.mean():.1f} distinct masks {len(np.unique(m))}")
    for k, o in res.items():
        print(f"   {k:32s} issued/alg {issued(m, o) / alg:.3f}")
pts = pcc_amd.synthetic.sphere_shell(grid=512, radius=130.0, half_width=0.5)[:, :3].astype(np.int32)
full = np.concatenate([np.zeros((len(pts), 1), np.int32), pts], axis=1)
run("dense surface", full)
for frac in (0.6, 0.3):
    keep = rng.random(len(pts)) < frac
    run(f"random {frac:.0%} of the surface", full[keep])
# candidate set: k3 children of a stride-2 surface
par = pcc_amd.synthetic.sphere_shell(grid=256, radius=65.0, half_width=0.5)[:, :3].astype(np.int32)
parc = np.concatenate([np.zeros((len(par), 1), np.int32), par * 2], axis=1)
cand = oc.children(parc, 2, 3) if hasattr(oc, "children") else None
if cand is not None:
    run("k3 children (candidates)", cand)
