#!/usr/bin/env python3
"""Micro-benchmark of the coordinate-side kernels on config-2-sized sets (HIP events, median of 5):
generative children of a 850 k-row stride-2 shell (~5 M candidates), the transposed parent->candidate kernel map,
the candidate->candidate kernel map, and the MFMA execution order (radix sort + permuted table) of both.

"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pcc_amd
from pcc_amd import synthetic as syn

dev = "cuda:0"
pts = syn.sphere_shell(**syn.CONFIG2)[:, :3].astype(np.int32)
par = np.concatenate([np.zeros((pts.shape[0], 1), np.int32), pts * 2], axis=1)           # stride-2 parents
par_d = torch.from_numpy(par).to(dev)


def timed(fn, reps=5):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return statistics.median(out), r


def fresh():
    return pcc_amd.CoordMap(par_d, 2, nbatch=1)


t, cand = timed(lambda: fresh().up(3))
print(f"children (unique) of {par.shape[0]} parents -> {cand.n} candidates: {t:.3f} ms")
p = fresh()
cand = p.up(3)
p.table(); cand.table()
def pc():
    for k in [k for k in p._cache if k[0] in ('kmap', 'okmap')]:
        p._cache.pop(k)
    return p.kernel_map(cand, 3, True)


t, _ = timed(pc)
nbr, mask, pairs = p.kernel_map(cand, 3, True)
print(f"kernel map parent->cand (transposed), {cand.n} rows: {t:.3f} ms  pairs/row {int(pairs.item()) / cand.n:.2f}")


def cc():
    for k in [k for k in cand._cache if k[0] in ('kmap', 'okmap')]:
        cand._cache.pop(k)
    return cand.kernel_map(cand, 3)


t, (nbr, mask, pairs) = timed(cc)
print(f"kernel map cand->cand, {cand.n} rows: {t:.3f} ms  pairs/row {int(pairs.item()) / cand.n:.2f}  ({27 * cand.n / t / 1e6:.1f} G probes/s)")


def order():
    for k in [k for k in cand._cache if k[0] == 'okmap']:
        cand._cache.pop(k)
    return cand.ordered_kernel_map(cand, 3)


t, _ = timed(order)
print(f"execution order (sort + permuted table) of {cand.n} rows: {t:.3f} ms")
for n in (1200, 4800, 19256, 72752, 265512, 850824):
    sub = pcc_amd.CoordMap(par_d[:n].contiguous(), 2, nbatch=1)
    sub.kernel_map(sub, 3)

    def o():
        for k in [k for k in sub._cache if k[0] == 'okmap']:
            sub._cache.pop(k)
        return sub.ordered_kernel_map(sub, 3)
    t, _ = timed(o)
    t2, _ = timed(lambda: sub.sort_permutation())
    print(f"  n={n:7d}: execution order {t * 1e3:8.1f} us   canonical sort (64-bit keys) {t2 * 1e3:8.1f} us")
