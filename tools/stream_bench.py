#!/usr/bin/env python3
"""Streamed sequence (BASELINE config 4 shape on one GPU): F frames of the config-2 shell with per-frame
radius jitter, coded and decoded by W worker threads of ONE process, each on its own HIP stream.  While one
worker waits for its serial host range coder (16-17 ms per frame, GIL released) the other's kernels run.
Not the headline metric: bench.py times one frame at a time.

  python tools/stream_bench.py [--frames 12] [--workers 2]"""
import argparse, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pcc_amd
from pcc_amd import synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--workers", type=int, default=2)
args = ap.parse_args()
dev = "cuda:0"
model = syn.make_model(seed=0, device=dev); model.update()
frames = []
for f in range(4):                                   # 4 distinct geometries, reused round-robin
    cfg = dict(syn.CONFIG2); cfg["radius"] -= 0.25 * f
    pts = syn.sphere_shell(**cfg)
    qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)
    frames.append((torch.from_numpy(pts).to(dev), torch.from_numpy(qc).to(dev), torch.from_numpy(qf).to(dev)))

def code(i):
    x, qc, qf = frames[i % len(frames)]
    Q = pcc_amd.SparseTensor(coordinates=qc, features=qf, device=dev)
    strings, shape, k, coords = model.compress(x, Q)
    rec = model.decompress(coordinates=coords, strings=strings, shape=shape, k=k)
    return rec.shape[0], pcc_amd.utils.count_bits(strings)

def run(workers, n_frames):
    nxt, lock, done = [0], threading.Lock(), []
    def worker():
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            while True:
                with lock:
                    i = nxt[0]; nxt[0] += 1
                if i >= n_frames:
                    break
                done.append(code(i))
            s.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ths = [threading.Thread(target=worker) for _ in range(workers)]
    for wi, t in enumerate(ths):
        if wi:
            time.sleep(STAGGER_S)                      # frames arrive one after the other (bench.py: lock-step otherwise)
        t.start()
    [t.join() for t in ths]
    torch.cuda.synchronize()
    return time.perf_counter() - t0, done

STAGGER_S = 0.0
for i in range(2): code(i)                            # warm-up (weight packing caches, allocator)
torch.cuda.synchronize(); _t = time.perf_counter(); code(0); torch.cuda.synchronize()
STAGGER_S = 0.5 * (time.perf_counter() - _t)          # half a sequential frame
out = {}
for w in sorted({1, args.workers}):
    run(w, w * 2)                                     # per-thread warm-up (pinned staging, side streams)
    el, done = run(w, args.frames)
    pts_total = sum(d[0] for d in done)
    out[f"workers_{w}"] = {"frames": args.frames, "seconds": el, "ms_per_frame": 1e3 * el / args.frames, "mpoints_per_s": pts_total / el / 1e6}
    print(w, "workers:", out[f"workers_{w}"], flush=True)
print(json.dumps(out))
