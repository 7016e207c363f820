"""GPU time of one top-k selection (pcc_topk_mask, one batch item) at the three candidate-set sizes of a config-2 frame: run under
`rocprofv3 --kernel-trace --stats -- python3 tools/topk_bench.py` and sum the topk_* kernels (the top-k note);
PCC_TOPK_SMALL=0 = the separate-launch form of rounds 1-3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcc_amd
from pcc_amd import sparse as sp
dev = "cuda:0"
for n, k in ((5_160_000, 850_000), (1_260_000, 265_000), (233_000, 72_000)):
    g = torch.Generator(device="cpu").manual_seed(1)
    logits = (torch.randn(n, 1, generator=g) * 2.0 - 1.0).to(dev)
    flat = torch.arange(n)
    side = 1024
    c = torch.stack([torch.zeros(n, dtype=torch.int64), flat // (side * side), (flat // side) % side, flat % side], 1).to(torch.int32).to(dev)
    for _ in range(3):
        m = sp.topk_mask(logits, c, [k], 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m = sp.topk_mask(logits, c, [k], 1)
    e1.record(); torch.cuda.synchronize()
    print(f"PCC_TOPK_SMALL={os.environ.get('PCC_TOPK_SMALL', '1')} n={n} k={k}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  kept={int(m.sum())}")
