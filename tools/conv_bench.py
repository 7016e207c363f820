#!/usr/bin/env python3
"""Development aid: time pcc_conv_fwd alone on synthetic neighbour tables to separate MFMA-pipe
efficiency from gather latency.  Patterns: 'local' (neighbours = nearby rows, L2-resident),
'random' (uniform random rows, HBM-resident), 'same' (every neighbour = row 0), 'streams' (27 sequential streams),
'runs32' (runs of 32 consecutive rows at random places)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pcc_amd
from pcc_amd import _lib
from pcc_amd._lib import ptr, check

dev = "cuda:0"
L = pcc_amd.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
shapes = [(128, 128), (64, 64), (128, 256)] if not os.environ.get("SHAPES") else [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]
K = 27
torch.manual_seed(0)
for cin, cout in shapes:
    F = torch.randn(n, cin, device=dev)
    W = torch.randn(K, cin, cout, device=dev) * 0.05
    if os.environ.get("ZERO_DATA"):      # does operand toggling (power) change the rate?
        F.zero_(); W.zero_()
    Wp = torch.empty(L.pcc_conv_packed_elems(K, cin, cout), device=dev)
    check(L.pcc_conv_pack_weights(ptr(W), K, cin, cout, ptr(Wp), _lib.stream()))
    out = torch.empty(n, cout, device=dev)
    order = torch.arange(n, dtype=torch.int32, device=dev)
    gmask = torch.full(((n + 31) // 32,), (1 << K) - 1, dtype=torch.int32, device=dev)
    rows = torch.arange(n, device=dev).unsqueeze(1)
    pats = {
        "same": torch.zeros(n, K, dtype=torch.int32, device=dev),
        "local": ((rows + torch.arange(K, device=dev).unsqueeze(0) * 3) % n).to(torch.int32),
        "random": torch.randint(0, n, (n, K), dtype=torch.int32, device=dev),
        # 27 sequential streams far apart (no L2 reuse between offsets, but consecutive rows gather consecutive rows)
        "streams": ((rows + torch.arange(K, device=dev).unsqueeze(0) * (n // K)) % n).to(torch.int32),
        # runs of 32 consecutive rows at random places (what a physically mask-sorted layout might give)
        "runs32": (((torch.randint(0, n // 32, (n // 32 + 1, K), device=dev).repeat_interleave(32, dim=0)[:n] * 32)
                    + (rows % 32)) % n).to(torch.int32),
    }
    for name, nbr in pats.items():
        nbr = nbr.contiguous()
        for it in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(L.pcc_conv_fwd(ptr(F), n, cin, ptr(W), ptr(Wp), None, ptr(nbr), ptr(order), ptr(gmask), K, ptr(out), n,
                                 cout, 0, None, None, _lib.stream()))
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        line = f"{cin:4d}->{cout:<4d} n={n} {name:7s} {ms:8.3f} ms  {2.0 * n * K * cin * cout / ms / 1e9:7.1f} TFLOP/s"
        print(line, flush=True)
