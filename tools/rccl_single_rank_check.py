#!/usr/bin/env python3
"""Exercise the `nccl` (= RCCL) code paths of parallel.py on ONE GPU with a world of one rank: the collectives degenerate
to copies, but the calls, dtypes (uint8 / int64 / float32 device tensors) and the process-group setup are the ones an
N-GPU run makes — a one-GPU box cannot run more than one rank per device.  Not a measurement.
  python tools/rccl_single_rank_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import pcc_amd
from pcc_amd import parallel as par
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
out = par.all_gather_bitstreams(b"hello bitstream" * 1000, dev)
assert out == [b"hello bitstream" * 1000]
assert par.all_gather_bitstreams(b"", dev) == [b""]
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
n = torch.tensor([7], dtype=torch.int64, device=dev)
dist.all_reduce(n)
dist.barrier()
lin = torch.nn.Linear(64, 64).to(dev)
red = par.GradBucketReducer(lin.parameters(), bucket_bytes=4096, always_reduce=True)
assert red.collective and len(red.buckets) > 1
lin(torch.randn(8, 64, device=dev)).sum().backward()
g0 = lin.weight.grad.clone()
red.finish()
assert torch.equal(lin.weight.grad, g0)            # the mean over one rank
dist.destroy_process_group()
print("rccl single-rank check ok")
