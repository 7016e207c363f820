"""Data-parallel gradient averaging (parallel.GradBucketReducer) on CPU: world-size-2 gloo processes,
the collective pattern bench / training run over RCCL on the GPUs."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(64, 256)
        self.b = torch.nn.Linear(256, 256)
        self.unused = torch.nn.Linear(8, 8)          # never produces a gradient, like the codec's gdn / conv_layers
        self.c = torch.nn.Linear(256, 3)

    def forward(self, x):
        h = torch.relu(self.a(x))
        return self.c(torch.relu(self.b(h)) + h)     # h is used twice: one accumulation per parameter all the same


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcc_amd.parallel import GradBucketReducer
    torch.manual_seed(0)
    net = Net()                                       # same initial weights on every rank
    red = GradBucketReducer(net.parameters(), bucket_bytes=64 * 1024)      # several buckets
    assert len(red.buckets) >= 3
    ok = True
    for step in range(3):
        g = torch.Generator().manual_seed(1000 * step + rank)
        x = torch.randn(32 + 8 * rank, 64, generator=g)                     # ranks see different (ragged) data
        net.zero_grad(set_to_none=True)
        net(x).square().mean().backward()
        local = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        red.finish()
        for n, p in net.named_parameters():
            if n.startswith("unused"):
                ok &= p.grad is not None and float(p.grad.abs().max()) == 0.0
                continue
            want = local[n].clone()
            dist.all_reduce(want)
            want /= world
            ok &= torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7)
    red.close()
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.destroy_process_group()


def test_bucketed_gradient_average_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), f"ok{r}")).read() == "1"


def test_single_process_is_identity():
    sys.path.insert(0, ROOT)
    from pcc_amd.parallel import GradBucketReducer
    torch.manual_seed(0)
    net = Net()
    red = GradBucketReducer(net.parameters(), bucket_bytes=64 * 1024)
    net(torch.randn(16, 64)).sum().backward()
    before = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    red.finish()
    for n, p in net.named_parameters():
        if n in before:
            assert torch.equal(p.grad, before[n])
    red.close()
