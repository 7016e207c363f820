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


def _worker(rank, world, port, out_dir, mode="all_reduce"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcc_amd.parallel import GradBucketReducer
    torch.manual_seed(0)
    net = Net()                                       # same initial weights on every rank
    red = GradBucketReducer(net.parameters(), bucket_bytes=64 * 1024, mode=mode)      # several buckets
    assert len(red.buckets) >= 3
    if mode == "reduce_scatter":                      # equal slices per rank; at least one bucket needed tail padding
        assert all(flat.numel() % world == 0 for flat, _ in red.buckets)
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


@pytest.mark.parametrize("mode,world", [("all_reduce", 2), ("reduce_scatter", 2), ("reduce_scatter", 3)])
def test_bucketed_gradient_average(tmp_path, mode, world):
    """both collective schedules (one all-reduce per bucket; reduce-scatter from the hooks + all-gather in finish()) hand
    every rank the mean gradient; world 3 makes the reduce-scatter slices need tail padding (bucket sizes not divisible)"""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), f"ok{r}")).read() == "1"


def test_reducer_rejects_an_unknown_mode():
    sys.path.insert(0, ROOT)
    from pcc_amd.parallel import GradBucketReducer
    with pytest.raises(ValueError):
        GradBucketReducer(Net().parameters(), mode="ring")


def _aux_worker(rank, world, port, out_dir):
    """the bottleneck optimiser of tools/train.py (train.py:205-211 of the reference) under data parallelism: main
    parameters see rank-dependent gradients, the .quantiles parameters go through their own reducer, and after a
    few steps update() must build IDENTICAL CDF tables on every rank"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hashlib
    import pcc_amd                                                  # noqa: F401  (import alias of the package)
    from pcc_amd.entropy import EntropyBottleneck
    from pcc_amd.parallel import GradBucketReducer
    torch.manual_seed(0)
    eb = EntropyBottleneck(8)                                         # same initial state on every rank
    params = [p for n, p in eb.named_parameters() if not n.endswith("quantiles")]
    aux_params = [p for n, p in eb.named_parameters() if n.endswith("quantiles")]
    opt, aux_opt = torch.optim.Adam(params, lr=1e-2), torch.optim.Adam(aux_params, lr=5e-2)
    red, aux_red = GradBucketReducer(params), GradBucketReducer(aux_params)
    for step in range(5):
        g = torch.Generator().manual_seed(100 * step + rank)         # ranks see different data
        v = torch.randn(1, 8, 40 + 8 * rank, generator=g) * (3.0 + rank)
        opt.zero_grad(set_to_none=True)
        aux_opt.zero_grad(set_to_none=True)
        _, lik = eb(v, training=True)
        (-torch.log2(lik).mean()).backward()
        red.finish()
        opt.step()
        aux = eb.loss()
        aux.backward()
        aux_red.finish()
        aux_opt.step()
    red.close(); aux_red.close()
    eb.update(force=True)
    cdf, length, offset = eb.tables()
    h = hashlib.sha256()
    import numpy as np
    for t in (cdf, length, offset, eb.quantiles.detach()):
        h.update(np.ascontiguousarray(t.cpu().numpy() if torch.is_tensor(t) else t).tobytes())
    with open(os.path.join(out_dir, f"tables{rank}"), "w") as f:
        f.write(h.hexdigest())
    dist.destroy_process_group()


def test_bottleneck_quantiles_stay_identical_across_ranks(tmp_path):
    """without averaging the aux gradients the EB quantiles drift per rank and ranks would code
    with different CDF tables"""
    world, port = 2, _free_port()
    mp.spawn(_aux_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    digests = [open(os.path.join(str(tmp_path), f"tables{r}")).read() for r in range(world)]
    assert len(digests[0]) == 64 and digests[0] == digests[1]


def test_aux_loss_matches_oracle_and_only_reaches_quantiles():
    """model/model.py:40-47 -> compressai EntropyBottleneck.loss(): |logits_cumulative(quantiles) - target| summed,
    density parameters held fixed (stop_gradient)"""
    sys.path.insert(0, ROOT)
    import numpy as np
    import pcc_amd                                                  # noqa: F401
    from pcc_amd.entropy import EntropyBottleneck
    from oracle.entropy import EntropyBottleneck as OracleEB
    torch.manual_seed(3)
    eb = EntropyBottleneck(16)
    with torch.no_grad():
        eb.quantiles.add_(torch.randn_like(eb.quantiles) * 0.7)
        for i in range(5):
            getattr(eb, f"_matrix{i}").add_(torch.randn_like(getattr(eb, f"_matrix{i}")) * 0.3)
    loss = eb.loss()
    loss.backward()
    for n, p in eb.named_parameters():
        assert (p.grad is not None) == n.endswith("quantiles"), n
    sd = {k: v.detach() for k, v in eb.state_dict().items()}
    want = float(OracleEB(sd).aux_loss())
    assert float(loss.detach()) == pytest.approx(want, rel=1e-5)


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
