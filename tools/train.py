#!/usr/bin/env python3
"""Minimal trainer on the HIP training path (the shape of /root/reference/train.py:171-221: Q_Map generator,
model optimizer + bottleneck (aux) optimizer, gradient clipping, update() + state_dict checkpoint).  Data:
cubes cut from synthetic shells (no dataset in this environment) or from PLY files given with --ply.

  python tools/train.py --steps 300 --out gpurun_out/weights_synth.pt [--bf16] [--batch 8] [--block 128]
  python -m torch.distributed.run --nproc-per-node N tools/train.py ...        (data-parallel, bucketed RCCL averaging)
"""
import argparse, glob, json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the pool's host driver shares device memory between processes (RCCL) through dmabuf only
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--block", type=int, default=128)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--aux-lr", type=float, default=1e-3)
    ap.add_argument("--clip", type=float, default=1.0)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--ply", nargs="*", default=[])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "weights.pt"))
    ap.add_argument("--log-every", type=int, default=20)
    args = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    import torch.distributed as dist
    active = world > 1 or os.environ.get("PCC_BENCH_FORCE_DIST") == "1"       # see tools/train_bench.py: a world of one rank over RCCL
    if active:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29561")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    import pcc_amd
    from pcc_amd import autograd as ag, io, parallel as par, synthetic as syn
    from pcc_amd.loss import OURS_LOSS, Loss
    from pcc_amd.q_map import Q_Map
    from pcc_amd.utils import sparse_collate
    ag.set_bf16(args.bf16)
    torch.manual_seed(0)
    random.seed(1 + rank)
    model = syn.make_model(seed=0, device=dev)
    model.train()

    clouds = [io.read_ply(p) for p in args.ply] if args.ply else \
             [syn.sphere_shell(grid=1024, radius=r, half_width=0.5, noise=0.02, seed=i) for i, r in enumerate((180.0, 230.0, 260.0))]
    cubes = []
    for c in clouds:
        _, rows = par.split_blocks(c, args.block)
        cubes += [c[r] for r in rows if len(r) >= 1500]
    if rank == 0:
        print(f"{len(cubes)} cubes of edge {args.block} from {len(clouds)} clouds", flush=True)
    rng = random.Random(1234 + rank)
    params = [p for n, p in model.named_parameters() if not n.endswith(".quantiles")]          # train.py:63-64
    aux_params = [p for n, p in model.named_parameters() if n.endswith(".quantiles")]
    opt = torch.optim.Adam(params, lr=args.lr)
    aux_opt = torch.optim.Adam(aux_params, lr=args.aux_lr)
    red = par.GradBucketReducer(params, always_reduce=active)
    # the bottleneck (.quantiles) parameters are averaged across ranks as well: they set the CDF tables update()
    # builds, and every rank must end up with the same tables (rank 0 alone writes the checkpoint)
    aux_red = par.GradBucketReducer(aux_params, always_reduce=active)
    qgen = Q_Map({"mode": "exponential", "lambda_A_max": 12800, "lambda_A_min": 100, "lambda_G_max": 1600, "lambda_G_min": 25})
    loss_fn = Loss(OURS_LOSS)
    t0 = time.time()
    def batch():
        pick = rng.sample(range(len(cubes)), min(args.batch, len(cubes)))
        cs, fs = [], []
        for i in pick:
            p = cubes[i]
            cs.append(torch.from_numpy(p[:, :3] - np.floor(p[:, :3].min(axis=0) / args.block) * args.block))
            fs.append(torch.from_numpy(p[:, 3:6]))
        return sparse_collate(cs, fs)

    feed = pcc_amd.utils.Prefetcher(batch, depth=2)       # cut on a background thread, like the reference's DataLoader workers
    for step in range(1, args.steps + 1):
        C, F = next(feed)
        C, F = C.to(dev, non_blocking=True), F.to(dev, non_blocking=True)
        inp = pcc_amd.SparseTensor(coordinates=C, features=F, device=dev)
        Q, Lam = qgen(inp)
        opt.zero_grad(set_to_none=True)
        aux_opt.zero_grad(set_to_none=True)
        total, parts = loss_fn(inp, model(inp, Q, Lam))
        total.backward()
        red.finish()
        torch.nn.utils.clip_grad_norm_(params, args.clip)
        opt.step()
        aux = model.aux_loss()
        aux.backward()
        aux_red.finish()
        aux_opt.step()
        if rank == 0 and (step % args.log_every == 0 or step == 1):
            print(f"step {step:5d}  loss {float(total.detach()):9.3f}  " +
                  "  ".join(f"{k} {float(v.detach()):.3f}" for k, v in parts.items()) +
                  f"  aux {float(aux.detach()):.1f}  {time.time() - t0:.0f} s", flush=True)
    feed.close()
    if rank == 0:
        model.eval()
        model.update()
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        torch.save(model.state_dict(), args.out)
        print("saved", args.out, flush=True)
    if active:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
