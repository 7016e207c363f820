#!/usr/bin/env python3
"""Training-step throughput (BASELINE config 5 shape: cubes cut from a 10-bit frame, batch of cubes per
step, Adam, data-parallel gradient averaging over RCCL).  fp32 by default; PCC_TRAIN_BF16=1 runs the wide
convolutions (forward, backward-data, weight gradient) on bf16 operands with fp32 accumulation.  Synthetic data: the config-2 shell cut into 128^3 cubes
(data/datasets/full_128), colours as in bench.py; seeded weights.

  python tools/train_bench.py [--gpus N] [--batch 8] [--steps 5] [--warmup 2] [--block 128]
  --gpus N > 1 without RANK in the environment starts the N ranks itself (bench.launch_ranks: a torchrun child process,
  one rank per GPU, before this process touches the GPU); under an external launcher
  (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py --gpus N ...)
  WORLD_SIZE must equal --gpus.
"""
import argparse, json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the pool's host driver shares device memory between processes (RCCL) through dmabuf only
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs; default: WORLD_SIZE of the launcher, else 1")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--block", type=int, default=128)
    ap.add_argument("--reducer", default="all_reduce", choices=["all_reduce", "reduce_scatter"],
                    help="collective schedule of the gradient buckets (parallel.GradBucketReducer): one all-reduce per bucket, or "
                         "reduce-scatter under backward + all-gather before the optimizer step")
    args = ap.parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 and "RANK" not in os.environ:
        from bench import launch_ranks
        sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if int(os.environ.get("WORLD_SIZE", 1)) != args.gpus:
        print(f"train_bench.py: WORLD_SIZE={os.environ.get('WORLD_SIZE', 1)} but --gpus {args.gpus}", file=sys.stderr, flush=True)
        sys.exit(2)
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    rehearse = os.environ.get("PCC_BENCH_REHEARSE") == "1"      # one-GPU rehearsal of the N > 1 control flow: gloo, every rank on cuda:0
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    import torch.distributed as dist
    # PCC_BENCH_FORCE_DIST=1: a world of ONE rank over nccl — the gradient reducer's bucketed all-reduces, the barriers and the
    # timing all-reduces are real RCCL calls (degenerate collectives); what a one-GPU box can exercise of the N-GPU path
    active = world > 1 or os.environ.get("PCC_BENCH_FORCE_DIST") == "1"
    if active:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29551")
        dist.init_process_group("gloo" if rehearse else "nccl", rank=rank, world_size=world)
    from bench import rank_devices
    rccl = rank_devices(dist if active else None, torch.device(dev), world, rehearse)
    import pcc_amd
    from pcc_amd import parallel as par, synthetic as syn
    from pcc_amd.loss import OURS_LOSS, Loss
    from pcc_amd.q_map import Q_Map

    model = syn.make_model(seed=0, device=dev)
    model.train()
    pts = syn.sphere_shell(**syn.CONFIG2, noise=0.02)
    _, rows = par.split_blocks(pts, args.block)
    rows = [r for r in rows if len(r) >= 2000]                 # drop slivers, like the reference's min-points filter
    rng = random.Random(1234 + rank)
    params = [p for n, p in model.named_parameters() if not n.endswith(".quantiles")]      # train.py:63-64
    opt = torch.optim.Adam(params, lr=1e-4)
    red = par.GradBucketReducer(params, always_reduce=active, mode=args.reducer)
    qgen = Q_Map({"mode": "exponential", "lambda_A_max": 12800, "lambda_A_min": 100, "lambda_G_max": 1600, "lambda_G_min": 25})
    loss_fn = Loss(OURS_LOSS)
    random.seed(99 + rank)

    def batch():
        pick = rng.sample(range(len(rows)), args.batch)
        cs, fs = [], []
        for b, i in enumerate(pick):
            p = pts[rows[i]]
            xyz = p[:, :3] - np.floor(p[:, :3].min(axis=0) / args.block) * args.block       # cube-local coordinates
            cs.append(np.concatenate([np.full((p.shape[0], 1), b, np.float32), xyz], axis=1))
            fs.append(p[:, 3:])
        return torch.from_numpy(np.concatenate(cs)), torch.from_numpy(np.concatenate(fs))

    # batches are cut on a background thread while the GPU works (the reference's DataLoader workers, train.py:171-187)
    feed = pcc_amd.utils.Prefetcher(batch, depth=2)

    def step():
        c, f = next(feed)
        c, f = c.to(dev, non_blocking=True), f.to(dev, non_blocking=True)
        inp = pcc_amd.SparseTensor(coordinates=c, features=f, device=dev)
        Q, Lam = qgen(inp)
        opt.zero_grad(set_to_none=True)
        total, _ = loss_fn(inp, model(inp, Q, Lam))
        total.backward()
        red.finish()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        return int(c.shape[0]), float(total.detach())

    for _ in range(args.warmup):
        step()
    if active:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    npts, last = 0, None
    for _ in range(args.steps):
        n, last = step()
        npts += n
    if active:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if active:
        t = torch.tensor([el, float(npts)], dtype=torch.float64, device="cpu" if rehearse else dev)
        tm = t.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ts = t.clone(); dist.all_reduce(ts)
        el, npts = float(tm[0]), float(ts[1])
    if rank == 0:
        print(json.dumps({"metric": "training points/sec", "value": npts / el, "unit": "points/s", "n_gpus": world, "steps": args.steps,
                          "ms_per_step": el / args.steps * 1e3, "batch_cubes_per_gpu": args.batch, "block": args.block,
                          "points_per_step": npts / args.steps, "dtype": ("bf16 operands on the wide convolutions, fp32 accumulation" if os.environ.get("PCC_TRAIN_BF16") == "1" else "f32"), "last_loss": last, "data": "synthetic", "reducer": args.reducer, "rccl": rccl}))
    feed.close()
    if active:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
