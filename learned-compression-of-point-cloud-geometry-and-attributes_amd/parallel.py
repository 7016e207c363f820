"""Multi-GPU partitioning of the codec (one process per GPU, torch.distributed over RCCL/xGMI).

The reference has no distributed code at all.  The path shards only over
independent units:

1. **frames** — the codec is intra-only, so frame f is coded by rank f mod world; every rank runs
   the unmodified single-GPU path and results are identical to a 1-GPU run;
2. **spatial blocks as batch items** — a frame is cut into cubes (the reference trains on exactly
   such cubes: data/dataloader.py:206-238, data/datasets/full_128/config.yaml:1-2); each cube is
   an independent unit with its own (header, y, z) triplet.  This changes the numbers relative to
   whole-frame coding, so its parity target is the oracle run with the same partition.

There is no collective inside the data path.  The only exchange is the all-gather(v) of the
finished bitstreams (10-100 KB per frame: latency-bound), done as lengths first, then padded
payloads — one fused all_gather each instead of a ring of point-to-point sends.
Works with the ``nccl`` (= RCCL) backend on GPUs and with ``gloo`` on CPU tensors (tests).
"""
import struct

import numpy as np
import torch
import torch.distributed as dist


def frames_for_rank(n_frames, rank, world):
    """Frame indices coded by ``rank`` (round-robin keeps a streamed sequence balanced)."""
    return list(range(rank, n_frames, world))


def split_blocks(points, block):
    """Cut a cloud [N, >=3] (xyz first) into cubes of edge ``block``; returns (block ids [M,3], list of
    row-index arrays), cubes ordered lexicographically.  Slicing rule of data/dataloader.py:206-238."""
    xyz = np.asarray(points)[:, :3]
    ids = np.floor_divide(xyz, block).astype(np.int64)
    key = (ids[:, 0] << 42) | (ids[:, 1] << 21) | ids[:, 2]
    order = np.argsort(key, kind="stable")
    uniq, start = np.unique(key[order], return_index=True)
    bounds = list(start) + [len(order)]
    rows = [order[bounds[i]:bounds[i + 1]] for i in range(len(uniq))]
    block_ids = np.stack([(uniq >> 42) & 0x1FFFFF, (uniq >> 21) & 0x1FFFFF, uniq & 0x1FFFFF], axis=1)
    return block_ids, rows


def assign_blocks(counts, world):
    """Greedy longest-processing-time assignment of blocks to ranks by point count.
    Returns a list (per rank) of block indices; deterministic."""
    loads = [0] * world
    out = [[] for _ in range(world)]
    for b in sorted(range(len(counts)), key=lambda i: (-counts[i], i)):
        r = min(range(world), key=lambda i: (loads[i], i))
        out[r].append(b)
        loads[r] += counts[b]
    return [sorted(v) for v in out]


def pack_unit(strings, shape, k, coords_bytes=b""):
    """One coded unit as bytes: the reference container layout (model/model.py:241-256)."""
    ks = [int(kk[0]) if isinstance(kk, (list, tuple)) else int(kk) for kk in k]
    hdr = struct.pack(">7i", int(shape[0]), len(coords_bytes), len(strings[0][0]), len(strings[1][0]), *ks)
    return hdr + coords_bytes + strings[0][0] + strings[1][0]


def all_gather_bitstreams(payload, device, group=None):
    """All ranks contribute ``payload`` (bytes); every rank gets the list of all payloads in rank
    order.  Two collectives: lengths (world x int64), then payloads padded to the longest."""
    world = dist.get_world_size(group)
    mine = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    lens = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(lens, mine, group=group)
    lens_host = lens.cpu().tolist()
    mx = max(max(lens_host), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=device)
    if payload:
        buf[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    out = torch.empty(world * mx, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, buf, group=group)
    host = out.cpu().numpy()
    return [host[r * mx: r * mx + lens_host[r]].tobytes() for r in range(world)]


# ---------------------------------------------------------------------------------------------
# spatial-block mode: every cube is an independent unit coded by the unmodified single-GPU path
# ---------------------------------------------------------------------------------------------
def compress_blocks(model, x, q_feats, block, rank=0, world=1):
    """Code the cubes of edge ``block`` assigned to ``rank``.

    x: float [N, 6] device tensor (xyz voxel coordinates + rgb), q_feats: [N, 2] (q_g, q_a).
    Coordinates stay absolute (stride alignment is the whole-frame one).  Returns
    (block ids [M,3], assignment, units) with units = [(block index, strings, shape, k, coords8)].
    """
    from .sparse import SparseTensor
    xyz = x[:, :3].detach().cpu().numpy()
    ids, rows = split_blocks(xyz, block)
    parts = assign_blocks([len(r) for r in rows], world)
    units = []
    for b in parts[rank]:
        sel = torch.from_numpy(rows[b]).to(x.device)
        xb = x.index_select(0, sel)
        qc = torch.cat([torch.zeros((xb.shape[0], 1), device=x.device), xb[:, :3]], dim=1)
        Q = SparseTensor(coordinates=qc, features=q_feats.index_select(0, sel), device=x.device)
        strings, shape, k, coords = model.compress(xb, Q)
        units.append((b, strings, shape, k, coords))
    return ids, parts, units


def decompress_blocks(model, units):
    """Decode units produced by ``compress_blocks`` -> float [N_hat, 6] (concatenated cubes)."""
    out = [model.decompress(coordinates=coords, strings=strings, shape=shape, k=k)
           for _, strings, shape, k, coords in units]
    return torch.cat(out, dim=0)
