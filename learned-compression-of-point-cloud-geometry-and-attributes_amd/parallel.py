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


def split_blocks_device(x, block):
    """split_blocks for a device tensor without leaving the device: (block ids [M,3] numpy, point counts per cube (list),
    cube index of every point (int64 device tensor)); cubes in the same lexicographic order.  The host version costs a
    pageable copy of the coordinates and a 850 k-key argsort per frame and rank — 30 ms in front of a 22 ms encode."""
    ids = torch.div(x[:, :3], block, rounding_mode="floor").to(torch.int64)
    key = (ids[:, 0] << 42) | (ids[:, 1] << 21) | ids[:, 2]
    uniq, inverse, counts = torch.unique(key, return_inverse=True, return_counts=True)       # sorted: lexicographic cubes
    uniq_h = uniq.cpu().numpy()
    block_ids = np.stack([(uniq_h >> 42) & 0x1FFFFF, (uniq_h >> 21) & 0x1FFFFF, uniq_h & 0x1FFFFF], axis=1)
    return block_ids, [int(v) for v in counts.cpu().tolist()], inverse


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
def pack_items_unit(strings, shape, k):
    """Container of a multi-item unit (compress(..., batch=...)): the reference header cannot hold one k per item,
    so: >4i [N_z, len_y, len_z, n_items], then 3 * n_items big-endian int32 counts (stage-major), y, z."""
    n_items = len(k[0])
    flat = [int(v) for stage in k for v in stage]
    hdr = struct.pack(">4i", int(shape[0]), len(strings[0][0]), len(strings[1][0]), n_items)
    return hdr + struct.pack(">%di" % len(flat), *flat) + strings[0][0] + strings[1][0]


def compress_blocks(model, x, q_feats, block, rank=0, world=1, batched=True):
    """Code the cubes of edge ``block`` assigned to ``rank``.

    x: float [N, 6] device tensor (xyz voxel coordinates + rgb), q_feats: [N, 2] (q_g, q_a).
    Coordinates stay absolute (stride alignment is the whole-frame one).  Returns
    (block ids [M,3], assignment, units).

    batched (default): the rank's cubes are the batch items of ONE compress call (the
    model counts k and selects top-k per item) -> one unit (block indices, strings, shape, k, coords8 with the
    item index in column 0).  Kernels then see all of the rank's points at once instead of one small cube at a
    time.  batched=False: one call and one unit (block index, strings, shape, k, coords8) per cube.
    """
    from .sparse import SparseTensor
    x = x.detach()
    ids, counts, cube_of_point = split_blocks_device(x, block)
    parts = assign_blocks(counts, world)
    units = []
    mine = parts[rank]

    def rows_of(cubes):
        """(row indices, item index per row) of the given cubes: cube after cube, original row order inside a cube"""
        lut = torch.full((len(counts),), -1, dtype=torch.int64, device=x.device)
        lut[torch.tensor(cubes, dtype=torch.int64, device=x.device)] = torch.arange(len(cubes), dtype=torch.int64, device=x.device)
        item_all = lut[cube_of_point]
        sel = torch.nonzero(item_all >= 0).flatten()
        item, order = torch.sort(item_all[sel], stable=True)
        return sel[order], item.to(torch.int32)

    if batched:
        if mine:
            sel, bt = rows_of(list(mine))
            xb = x.index_select(0, sel)
            qc = torch.cat([bt.reshape(-1, 1).to(xb.dtype), xb[:, :3]], dim=1)
            Q = SparseTensor(coordinates=qc, features=q_feats.index_select(0, sel), device=x.device, nbatch=len(mine))
            strings, shape, k, coords = model.compress(xb, Q, batch=bt)
            units.append((list(mine), strings, shape, k, coords))
        return ids, parts, units
    for b in mine:
        sel, _ = rows_of([b])
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


# ---------------------------------------------------------------------------------------------
# data-parallel training (BASELINE config 5; the reference trains on one GPU, train.py:171-221)
# ---------------------------------------------------------------------------------------------
class GradBucketReducer:
    """Gradient averaging across ranks, overlapped with backward.

    Parameters are laid out (in reverse registration order — roughly the order backward produces their
    gradients) into flat fp32 buckets of ``bucket_bytes``; a bucket is all-reduced asynchronously as soon
    as the last of its gradients has been accumulated (post-accumulate-grad hooks), so the collectives
    of the late layers run under the backward of the early ones.  xGMI is point-to-point: a ring
    all-reduce moves 2 (W-1)/W of a bucket over one ~150 GB/s link per hop, so buckets are sized in
    tens of MB — large enough to be bandwidth-bound, small enough to leave backward to hide behind
    (31.5 M parameters = 126 MB = 4 buckets).  Parameters that received no gradient this step (the
    codec has structurally unused ones: gdn, conv_layers, q_up_i.conv_2) contribute zeros, so every
    rank issues the same collectives whatever its data.

    Two collective schedules per bucket (``mode``), the same averaged gradients on every rank either way:
    * ``"all_reduce"`` (default): one asynchronous all-reduce launched from the backward hooks;
    * ``"reduce_scatter"``: an asynchronous reduce-scatter from the hooks (each rank ends up owning the sum of a 1/W
      slice), and in ``finish()`` the scaling of the owned slice followed by an all-gather.  On the point-to-point
      xGMI mesh of an 8 x MI355X node this is the schedule SURVEY.md §5 argues for: the reduce-scatter half hides
      under backward like the all-reduce did, the all-gather half moves (W-1)/W of a bucket once, and a sharded
      optimizer could later step between the two halves.  Which one is faster on the node is a measurement the 8-GPU
      box has to make (tools/train_bench.py --reducer); both are covered by world-2 gloo tests and run over RCCL with
      a world of one rank.

    usage:  red = GradBucketReducer(model.parameters());  loss.backward();  red.finish();  opt.step()
    """

    def __init__(self, params, bucket_bytes=32 << 20, group=None, always_reduce=False, mode="all_reduce"):
        """always_reduce: issue the collectives even in a world of one rank (a process group must be up) — how a one-GPU
        box exercises the RCCL calls of the N-GPU path; the result is unchanged (sum over one rank, divided by one)"""
        if mode not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"GradBucketReducer: unknown mode {mode!r}")
        self.mode = mode
        self.group = group
        up = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if up else 1
        self.collective = self.world > 1 or (bool(always_reduce) and up)
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []          # (flat tensor, [(param, offset, numel)])
        cur, cur_elems = [], 0
        limit = max(1, bucket_bytes // 4)
        for p in reversed(self.params):
            if cur and cur_elems + p.numel() > limit:
                self._close(cur, cur_elems)
                cur, cur_elems = [], 0
            cur.append(p)
            cur_elems += p.numel()
        if cur:
            self._close(cur, cur_elems)
        self._pending = [0] * len(self.buckets)
        self._works = [None] * len(self.buckets)
        self._where = {}
        for b, (_, items) in enumerate(self.buckets):
            for p, _, _ in items:
                self._where[p] = b
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._arm()

    def _close(self, plist, elems):
        if self.mode == "reduce_scatter":
            elems_p = (elems + self.world - 1) // self.world * self.world          # equal slices: zero padding at the tail
        else:
            elems_p = elems
        flat = torch.zeros(elems_p, dtype=torch.float32, device=plist[0].device)
        items, off = [], 0
        for p in plist:
            items.append((p, off, p.numel()))
            off += p.numel()
        self.buckets.append((flat, items))

    def _arm(self):
        for b, (_, items) in enumerate(self.buckets):
            self._pending[b] = len(items)
            self._works[b] = None
        self._seen = set()

    def _launch(self, b):
        flat, items = self.buckets[b]
        for p, off, n in items:
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
        if self.collective and self.mode == "reduce_scatter":
            n_shard = flat.numel() // self.world
            shard = flat.new_empty(n_shard)
            work = dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works[b] = (work, shard)
        elif self.collective:
            self._works[b] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        if p in self._seen:
            return
        self._seen.add(p)
        b = self._where[p]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def finish(self):
        """wait for the collectives, write the averaged gradients back, re-arm for the next step"""
        for b, (flat, items) in enumerate(self.buckets):
            if self._pending[b] > 0:          # holds parameters without a gradient this step
                self._launch(b)
            if isinstance(self._works[b], tuple):                  # reduce-scatter done: scale my slice, gather all slices
                work, shard = self._works[b]
                work.wait()
                if self.world > 1:
                    shard.div_(self.world)
                dist.all_gather_into_tensor(flat, shard, group=self.group)
            else:
                if self._works[b] is not None:
                    self._works[b].wait()
                if self.world > 1:
                    flat.div_(self.world)
            for p, off, n in items:
                if p.grad is not None:
                    p.grad.copy_(flat[off:off + n].view_as(p.grad))
                elif self.world > 1:
                    p.grad = flat[off:off + n].view_as(p).clone()
        self._arm()

    def close(self):
        for h in self._handles:
            h.remove()
