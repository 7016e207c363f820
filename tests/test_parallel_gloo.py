"""N > 1 host logic on CPU: world_size-2 gloo processes exercise the frame sharding and the
bitstream all-gather(v) that bench.py runs over RCCL on the GPUs."""
import os
import socket
import sys

import numpy as np
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


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pcc_amd
    from pcc_amd import parallel as par
    frames = par.frames_for_rank(7, rank, world)
    rng = np.random.default_rng(100 + rank)
    mine = []
    for f in frames:     # stand-in bitstreams of ragged length (rank 1's first one is empty)
        n = 0 if (rank == 1 and f == frames[0]) else int(rng.integers(1, 5000))
        strings = [[bytes(rng.integers(0, 256, n, dtype=np.uint8))], [bytes([f] * 3)]]
        mine.append(par.pack_unit(strings, [f], [[1], [2], [3]]))
    got = []
    for i in range(max(len(par.frames_for_rank(7, r, world)) for r in range(world))):
        payload = mine[i] if i < len(mine) else b""
        got.append(par.all_gather_bitstreams(payload, torch.device("cpu")))
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([len(b) for row in got for b in row]))
    # every rank must see byte-identical results
    flat = b"".join(b for row in got for b in row)
    t = torch.tensor([sum(flat) % (2 ** 31), len(flat)], dtype=torch.int64)
    ref = t.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(t, ref)
    # own payloads come back unchanged at the right slot
    for i, p in enumerate(mine):
        assert got[i][rank] == p
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharding_and_bitstream_allgather_gloo(tmp_path):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    a, b = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert (a == b).all() and a.size == 4 * world


def test_partition_helpers(pcc):
    from pcc_amd import parallel as par
    assert par.frames_for_rank(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((par.frames_for_rank(300, r, 8) for r in range(8)), [])) == list(range(300))
    pts = pcc.synthetic.sphere_shell(64, 27.0, 0.6)
    ids, rows = par.split_blocks(pts, 32)
    assert sum(len(r) for r in rows) == pts.shape[0] and len(set(map(tuple, ids))) == len(rows)
    for bid, r in zip(ids, rows):
        assert (np.floor_divide(pts[r, :3], 32) == bid).all()
    counts = [len(r) for r in rows]
    # the device version (what compress_blocks runs; here on CPU tensors): same cubes in the same order, same membership
    import torch
    ids_d, counts_d, cube_of_point = par.split_blocks_device(torch.from_numpy(pts), 32)
    assert np.array_equal(ids_d, ids) and counts_d == counts
    for i, r in enumerate(rows):
        assert np.array_equal(np.nonzero(cube_of_point.numpy() == i)[0], np.sort(r))
    parts = par.assign_blocks(counts, 3)
    assert sorted(sum(parts, [])) == list(range(len(rows)))
    loads = [sum(counts[b] for b in p) for p in parts]
    assert max(loads) - min(loads) <= max(counts)


def test_block_partition_at_eight_ranks(pcc):
    """BASELINE config 4 / north_star's whole-frame mode at its stated width: the cubes of one frame over 8 ranks — every cube
    to exactly one rank, loads balanced to within one cube, deterministic; with the bench's 512^3 cubes of a 1024^3 frame
    every rank gets exactly one cube, and a world larger than the cube count leaves ranks empty without failing"""
    from pcc_amd import parallel as par
    pts = pcc.synthetic.sphere_shell(128, 54.0, 0.5)                     # a small stand-in with the same topology
    for block, world in ((64, 8), (32, 8), (64, 16)):
        ids, rows = par.split_blocks(pts, block)
        counts = [len(r) for r in rows]
        parts = par.assign_blocks(counts, world)
        assert len(parts) == world and sorted(sum(parts, [])) == list(range(len(rows)))
        assert parts == par.assign_blocks(counts, world)
        loads = [sum(counts[b] for b in p) for p in parts]
        assert max(loads) - min(l for l in loads if l or len(rows) >= world) <= max(counts)
        if block == 64 and world == 8:
            assert len(rows) == 8 and all(len(p) == 1 for p in parts)      # 2 x 2 x 2 cubes: one per rank
        if world > len(rows):
            assert sum(1 for p in parts if not p) == world - len(rows)
