"""Latent-coordinate coder "PCO1" (file mode; stands where the reference calls tmc3,
model/model.py:318-395): CPU twin self-checks without a GPU, GPU == CPU byte equality with one."""
import struct

import numpy as np
import pytest
import torch

from oracle import octree as oo


def shell(grid, radius, thick, stride=1, origin=(0, 0, 0)):
    g = np.stack(np.meshgrid(*[np.arange(grid)] * 3, indexing="ij"), -1).reshape(-1, 3)
    keep = np.abs(np.linalg.norm(g - (grid - 1) / 2, axis=1) - radius) < thick
    return g[keep].astype(np.int64) * stride + np.asarray(origin, dtype=np.int64)


def as_set(a):
    return sorted(map(tuple, np.asarray(a).tolist()))


# ---------------------------------------------------------------------------------------------- CPU
def test_two_point_stream_by_hand():
    """g = (0,0,0), (1,0,0): depth 1, the root's children are 0 and 4 (x is the top bit) -> byte 0x11"""
    pts = np.array([[16, 8, 24], [24, 8, 24]])
    data = oo.encode_points(pts, 8)
    assert data[:4] == b"PCO1"
    depth, _, _, stride, ox, oy, oz, n = struct.unpack("<BBHi3iI", data[4:28])
    assert (depth, stride, ox, oy, oz, n) == (1, 8, 16, 8, 24, 2)
    assert struct.unpack("<I", data[28:32]) == (1,) and data[32] == 0           # one root node, flat table
    origin, _, _, _, levels = oo.unpack_stream(data)
    assert [lv.tolist() for lv in levels] == [[0x11]]
    assert as_set(oo.decode_points(data)) == as_set(pts)


def test_morton_order_and_inverse():
    rng = np.random.default_rng(0)
    g = rng.integers(0, 1 << 9, size=(500, 3))
    keys = oo.morton_keys(g, 9)
    assert np.array_equal(oo.keys_to_grid(keys, 9), g)
    a, b = np.array([[1, 0, 0]]), np.array([[0, 1, 1]])
    assert oo.morton_keys(a, 1)[0] == 4 and oo.morton_keys(b, 1)[0] == 3


@pytest.mark.parametrize("n", [0, 1, 2, 3, 63, 64, 65, 1000])
def test_oracle_round_trip_small(n):
    rng = np.random.default_rng(n)
    pts = np.unique(rng.integers(-40, 40, size=(n * 2 + 2, 3)) * 8, axis=0)[:n]
    data = oo.encode_points(pts, 8)
    assert as_set(oo.decode_points(data)) == as_set(pts)


def test_oracle_round_trip_shell_and_rate():
    pts = shell(128, 32.5, 0.7, stride=8, origin=(16, 8, 24))
    data = oo.encode_points(pts, 8)
    assert as_set(oo.decode_points(data)) == as_set(pts)
    assert len(data) * 8 / pts.shape[0] < 3.5            # ~2.8 bits per latent point on a surface


def test_oracle_rejects_bad_input():
    with pytest.raises(ValueError):
        oo.encode_points(np.array([[0, 0, 0], [0, 0, 0]]), 8)
    with pytest.raises(ValueError):
        oo.encode_points(np.array([[0, 0, 0], [4, 0, 0]]), 8)
    with pytest.raises(ValueError):
        oo.unpack_stream(b"XXXX" + bytes(40))


# ---------------------------------------------------------------------------------------------- GPU
def _gpu_encode(pcc, pts, stride):
    from pcc_amd import octree
    c = np.concatenate([np.zeros((pts.shape[0], 1), np.int64), pts], axis=1).astype(np.int32)
    return octree.encode_coordinates(torch.from_numpy(c).to("cuda:0"), stride)


CASES = {
    "shell8": lambda: (shell(128, 32.5, 0.7, stride=8, origin=(16, 8, 24)), 8),
    "one": lambda: (np.array([[40, 48, 56]]), 8),
    "two": lambda: (np.array([[16, 8, 24], [24, 8, 24]]), 8),
    "line": lambda: (np.stack([np.arange(70) * 8, np.zeros(70, int), np.full(70, 8)], axis=1), 8),
    "random": lambda: (np.unique(np.random.default_rng(1).integers(-300, 300, size=(5000, 3)) * 2, axis=0), 2),
    "deep": lambda: (np.unique(np.random.default_rng(2).integers(0, 1 << 20, size=(3000, 3)), axis=0), 1),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_gpu_stream_equals_oracle_stream_and_round_trips(pcc, case):
    from pcc_amd import octree
    pts, stride = CASES[case]()
    perm = np.random.default_rng(3).permutation(pts.shape[0])      # input order must not matter
    data = _gpu_encode(pcc, pts[perm], stride)
    assert data == oo.encode_points(pts, stride)
    back = octree.decode_coordinates(data, "cuda:0", batch=0).cpu().numpy()
    assert (back[:, 0] == 0).all()
    assert np.array_equal(back[:, 1:], oo.decode_points(data))     # same (Morton) order as the CPU twin
    assert as_set(back[:, 1:]) == as_set(pts)


@pytest.mark.gpu
def test_gpu_empty_and_errors(pcc):
    from pcc_amd import octree
    empty = octree.encode_coordinates(torch.zeros((0, 4), dtype=torch.int32, device="cuda:0"), 8)
    assert empty == oo.encode_points(np.zeros((0, 3), int), 8)
    assert octree.decode_coordinates(empty, "cuda:0").shape == (0, 4)
    with pytest.raises(ValueError):
        _gpu_encode(pcc, np.array([[0, 0, 0], [8, 0, 0], [8, 0, 0]]), 8)          # duplicate
    with pytest.raises(ValueError):
        _gpu_encode(pcc, np.array([[0, 0, 0], [12, 0, 0]]), 8)                    # off the lattice
    with pytest.raises(RuntimeError):
        octree.encode_coordinates(torch.zeros((1, 4), dtype=torch.int32), 8)      # host tensor
    good = _gpu_encode(pcc, shell(64, 20.0, 0.7, stride=8), 8)
    for bad in (b"XXXX" + good[4:], good[:40], good[:-7]):
        with pytest.raises((ValueError, RuntimeError)):
            octree.decode_coordinates(bad, "cuda:0")
    # a consistent header with a lying level size is caught before any kernel runs
    depth = good[4]
    lie = bytearray(good)
    lie[28 + 4 * (depth - 1):28 + 4 * depth] = struct.pack("<I", struct.unpack("<I", good[28 + 4 * (depth - 1):28 + 4 * depth])[0] - 1)
    with pytest.raises((ValueError, RuntimeError)):
        octree.decode_coordinates(bytes(lie), "cuda:0")


@pytest.mark.gpu
def test_gpu_full_size_round_trip(pcc):
    """config-2 geometry itself (850,824 voxels, depth 10): too slow for the python twin, checked by
    the round-trip property and by the rate"""
    from pcc_amd import octree, synthetic as syn
    pts = syn.sphere_shell(**syn.CONFIG2)
    xyz = np.asarray(pts)[:, :3].astype(np.int64)
    data = _gpu_encode(pcc, xyz, 1)
    back = octree.decode_coordinates(data, "cuda:0").cpu().numpy()[:, 1:]
    key = lambda a: (a[:, 0].astype(np.int64) << 40) | (a[:, 1].astype(np.int64) << 20) | a[:, 2].astype(np.int64)
    assert np.array_equal(np.sort(key(back)), np.sort(key(xyz)))
    assert len(data) * 8 / xyz.shape[0] < 3.0
