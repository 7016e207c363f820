"""Differential test on irregular geometry: the synthetic shells of the other tests are smooth closed surfaces; here the
clouds are scattered voxels, filled blocks, planes, lines and far-apart clusters at several densities (seeded), each through
the full codec on the GPU and on the CPU oracle under the stage-by-stage rule of tests/_parity.py.  Such inputs put
isolated voxels, fully occupied 27-neighbourhoods, empty strided levels and coordinates on cube faces through the hash
tables, kernel maps, top-k and the entropy coder."""
import numpy as np
import pytest
import torch

from _parity import compare_codec

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _colors(rng, n):
    return (rng.integers(0, 256, (n, 3)) / 255.0).astype(np.float32)


def _cloud(kind, rng):
    if kind == "scattered":                      # isolated voxels: almost no neighbours at stride 1
        xyz = rng.integers(0, 200, (1500, 3))
    elif kind == "filled_block":                 # every 27-neighbourhood complete inside the block
        g = np.stack(np.meshgrid(np.arange(14), np.arange(14), np.arange(14), indexing="ij"), -1).reshape(-1, 3)
        xyz = g + np.array([37, 5, 101])
    elif kind == "plane":                        # one voxel thick, axis-aligned, on a cube face of the stride-8 grid
        a, b = np.meshgrid(np.arange(60), np.arange(45), indexing="ij")
        xyz = np.stack([a.ravel() + 16, np.full(a.size, 64), b.ravel() + 8], axis=1)
    elif kind == "lines":                        # three crossing lines along the axes
        t = np.arange(90)
        z = np.zeros_like(t)
        xyz = np.concatenate([np.stack([t + 10, z + 50, z + 50], 1), np.stack([z + 50, t + 10, z + 50], 1),
                              np.stack([z + 50, z + 50, t + 10], 1)])
    elif kind == "two_clusters":                 # dense blobs 900 voxels apart: separate latents, shared top-k budget
        a = rng.normal(0, 3.0, (1200, 3)).round().astype(int) + 30
        b = rng.normal(0, 5.0, (1800, 3)).round().astype(int) + 960
        xyz = np.concatenate([a, b])
    elif kind == "noisy_shell":                  # a thick, noisy shell: varied neighbour counts
        p = rng.normal(0, 1, (6000, 3))
        p = p / np.linalg.norm(p, axis=1, keepdims=True) * (20 + rng.normal(0, 1.5, (6000, 1)))
        xyz = p.round().astype(int) + 40
    else:
        raise ValueError(kind)
    xyz = np.unique(np.clip(xyz, 0, 1023), axis=0)
    return np.concatenate([xyz.astype(np.float32), _colors(rng, xyz.shape[0])], axis=1)


@pytest.fixture(scope="module")
def model(pcc):
    m = pcc.synthetic.make_model(0, DEV)
    m.update()
    return m


@pytest.mark.parametrize("kind", ["scattered", "filled_block", "plane", "lines", "two_clusters", "noisy_shell"])
def test_irregular_clouds_vs_oracle(pcc, model, oracle_codec, kind):
    rng = np.random.default_rng(sum(map(ord, kind)))
    pts = _cloud(kind, rng)
    N = pts.shape[0]
    coords = np.concatenate([np.zeros((N, 1), np.float32), pts[:, :3]], axis=1)
    for q_g, q_a in ((0.5, 0.5), (0.05, 0.9)):
        qf = np.tile(np.array([[q_g, q_a]], np.float32), (N, 1))
        r = compare_codec(pcc, model, oracle_codec, pts, coords, qf, (kind, q_g, q_a), DEV)
        assert r["bpp"] > 0
    # a per-point random q-map on the same geometry
    qf = rng.random((N, 2)).astype(np.float32)
    compare_codec(pcc, model, oracle_codec, pts, coords, qf, (kind, "random q"), DEV)


def test_ragged_batch_of_irregular_items_vs_oracle(pcc, model, oracle_codec):
    """the reference's batch mechanism (one compress call, per-item k and top-k, one stream pair) on items of very different
    size and shape that overlap in (x, y, z): 1,500 scattered voxels, 270 voxels on three lines, ONE voxel, a filled block"""
    from oracle.codec import count_bits
    from oracle.metrics import pc_metrics
    from _parity import assert_exact, assert_psnr_parity, voxel_flips
    rng = np.random.default_rng(11)
    items = [_cloud("scattered", rng), _cloud("lines", rng), np.array([[50, 50, 50, 0.3, 0.6, 0.9]], np.float32),
             _cloud("filled_block", rng)]
    pts = np.concatenate(items)
    item = np.concatenate([np.full(len(p), i) for i, p in enumerate(items)])
    N = pts.shape[0]
    qf = rng.random((N, 2)).astype(np.float32)
    qc = np.concatenate([item.reshape(-1, 1).astype(np.float32), pts[:, :3]], axis=1)
    x = torch.from_numpy(pts).to(DEV)
    Q = pcc.SparseTensor(coordinates=torch.from_numpy(qc).to(DEV), features=torch.from_numpy(qf).to(DEV), device=DEV)
    strings, shape, k, coords = model.compress(x, Q, batch=torch.from_numpy(item).to(DEV))
    o_strings, o_shape, o_k, o_coords = oracle_codec.compress(pts, qc, qf, batch=item)
    assert shape == o_shape and k == o_k and [len(stage) for stage in k] == [4, 4, 4]
    assert k[2] == [len(p) for p in items]                                     # the finest stage keeps every item's own count
    assert set(map(tuple, coords.cpu().numpy().tolist())) == set(map(tuple, o_coords.tolist()))
    bits, o_bits = count_bits(strings), count_bits(o_strings)
    assert abs(bits - o_bits) <= 3e-3 * o_bits + 64
    rec, rec_item = model.decompress(coordinates=coords, strings=strings, shape=shape, k=k, return_batch=True)
    rec, rec_item = rec.cpu().numpy(), rec_item.cpu().numpy()
    o_rec = oracle_codec.decompress(o_coords, o_strings, o_shape, o_k)
    assert rec.shape == o_rec.shape == (N, 6)
    for i, p in enumerate(items):
        assert int((rec_item == i).sum()) == len(p)
    o_item = oracle_codec.last_batch
    flips = 0
    for i, p in enumerate(items):                                              # per item: the same voxels, the same quality
        a, b = rec[rec_item == i], o_rec[o_item == i]
        f = voxel_flips(a, b)
        flips += f
        if len(p) > 1:
            assert_psnr_parity(pc_metrics(p, a), pc_metrics(p, b), f, len(p), ("item", i))
    assert flips <= max(8, int(5e-3 * N)), flips
    # and byte for byte against the kernel-order oracle on the same items
    assert_exact(oracle_codec, pts, qc, qf, strings, shape, k, coords.cpu().numpy(), rec, "ragged batch", batch=item, rec_item=rec_item)
