"""Data front-end and harness: PLY IO, q-map builders (data/q_map.py:143-291), the evaluation
harness (utils.py:418-472)."""
import random

import numpy as np
import pytest
import torch


def cloud(n=500, seed=0):
    rng = np.random.default_rng(seed)
    xyz = np.unique(rng.integers(0, 64, size=(n, 3)), axis=0).astype(np.float32)
    rgb = rng.integers(0, 256, size=(xyz.shape[0], 3)).astype(np.float32) / 255.0
    return np.concatenate([xyz, rgb], axis=1)


@pytest.mark.parametrize("binary", [True, False])
def test_ply_round_trip(pcc, tmp_path, binary):
    from pcc_amd import io
    c = cloud()
    p = str(tmp_path / "c.ply")
    io.write_ply(p, c, binary=binary)
    back = io.read_ply(p)
    assert back.dtype == np.float32 and back.shape == c.shape
    assert np.array_equal(back[:, :3], c[:, :3]) and np.allclose(back[:, 3:], c[:, 3:], atol=1e-7)


def test_ply_reads_foreign_layouts(pcc, tmp_path):
    """8iVFB style: extra properties, doubles, comments; geometry-only files"""
    from pcc_amd import io
    p = tmp_path / "a.ply"
    p.write_text("ply\nformat ascii 1.0\ncomment made elsewhere\nelement vertex 2\nproperty double x\nproperty double y\n"
                 "property double z\nproperty float nx\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
                 "end_header\n1 2 3 0.5 255 0 51\n4 5 6 0.5 0 102 255\n")
    got = io.read_ply(str(p))
    assert np.allclose(got, [[1, 2, 3, 1, 0, 0.2], [4, 5, 6, 0, 0.4, 1]])
    rec = np.zeros(2, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("alpha", "u1")])
    rec["x"], rec["y"], rec["z"] = [7, 8], [9, 10], [11, 12]
    q = tmp_path / "b.ply"
    q.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\n"
                  b"property uchar alpha\nend_header\n" + rec.tobytes())
    got = io.read_ply(str(q))
    assert np.array_equal(got[:, :3], [[7, 9, 11], [8, 10, 12]]) and (got[:, 3:] == 0).all()
    bad = tmp_path / "c.ply"
    bad.write_text("ply\nformat binary_big_endian 1.0\nelement vertex 0\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
    with pytest.raises(ValueError):
        io.read_ply(str(bad))
    with pytest.raises(ValueError):
        (tmp_path / "d.ply").write_text("not a ply\n")
        io.read_ply(str(tmp_path / "d.ply"))


@pytest.mark.gpu
def test_q_map_generator_and_lambda_scaling(pcc):
    from pcc_amd import q_map as qm
    c = cloud(800)
    c2 = cloud(600, seed=1)
    coords = np.concatenate([np.concatenate([np.zeros((c.shape[0], 1)), c[:, :3]], 1), np.concatenate([np.ones((c2.shape[0], 1)), c2[:, :3]], 1)])
    geom = pcc.SparseTensor(coordinates=torch.from_numpy(coords).to("cuda:0"), features=torch.ones(coords.shape[0], 1, device="cuda:0"),
                            device="cuda:0")
    cfg = {"mode": "exponential", "lambda_A_max": 12800, "lambda_A_min": 100, "lambda_G_max": 1600, "lambda_G_min": 25}
    gen = qm.Q_Map(cfg)
    random.seed(4)
    q, lam = gen(geom)
    # replay the draws the reference's generator would make with this seed (data/q_map.py:208-266)
    random.seed(4)
    want = torch.zeros(coords.shape[0], 2)
    C = q.C.cpu()
    for b in (0, 1):
        m = C[:, 0] == b
        if random.choice(range(2)) == 0:
            d = random.randint(1, 3)
            v = C[m][:, d].float()
            want[m] = torch.clamp((v - v.min()) / (v.max() - v.min() + 1e-10), 0, 1).unsqueeze(1).repeat(1, 2)
        else:
            sg, sa = random.uniform(0, 1), random.uniform(0, 1)
            want[m] = torch.tensor([sg, sa])
    assert torch.allclose(q.F.cpu(), want, atol=1e-7)
    import math
    aG, bG = math.log2(1600 + 25), 25 - 1
    assert torch.allclose(lam.F[:, 0].cpu(), 2 ** (want[:, 0] * aG) + bG, rtol=1e-5)
    quad = qm.Q_Map(dict(cfg, mode="quadratic"))
    lam2 = quad.scale_q_map(q)
    assert torch.allclose(lam2.F[:, 1].cpu(), want[:, 1] ** 2 * (12800 - 100) + 100, rtol=1e-5)
    g = qm.gradient_map(geom.map, 2, 0.1, 0.9)
    assert float(g.F.min()) == pytest.approx(0.1, abs=1e-6) and float(g.F.max()) == pytest.approx(0.9, abs=1e-6)
    v = qm.view_dependent_map(geom.map, 0.4, 0.8, 1, 10, 50)
    x1 = geom.C[:, 1].float().cpu()
    assert torch.allclose(v.F[:, 1].cpu(), 0.8 * torch.clamp((x1 - 10) / 40, 0, 1), atol=1e-6)
    r = qm.roi_map(geom.map, 0.4, 0.8, 3, 32)
    z = geom.C[:, 3].cpu()
    assert torch.equal(r.F[:, 0].cpu() > 0, z >= 32) and float(r.F[:, 0].max()) == pytest.approx(0.4)
    with pytest.raises(ValueError):
        qm.Q_Map({"mode": "linear"})


@pytest.mark.gpu
def test_harness_file_mode_row(pcc, tmp_path):
    """utils.py:418-472 on the config-1 frame: file on disk, bpp from its size, metrics consistent with
    the in-memory path"""
    from pcc_amd import synthetic as syn
    from pcc_amd.harness import evaluate_frame, compress_model_ours
    model = syn.make_model(seed=0, device="cuda:0")
    model.update()
    pts = syn.sphere_shell(**syn.CONFIG1)
    data = {"src": {"points": torch.from_numpy(pts[None, :, :3]), "colors": torch.from_numpy(pts[None, :, 3:])}}
    row = evaluate_frame("exp", model, data, 0.5, 0.5, "cuda:0", str(tmp_path), resolution=31)
    assert row["n_source"] == pts.shape[0] and row["n_decoded"] == pts.shape[0]
    src, rec, bpp, tc, td = compress_model_ours("exp", model, data, 0.5, 0.5, "cuda:0", str(tmp_path))
    assert bpp == row["bpp"] and tc > 0 and td > 0
    import os
    assert os.path.getsize(os.path.join(str(tmp_path), "exp", "tmp", "bitstream.bin")) * 8 / pts.shape[0] == bpp
    Q = pcc.SparseTensor(coordinates=torch.cat([torch.zeros(pts.shape[0], 1), torch.from_numpy(pts[:, :3])], 1).to("cuda:0"),
                         features=torch.full((pts.shape[0], 2), 0.5, device="cuda:0"), device="cuda:0")
    strings, shape, k, coords = model.compress(torch.from_numpy(pts).to("cuda:0"), Q)
    mem = model.decompress(coordinates=coords, strings=strings, shape=shape, k=k)
    key = lambda t: (t[:, 0].long() << 40) | (t[:, 1].long() << 20) | t[:, 2].long()
    a, b = rec[torch.argsort(key(rec))], mem[torch.argsort(key(mem))]
    assert torch.equal(a, b)                                   # file mode decodes exactly what memory mode decodes
    # per-point q arrays are accepted like scalars (utils.py:442-445)
    qa = np.full((pts.shape[0], 1), 0.5, dtype=np.float32)
    row2 = evaluate_frame("exp2", model, data, qa, qa, "cuda:0", str(tmp_path), resolution=31)
    assert row2["bpp"] == row["bpp"]


def test_sparse_collate(pcc):
    from pcc_amd.utils import sparse_collate
    a, b = cloud(50, 1), cloud(70, 2)
    C, F = sparse_collate([torch.from_numpy(a[:, :3]), torch.from_numpy(b[:, :3])], [torch.from_numpy(a[:, 3:]), torch.from_numpy(b[:, 3:])])
    assert C.dtype == torch.int32 and C.shape == (a.shape[0] + b.shape[0], 4) and F.shape == (C.shape[0], 3)
    assert (C[:a.shape[0], 0] == 0).all() and (C[a.shape[0]:, 0] == 1).all()
    assert torch.equal(C[a.shape[0]:, 1:].float(), torch.from_numpy(b[:, :3]))
    with pytest.raises(ValueError):
        sparse_collate([torch.zeros(3, 3)], [torch.zeros(2, 3)])


def test_prefetcher_keeps_order_and_surfaces_errors(pcc):
    """batch pipeline of the training tools: same sequence as a plain loop, exceptions reach the consumer"""
    import random
    import torch
    from pcc_amd.utils import Prefetcher
    rng = random.Random(7)
    want = [random.Random(7).sample(range(100), 3)]
    r2 = random.Random(7)
    want = [r2.sample(range(100), 3) for _ in range(6)]
    feed = Prefetcher(lambda: (torch.tensor(rng.sample(range(100), 3)), "tag"), depth=2, pin=False)
    got = [next(feed) for _ in range(6)]
    feed.close()
    assert [g[0].tolist() for g in got] == want and all(g[1] == "tag" for g in got)

    calls = {"n": 0}

    def bad():
        calls["n"] += 1
        if calls["n"] == 3:
            raise ValueError("broken sample")
        return calls["n"]

    feed = Prefetcher(bad, depth=1, pin=False)
    assert next(feed) == 1 and next(feed) == 2
    with pytest.raises(ValueError):
        next(feed)
    feed.close()
