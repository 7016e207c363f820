"""D1 / Y-PSNR metrics (PointCloudMetric, /root/reference/metrics/metric.py:6-189) and Bjontegaard
deltas (metrics/bjontegaard.py:6-79): GPU voxel-hash association against the CPU KD-tree oracle."""
import numpy as np
import pytest
import torch

from oracle import metrics as om


def make_pair(rng, grid=48, n=6000, jitter=2, drop=0.1):
    """a voxelised source and a perturbed, partly dropped 'reconstruction' with different colours"""
    g = np.stack(np.meshgrid(*[np.arange(grid)] * 3, indexing="ij"), -1).reshape(-1, 3)
    keep = np.abs(np.linalg.norm(g - (grid - 1) / 2, axis=1) - grid * 0.35) < 0.8
    src = g[keep][:n]
    col = rng.integers(0, 256, size=(src.shape[0], 3)) / 255.0
    rec = src + rng.integers(-jitter, jitter + 1, size=src.shape)
    rec = rec[rng.random(rec.shape[0]) > drop]
    rec = np.unique(np.clip(rec, 0, grid - 1), axis=0)
    rcol = rng.random((rec.shape[0], 3))
    return np.concatenate([src, col], 1).astype(np.float64), np.concatenate([rec, rcol], 1).astype(np.float64)


def test_bjontegaard_identities_and_known_shift(pcc):
    from pcc_amd.metrics import Bjontegaard_Delta, Bjontegaard_Model
    r = np.array([0.1, 0.25, 0.6, 1.4, 3.0])
    d = 30 + 6 * np.log10(r) + 0.8 * np.log10(r) ** 2
    m1 = Bjontegaard_Model(r, d)
    assert abs(Bjontegaard_Delta().compute_BD_PSNR(m1, m1)) < 1e-12
    assert abs(Bjontegaard_Delta().compute_BD_Rate(m1, m1)) < 1e-12
    m2 = Bjontegaard_Model(r, d + 1.5)                         # the same curve 1.5 dB higher
    assert abs(Bjontegaard_Delta().compute_BD_PSNR(m1, m2) - 1.5) < 1e-9
    m3 = Bjontegaard_Model(r * 0.8, d)                         # the same quality at 20 % less rate
    assert abs(Bjontegaard_Delta().compute_BD_Rate(m1, m3) - (-0.2)) < 1e-6
    assert np.allclose(m1.evaluate(r), d, atol=1e-9)


def test_bjontegaard_matches_the_reference_module(pcc):
    """tests/golden/bjontegaard_ref.json was produced by the reference's own metrics/bjontegaard.py on RD rows of
    its results/Ours/test.csv (tests/golden/make_bd_golden.py, build container only): the one reference-generated
    fixture of this repository.  It pins the BD half of SURVEY.md 8f rank 3 — not the codec."""
    import json
    import os
    from pcc_amd.metrics import Bjontegaard_Delta, Bjontegaard_Model
    with open(os.path.join(os.path.dirname(__file__), "golden", "bjontegaard_ref.json")) as f:
        gold = json.load(f)
    models = {}
    for key, want in gold["models"].items():
        name, metric = key.split("/")
        c = gold["curves"][name]
        m = models[key] = Bjontegaard_Model(c["bpp"], c[metric])
        assert np.allclose(m.parameters_PSNR, want["parameters_PSNR"], rtol=1e-9, atol=1e-9), key
        assert np.allclose(m.parameters_Rate, want["parameters_Rate"], rtol=1e-9, atol=1e-12), key
        assert np.allclose([m.evaluate(r) for r in want["probe_rates"]], want["evaluate"], rtol=0, atol=1e-9), key
        assert np.allclose([m.evaluate_rate(d) for d in want["probe_psnr"]], want["evaluate_rate"], rtol=0, atol=1e-9), key
        _, _, xs, ys = m.get_plot_data()
        assert len(xs) == 100 and len(ys) == 100
        assert np.allclose([xs[0], xs[-1]], want["plot_x_first_last"], atol=1e-12)
        assert np.allclose([ys[0], ys[-1]], want["plot_y_first_last"], atol=1e-9)
    bd = Bjontegaard_Delta()
    assert len(gold["deltas"]) == 12
    for d in gold["deltas"]:
        m1, m2 = models[f"{d['model1']}/{d['metric']}"], models[f"{d['model2']}/{d['metric']}"]
        assert bd.compute_BD_PSNR(m1, m2) == pytest.approx(d["BD_PSNR"], abs=1e-9), d
        assert bd.compute_BD_Rate(m1, m2) == pytest.approx(d["BD_Rate"], abs=1e-9), d
    ex = gold["module_example"]                      # the example in the reference module's __main__
    m1, m2 = Bjontegaard_Model(ex["bitrates1"], ex["d1"]), Bjontegaard_Model(ex["bitrates2"], ex["d1"])
    assert bd.compute_BD_PSNR(m1, m2) == pytest.approx(ex["BD_PSNR_12"], abs=1e-9)
    assert bd.compute_BD_Rate(m1, m2) == pytest.approx(ex["BD_Rate_12"], abs=1e-9)
    assert bd.compute_BD_PSNR(m2, m1) == pytest.approx(ex["BD_PSNR_21"], abs=1e-9)
    assert bd.compute_BD_Rate(m2, m1) == pytest.approx(ex["BD_Rate_21"], abs=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,jitter", [(0, 0), (1, 1), (2, 3)])
def test_metrics_match_oracle_both_modes(pcc, seed, jitter):
    from pcc_amd.metrics import PointCloudMetric
    rng = np.random.default_rng(seed)
    src, rec = make_pair(rng, jitter=jitter)
    met = PointCloudMetric(torch.from_numpy(src).to("cuda:0"), torch.from_numpy(rec).to("cuda:0"), resolution=1023)
    for skip_avg in (True, False):
        got, _ = met.compute_pointcloud_metrics(drop_duplicates=skip_avg)
        want = om.pc_metrics(src, rec, 1023, average_ties=not skip_avg)
        for key in ("AB_mse", "BA_mse", "AB_hausdorff", "BA_hausdorff"):
            assert got[key] == pytest.approx(want[key], rel=1e-12), key           # integer sums: exact
        for key in ("sym_psnr_mse", "sym_psnr_hausdorff", "AB_y_psnr", "BA_y_psnr", "sym_y_psnr", "sym_u_psnr", "sym_v_psnr"):
            assert got[key] == pytest.approx(want[key], abs=1e-3), (key, skip_avg)   # dB; f32 mean order differs


@pytest.mark.gpu
def test_nearest_neighbour_association_is_exact(pcc):
    from pcc_amd.metrics import nearest_neighbours
    from pcc_amd import CoordMap
    rng = np.random.default_rng(5)
    tgt = np.unique(rng.integers(0, 300, size=(4000, 3)), axis=0)          # sparse: neighbours tens of voxels away
    qry = rng.integers(-20, 320, size=(3000, 3))
    c = lambda a: torch.from_numpy(np.concatenate([np.zeros((a.shape[0], 1), np.int64), a], 1).astype(np.int32)).to("cuda:0")
    rgb = torch.from_numpy(rng.random((tgt.shape[0], 3))).to("cuda:0")
    idx, d2, ties, tsum = nearest_neighbours(c(qry), CoordMap(c(tgt), 1, nbatch=1), rgb)
    idx, d2, ties = idx.cpu().numpy(), d2.cpu().numpy(), ties.cpu().numpy()
    all_d2 = ((qry[:, None, :].astype(np.int64) - tgt[None, :, :]) ** 2).sum(-1)
    assert np.array_equal(d2, all_d2.min(axis=1))
    assert np.array_equal(ties, (all_d2 == all_d2.min(axis=1, keepdims=True)).sum(axis=1))
    key = (tgt[:, 0].astype(np.int64) << 42) + (tgt[:, 1].astype(np.int64) << 21) + tgt[:, 2]
    want = np.where(all_d2 == all_d2.min(axis=1, keepdims=True), key[None, :], np.iinfo(np.int64).max).argmin(axis=1)
    assert np.array_equal(idx, want)
    want_sum = (rgb.cpu().numpy()[None, :, :] * (all_d2 == all_d2.min(axis=1, keepdims=True))[:, :, None]).sum(axis=1)
    assert np.allclose(tsum.cpu().numpy(), want_sum, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_identical_clouds_and_input_checks(pcc):
    from pcc_amd.metrics import PointCloudMetric
    rng = np.random.default_rng(9)
    src, _ = make_pair(rng)
    t = torch.from_numpy(src).to("cuda:0")
    res, _ = PointCloudMetric(t, t.clone()).compute_pointcloud_metrics(drop_duplicates=True)
    assert res["sym_mse"] == 0 and res["sym_psnr_mse"] == float("inf") and res["sym_y_psnr"] == float("inf")
    dup = torch.cat([t, t[:10]], dim=0)                                    # duplicated points are dropped (first wins)
    res2, _ = PointCloudMetric(dup, t).compute_pointcloud_metrics(drop_duplicates=True)
    assert res2["sym_mse"] == 0
    bad = t.clone()
    bad[0, 0] += 0.5
    with pytest.raises(ValueError):
        PointCloudMetric(bad, t)


@pytest.mark.gpu
def test_codec_output_metrics_match_oracle(pcc):
    """the use the reference makes of it (train.py:263-264): source frame vs. decoded frame"""
    from pcc_amd import synthetic as syn
    from pcc_amd.metrics import PointCloudMetric
    model = syn.make_model(seed=0, device="cuda:0")
    model.update()
    pts = syn.sphere_shell(**syn.CONFIG1)
    qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)
    import pcc_amd
    Q = pcc_amd.SparseTensor(coordinates=torch.from_numpy(qc).to("cuda:0"), features=torch.from_numpy(qf).to("cuda:0"), device="cuda:0")
    x = torch.from_numpy(pts).to("cuda:0")
    strings, shape, k, coords = model.compress(x, Q)
    rec = model.decompress(coordinates=coords, strings=strings, shape=shape, k=k)
    got, _ = PointCloudMetric(x, rec, resolution=31).compute_pointcloud_metrics(drop_duplicates=True)
    want = om.pc_metrics(pts.astype(np.float64), rec.cpu().numpy().astype(np.float64), 31)
    assert got["sym_psnr_mse"] == pytest.approx(want["sym_psnr_mse"], abs=1e-6)
    assert got["sym_y_psnr"] == pytest.approx(want["sym_y_psnr"], abs=1e-3)
