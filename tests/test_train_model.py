"""Training path, model level: ColorModel.forward in training mode + the
configured losses, differentiated through the HIP kernels, against the CPU oracle differentiated by
torch autograd with the same noise draws (reference: train.py:171-221, model/model.py:51-93,
loss.py:67-195)."""
import numpy as np
import pytest
import torch

from oracle import train as ot
from oracle.codec import Codec
from oracle.nn import SparseTensor as OSparseTensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def coord_noise(shape, coords):
    """U(-0.5, 0.5)-like noise that is a function of (channel, coordinate), not of the row position"""
    c = np.asarray(coords, dtype=np.float64)
    n = c.shape[0]
    ch = shape[0] if shape[0] != 1 else shape[1]
    phase = c[:, 1] * 12.9898 + c[:, 2] * 78.233 + c[:, 3] * 37.719
    val = np.sin(phase[None, :] * (1.0 + 0.37 * np.arange(ch)[:, None])) * 43758.5453
    val = (val - np.floor(val) - 0.5).astype(np.float32)            # [ch, n]
    assert val.shape == (ch, n)
    return torch.from_numpy(val.reshape(shape))


def coord_noise_b(shape, coords):
    """like coord_noise, with the batch index in the key (two items may share x, y, z)"""
    c = np.asarray(coords, dtype=np.float64).copy()
    c[:, 1] += 1000.0 * c[:, 0]
    return coord_noise(shape, c)


def _setup(pcc, geometry="sphere"):
    from pcc_amd import synthetic as syn
    # "two_hyperprior": the variant model/model.py:22-24 builds from an "entropy_model_map" section (likelihood LISTS, a second
    # hyperprior on the stride-8 q-map with 3-channel layers), on the sphere
    model = syn.make_model(seed=0, device=DEV, config=syn.TWO_HYPERPRIOR_CONFIG if geometry == "two_hyperprior" else None)
    if geometry in ("sphere", "two_hyperprior"):
        pts = syn.sphere_shell(**syn.CONFIG1)
    else:
        # irregular: a thick noisy shell plus scattered voxels (isolated rows, varied neighbour counts, sparse strided levels)
        rng = np.random.default_rng(3)
        p = rng.normal(0, 1, (5000, 3))
        p = p / np.linalg.norm(p, axis=1, keepdims=True) * (18 + rng.normal(0, 1.5, (5000, 1)))
        xyz = np.unique(np.clip(np.concatenate([p.round().astype(int) + 32, rng.integers(0, 64, (300, 3))]), 0, 63), axis=0)
        rgb = (rng.integers(0, 256, (xyz.shape[0], 3)) / 255.0).astype(np.float32)
        pts = np.concatenate([xyz.astype(np.float32), rgb], axis=1)
    qc, qf = syn.uniform_qmap(pts[:, :3], 0.3, 0.7)
    qf = qf.copy()
    qf[:, 0] = 0.2 + 0.6 * (pts[:, 0] - pts[:, 0].min()) / (pts[:, 0].max() - pts[:, 0].min())     # a gradient map: pooling matters
    lam = np.stack([2 ** (qf[:, 0] * 6) + 24, 2 ** (qf[:, 1] * 7) + 99], axis=1).astype(np.float32)
    return model, pts, qc, qf, lam


@pytest.mark.parametrize("geometry", ["sphere", "irregular", "two_hyperprior"])
def test_training_step_matches_oracle_autograd(pcc, geometry):
    from pcc_amd import entropy as pe
    from pcc_amd.loss import OURS_LOSS, Loss
    model, pts, qc, qf, lam = _setup(pcc, geometry)
    model.train()
    coords = torch.from_numpy(qc).to(DEV)
    inp = pcc.SparseTensor(coordinates=coords, features=torch.from_numpy(pts[:, 3:]).to(DEV), device=DEV)
    Q = pcc.SparseTensor(torch.from_numpy(qf).to(DEV), coordinate_map=inp.map)
    Lam = pcc.SparseTensor(torch.from_numpy(lam).to(DEV), coordinate_map=inp.map)
    from oracle import nn as on
    from pcc_amd import sparse as sp
    pe.NOISE_SOURCE = coord_noise
    sp.GATE_LOG = []
    try:
        out = model(inp, Q, Lam)
    finally:
        pe.NOISE_SOURCE = None
        gate_log, sp.GATE_LOG = sp.GATE_LOG, None
    total, parts = Loss(OURS_LOSS)(inp, out)
    total.backward()
    # the HIP run's ReLU / LeakyReLU gates, by layer name, for the oracle to differentiate through (oracle/nn.py: FORCED_GATES)
    name_of = {id(m): n for n, m in model.named_modules()}
    gates = {name_of[id(mod)]: (c.cpu().numpy(), g.cpu()) for mod, c, g in gate_log}
    assert len(gates) == len(gate_log) >= 40, (len(gates), len(gate_log))        # every activated layer once

    sd = ot.leaf_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    from pcc_amd import synthetic as syn
    codec = Codec(sd, syn.TWO_HYPERPRIOR_CONFIG if geometry == "two_hyperprior" else None, leaves=True)
    on.FORCED_GATES, on.GATE_FLIPS = gates, []
    try:
        o_out = ot.forward_train(codec, qc, pts[:, 3:], qc, qf, coord_noise)
    finally:
        flips, on.FORCED_GATES, on.GATE_FLIPS = on.GATE_FLIPS, None, None
    o_total, o_parts = ot.losses(qc, pts[:, 3:], o_out, OSparseTensor(qc, torch.from_numpy(lam), 1))
    o_total.backward()
    # a gate may differ only where the pre-activation is zero to within the two implementations' rounding difference
    for tag, v, top in flips:
        assert v <= 1e-5 * max(top, 1e-30), ("a gate differs on a pre-activation that is not near zero", tag, v, top)
    print(f"{geometry}: {len(flips)} of the oracle's own gates differ from the HIP run's (forced to the HIP run's)")

    for key in o_parts:
        assert float(parts[key].detach()) == pytest.approx(float(o_parts[key].detach()), rel=2e-4), key
    assert out["prediction"].F.shape == (pts.shape[0], 3)
    for pr, opr in zip(out["occ_predictions"], o_out["occ_predictions"]):      # same candidate sets at every stage
        assert set(map(tuple, pr.C.cpu().numpy().tolist())) == set(map(tuple, opr.C.tolist()))
    assert set(map(tuple, out["prediction"].C.cpu().numpy().tolist())) == set(map(tuple, o_out["prediction"].C.tolist()))
    checked = 0
    named = dict(model.named_parameters())
    worst = worst_l2 = (0.0, None)
    for name, leaf in sd.items():
        if leaf.grad is None or name not in named:
            continue
        g = named[name].grad
        assert g is not None, name
        ref = leaf.grad
        scale = float(ref.abs().max())
        if scale == 0.0:
            assert float(g.abs().max()) == 0.0, name
            continue
        err = float((g.cpu() - ref).abs().max()) / scale
        l2 = float((g.cpu() - ref).norm() / ref.norm())
        worst = max(worst, (err, name))
        worst_l2 = max(worst_l2, (l2, name))
        checked += 1
    assert checked > 150, checked
    # Both sides differentiate through the SAME gates (the HIP run's, forced onto the oracle above — where the oracle's own
    # sign differed, the pre-activation was checked to be zero to 1e-5 of the layer's largest), so there is no gate-flip
    # allowance: every parameter gradient agrees to fp32 summation-order differences on both geometries.  (Round 2 bounded
    # the irregular cloud at 10 % of the largest element / 2 % l2 to cover a couple of flipped gates — wide enough to hide a
    # wrong term)
    assert worst[0] < 5e-3 and worst_l2[0] < 1e-3, (worst, worst_l2, len(flips))


def test_training_mode_draws_fresh_noise_and_eval_is_unchanged(pcc):
    model, pts, qc, qf, lam = _setup(pcc)
    coords = torch.from_numpy(qc).to(DEV)
    inp = pcc.SparseTensor(coordinates=coords, features=torch.from_numpy(pts[:, 3:]).to(DEV), device=DEV)
    Q = pcc.SparseTensor(torch.from_numpy(qf).to(DEV), coordinate_map=inp.map)
    model.train()
    a = model(inp, Q, None)["likelihoods"]["y"]
    b = model(inp, Q, None)["likelihoods"]["y"]
    assert not torch.equal(a, b)
    model.eval()
    with torch.no_grad():
        c = model(inp, Q, None)["likelihoods"]["y"]
        d = model(inp, Q, None)["likelihoods"]["y"]
    assert torch.equal(c, d)


def test_optimizer_step_reduces_the_loss(pcc):
    """three Adam steps on one frame (train.py:194-206 without the data loader)"""
    from pcc_amd.loss import OURS_LOSS, Loss
    model, pts, qc, qf, lam = _setup(pcc)
    model.train()
    coords = torch.from_numpy(qc).to(DEV)
    inp = pcc.SparseTensor(coordinates=coords, features=torch.from_numpy(pts[:, 3:]).to(DEV), device=DEV)
    Q = pcc.SparseTensor(torch.from_numpy(qf).to(DEV), coordinate_map=inp.map)
    Lam = pcc.SparseTensor(torch.from_numpy(lam).to(DEV), coordinate_map=inp.map)
    params = [p for n, p in model.named_parameters() if not n.endswith(".quantiles")]      # train.py:63-64
    opt = torch.optim.Adam(params, lr=1e-4)
    loss_fn = Loss(OURS_LOSS)
    values = []
    for _ in range(4):
        opt.zero_grad()
        total, _ = loss_fn(inp, model(inp, Q, Lam))
        total.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        values.append(float(total.detach()))
    assert all(np.isfinite(values)) and values[-1] < values[0]
    aux = model.aux_loss()
    aux.backward()
    assert model.entropy_model.entropy_bottleneck.quantiles.grad is not None


def test_training_step_batch_of_two_matches_oracle(pcc):
    """two clouds in one batch (train.py:185-194: sparse_collate): per-item top-k, per-item q-maps"""
    from pcc_amd import entropy as pe, synthetic as syn
    from pcc_amd.loss import OURS_LOSS, Loss
    from pcc_amd.utils import sparse_collate
    model = syn.make_model(seed=0, device=DEV)
    model.train()
    a = syn.sphere_shell(**syn.CONFIG1)
    b = syn.sphere_shell(grid=32, radius=11.0, half_width=0.8)
    C, F = sparse_collate([torch.from_numpy(a[:, :3]), torch.from_numpy(b[:, :3])], [torch.from_numpy(a[:, 3:]), torch.from_numpy(b[:, 3:])])
    qc = C.numpy().astype(np.float32)
    colors = F.numpy()
    qf = np.concatenate([np.tile([[0.3, 0.7]], (a.shape[0], 1)), np.tile([[0.8, 0.2]], (b.shape[0], 1))]).astype(np.float32)
    lam = np.stack([2 ** (qf[:, 0] * 6) + 24, 2 ** (qf[:, 1] * 7) + 99], axis=1).astype(np.float32)
    inp = pcc.SparseTensor(coordinates=C.to(DEV), features=F.to(DEV), device=DEV)
    Q = pcc.SparseTensor(torch.from_numpy(qf).to(DEV), coordinate_map=inp.map)
    Lam = pcc.SparseTensor(torch.from_numpy(lam).to(DEV), coordinate_map=inp.map)
    pe.NOISE_SOURCE = coord_noise_b
    try:
        out = model(inp, Q, Lam)
    finally:
        pe.NOISE_SOURCE = None
    total, parts = Loss(OURS_LOSS)(inp, out)
    total.backward()
    sd = ot.leaf_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    codec = Codec(sd, leaves=True)
    o_out = ot.forward_train(codec, qc, colors, qc, qf, coord_noise_b)
    o_total, o_parts = ot.losses(qc, colors, o_out, OSparseTensor(qc, torch.from_numpy(lam), 1))
    o_total.backward()
    assert o_out["k"] == [[int(v) for v in kk] for kk in o_out["k"]] and len(o_out["k"][0]) == 2      # per-item counts
    got_c = set(map(tuple, out["prediction"].C.cpu().numpy().tolist()))
    assert got_c == set(map(tuple, o_out["prediction"].C.tolist()))                    # no top-k boundary flips
    named = dict(model.named_parameters())
    worst = (0.0, None)
    errs = []
    for name, leaf in sd.items():
        if leaf.grad is None or name not in named or float(leaf.grad.abs().max()) == 0.0:
            continue
        err = float((named[name].grad.cpu() - leaf.grad).abs().max()) / float(leaf.grad.abs().max())
        worst = max(worst, (err, name))
        errs.append((err, name))
    # A pre-activation within ~1e-6 of zero can land on either side of a ReLU in two fp32 implementations; one
    # such element (seen here: one channel of one voxel in post_conv.0, |pre-activation| 2.3e-6) moves the
    # max-norm error of that layer's weight gradient to 2 % while every other gradient agrees to < 1e-3.
    # So: a tight bound on all but a few tensors, a loose one on every tensor.
    errs.sort(reverse=True)
    assert errs[0][0] < 5e-2, errs[:3]
    assert sum(1 for e, _ in errs if e > 5e-3) <= 3, errs[:6]


def test_bf16_compute_tracks_the_fp32_training_step(pcc):
    """BASELINE config 5's precision: bf16 operands on the wide convolutions (forward and backward-data), fp32
    accumulation, weight gradients, entropy models and losses.  Same noise, same data: the loss terms stay within
    2 % and the gradient keeps its direction."""
    from pcc_amd import autograd as ag, entropy as pe
    from pcc_amd.loss import OURS_LOSS, Loss
    model, pts, qc, qf, lam = _setup(pcc)
    model.train()
    coords = torch.from_numpy(qc).to(DEV)
    inp = pcc.SparseTensor(coordinates=coords, features=torch.from_numpy(pts[:, 3:]).to(DEV), device=DEV)
    Q = pcc.SparseTensor(torch.from_numpy(qf).to(DEV), coordinate_map=inp.map)
    Lam = pcc.SparseTensor(torch.from_numpy(lam).to(DEV), coordinate_map=inp.map)
    res = {}
    for mode in (False, True):
        ag.set_bf16(mode)
        pe.NOISE_SOURCE = coord_noise
        try:
            model.zero_grad(set_to_none=True)
            total, parts = Loss(OURS_LOSS)(inp, model(inp, Q, Lam))
            total.backward()
        finally:
            pe.NOISE_SOURCE = None
            ag.set_bf16(False)
        grads = torch.cat([p.grad.reshape(-1) for n, p in model.named_parameters() if p.grad is not None and not n.endswith("quantiles")])
        res[mode] = ({k: float(v.detach()) for k, v in parts.items()}, grads.clone())
    for key in res[False][0]:
        assert res[True][0][key] == pytest.approx(res[False][0][key], rel=2e-2), key
    g32, g16 = res[False][1], res[True][1]
    cos = float(torch.dot(g32, g16) / (g32.norm() * g16.norm()))
    assert cos > 0.99, cos
    assert float((g32 - g16).norm() / g32.norm()) < 0.15


def test_overfitting_one_frame_learns_it(pcc):
    """the system test of the training path: 120 Adam steps on one 64^3 frame (train.py:194-206), then the
    trained weights go through update() / compress / decompress — the decoded geometry must have become (nearly)
    the source's and the colours recognisable (untrained: D1 25.7 dB, Y 11.2 dB)"""
    from pcc_amd import synthetic as syn
    from pcc_amd.loss import OURS_LOSS, Loss
    from pcc_amd.metrics import PointCloudMetric
    import random
    torch.manual_seed(0)
    random.seed(0)
    model = syn.make_model(seed=0, device=DEV)
    pts = syn.sphere_shell(grid=64, radius=27.0, half_width=0.6)
    qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)
    x = torch.from_numpy(pts).to(DEV)
    coords = torch.from_numpy(qc).to(DEV)
    lam = np.stack([np.full(len(pts), 400.0), np.full(len(pts), 3000.0)], axis=1).astype(np.float32)
    inp = pcc.SparseTensor(coordinates=coords, features=x[:, 3:].contiguous(), device=DEV)
    Q = pcc.SparseTensor(torch.from_numpy(qf).to(DEV), coordinate_map=inp.map)
    Lam = pcc.SparseTensor(torch.from_numpy(lam).to(DEV), coordinate_map=inp.map)
    params = [p for n, p in model.named_parameters() if not n.endswith(".quantiles")]
    opt = torch.optim.Adam(params, lr=2e-4)
    loss_fn = Loss(OURS_LOSS)
    model.train()
    first = last = None
    for step in range(120):
        opt.zero_grad(set_to_none=True)
        total, _ = loss_fn(inp, model(inp, Q, Lam))
        total.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        first = float(total.detach()) if first is None else first
        last = float(total.detach())
    assert last < 0.05 * first, (first, last)
    model.eval()
    model.update()
    Qc = pcc.SparseTensor(coordinates=coords, features=torch.from_numpy(qf).to(DEV), device=DEV)
    strings, shape, k, c8 = model.compress(x, Qc)
    rec = model.decompress(coordinates=c8, strings=strings, shape=shape, k=k)
    res, _ = PointCloudMetric(x, rec, resolution=63).compute_pointcloud_metrics(drop_duplicates=True)
    assert res["sym_psnr_mse"] > 55.0 and res["sym_y_psnr"] > 22.0, (res["sym_psnr_mse"], res["sym_y_psnr"])
    # the same TRAINED weights through the CPU oracle: learnt occupancy logits are far from ties and the
    # reconstruction is a real cloud — the regime the seeded-weight parity tests do not reach
    from oracle.codec import Codec, count_bits
    from oracle.metrics import pc_metrics
    oracle = Codec({n: t.detach().cpu() for n, t in model.state_dict().items()})
    oracle.update()
    o_strings, o_shape, o_k, o_c8 = oracle.compress(pts, qc, qf)
    assert shape == o_shape and k == o_k
    assert set(map(tuple, c8.cpu().numpy().tolist())) == set(map(tuple, o_c8.tolist()))
    o_rec = oracle.decompress(o_c8, o_strings, o_shape, o_k)
    bits, o_bits = count_bits(strings), count_bits(o_strings)
    assert abs(bits - o_bits) <= 2e-3 * o_bits + 64, (bits, o_bits)
    got, want = pc_metrics(pts, rec.cpu().numpy(), 63), pc_metrics(pts, o_rec, 63)
    from _parity import assert_psnr_parity, voxel_flips
    flips = voxel_flips(rec.cpu().numpy(), o_rec)
    assert flips == 0, flips                                        # learnt geometry: no top-k near-ties to flip
    assert_psnr_parity(got, want, flips, pts.shape[0], "trained")   # 1e-3 dB, BASELINE's bound
