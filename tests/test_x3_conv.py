"""Split-bf16 arithmetic on fp32 data (pcc_conv_fwd_x3, opt-in PCC_INFER_X3): every fp32 operand is split exactly into
three bf16 numbers and the six significant products run on the bf16 MFMA with fp32 accumulation.  Checked here:
the split is exact, the convolution agrees with a float64 evaluation as well as the fp32 MFMA kernel does, the fused
epilogue is unchanged, results are deterministic and row-order invariant, and the whole codec in this mode meets the
same oracle parity bounds as the fp32 codec (bpp 2e-3, D1 / Y-PSNR 1e-3 dB) with encoder and decoder agreeing bit for
bit."""
import numpy as np
import pytest
import torch

from oracle import coords as oc
from oracle import nn as on
from _parity import compare_codec

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def shell(grid=36, radius=13.0, thick=0.9):
    g = np.stack(np.meshgrid(*[np.arange(grid)] * 3, indexing="ij"), -1).reshape(-1, 3)
    keep = np.abs(np.linalg.norm(g - (grid - 1) / 2, axis=1) - radius) < thick
    return np.concatenate([np.zeros((int(keep.sum()), 1), np.int32), g[keep].astype(np.int32)], axis=1)


def conv_f64(F, W, b, nbr, n_out):
    """float64 evaluation of the sparse convolution (reference for the error bounds below)"""
    out = torch.zeros((n_out, W.shape[2]), dtype=torch.float64)
    for k in range(nbr.shape[1]):
        rows = np.nonzero(nbr[:, k] >= 0)[0]
        if rows.size:
            out.index_add_(0, torch.from_numpy(rows), F.double()[torch.from_numpy(nbr[rows, k]).long()] @ W[k].double())
    return out + b.double().reshape(1, -1)


def test_weight_planes_sum_to_the_weights_exactly(pcc):
    from pcc_amd import _lib
    from pcc_amd._lib import check, ptr
    L = pcc.lib()
    torch.manual_seed(0)
    K, cin, cout = 27, 64, 128
    W = (torch.randn(K, cin, cout) * torch.exp(torch.randn(K, cin, cout) * 3)).to(DEV).contiguous()      # wide dynamic range
    W[0, 0, :4] = torch.tensor([0.0, -0.0, 1.0, -3.0e-30], device=DEV)
    wp = torch.empty(L.pcc_conv_packed_elems_x3(K, cin, cout), dtype=torch.bfloat16, device=DEV)
    check(L.pcc_conv_pack_weights_x3(ptr(W), K, cin, cout, ptr(wp), _lib.stream()))
    planes = wp.reshape(K, cin // 32, 3, 4, cout, 8).float()                      # [k][chunk][plane][group][col][j]
    total = planes[:, :, 0] + planes[:, :, 1] + planes[:, :, 2]                    # fp32 adds of exact pieces, big to small
    back = total.permute(0, 1, 2, 4, 3).reshape(K, cin, cout)                      # ci = 32 chunk + 8 group + j
    assert torch.equal(back, W)


@pytest.mark.parametrize("cin,cout,ksize", [(64, 64, 3), (128, 128, 3), (128, 256, 3), (64, 128, 3), (128, 64, 3), (192, 256, 3),
                                             (96, 64, 3), (32, 64, 3), (128, 128, 1)])
def test_x3_conv_is_fp32_accurate(pcc, cin, cout, ksize):
    from pcc_amd import _lib
    from pcc_amd._lib import check, ptr
    L = pcc.lib()
    torch.manual_seed(cin * 7 + cout)
    c = shell()
    n = c.shape[0]
    K = ksize ** 3
    F = torch.randn(n, cin) * torch.exp(torch.randn(n, 1))
    W = torch.randn(K, cin, cout) / np.sqrt(cin * 10)
    b = torch.randn(cout) * 0.1
    film = torch.cat([1 + 0.1 * torch.randn(n, cout), 0.1 * torch.randn(n, cout)], dim=1)
    res = torch.randn(n, cout)
    m = pcc.CoordMap(torch.from_numpy(c).to(DEV), 1)
    if ksize == 1:
        nbr = order = gmask = None
        ref64 = F.double() @ W.double()[0] + b.double()
    else:
        nbr, order, gmask, _ = m.ordered_kernel_map(m, ksize)
        ref64 = conv_f64(F, W, b, oc.kernel_map(c, c, ksize, 1), n)
    ref64 = torch.relu(ref64 * film[:, :cout].double() + film[:, cout:].double()) + res.double()
    Wd = W.to(DEV).contiguous()
    wp3 = torch.empty(L.pcc_conv_packed_elems_x3(K, cin, cout), dtype=torch.bfloat16, device=DEV)
    check(L.pcc_conv_pack_weights_x3(ptr(Wd), K, cin, cout, ptr(wp3), _lib.stream()))
    wp = torch.empty(L.pcc_conv_packed_elems(K, cin, cout), dtype=torch.float32, device=DEV)
    check(L.pcc_conv_pack_weights(ptr(Wd), K, cin, cout, ptr(wp), _lib.stream()))
    x, bd, fd, rd = F.to(DEV).contiguous(), b.to(DEV), film.to(DEV).contiguous(), res.to(DEV).contiguous()
    out3 = torch.empty((n, cout), dtype=torch.float32, device=DEV)
    out32 = torch.empty_like(out3)
    check(L.pcc_conv_fwd_x3(ptr(x), n, cin, ptr(wp3), ptr(bd), ptr(nbr), ptr(order), ptr(gmask), K, ptr(out3), n, cout, 1, ptr(fd),
                            ptr(rd), _lib.stream()))
    check(L.pcc_conv_fwd(ptr(x), n, cin, ptr(Wd), ptr(wp), ptr(bd), ptr(nbr), ptr(order), ptr(gmask), K, ptr(out32), n, cout, 1,
                         ptr(fd), ptr(rd), _lib.stream()))
    again = torch.empty_like(out3)
    check(L.pcc_conv_fwd_x3(ptr(x), n, cin, ptr(wp3), ptr(bd), ptr(nbr), ptr(order), ptr(gmask), K, ptr(again), n, cout, 1, ptr(fd),
                            ptr(rd), _lib.stream()))
    assert torch.equal(out3, again)                                             # deterministic
    scale = float(ref64.abs().max())
    e3 = float((out3.cpu().double() - ref64).abs().max()) / scale
    e32 = float((out32.cpu().double() - ref64).abs().max()) / scale
    # the fp32 MFMA kernel itself is ~1e-7 .. 1e-6 from float64 here; the split arithmetic must be in the same class
    assert e3 < 4e-6 and e3 < 4 * e32 + 1e-6, (e3, e32)
    # natural row order (no execution order, no tile skipping): same bits
    if ksize == 3:
        nbr_nat, _, _ = m.kernel_map(m, ksize)
        nat = torch.empty_like(out3)
        check(L.pcc_conv_fwd_x3(ptr(x), n, cin, ptr(wp3), ptr(bd), ptr(nbr_nat), None, None, K, ptr(nat), n, cout, 1, ptr(fd), ptr(rd),
                                _lib.stream()))
        assert torch.equal(nat, out3)


def test_codec_in_x3_mode_meets_the_oracle_parity_bounds(pcc, oracle_codec):
    from pcc_amd import sparse as sp
    model = pcc.synthetic.make_model(0, DEV)
    model.update()
    assert not sp.INFER_X3
    sp.set_infer_x3(True)
    try:
        for cfg in (dict(grid=32, radius=15.0, half_width=0.875), dict(grid=96, radius=40.0, half_width=0.5)):
            pts = pcc.synthetic.sphere_shell(**cfg)
            qc, qf = pcc.synthetic.uniform_qmap(pts[:, :3], 0.5, 0.5)
            N = pts.shape[0]
            x = torch.from_numpy(pts).to(DEV)

            def code():
                Q = pcc.SparseTensor(coordinates=torch.from_numpy(qc).to(DEV), features=torch.from_numpy(qf).to(DEV), device=DEV)
                strings, shape, k, coords = model.compress(x, Q)
                rec = model.decompress(coordinates=coords, strings=strings, shape=shape, k=k)
                return strings, rec
            s1, rec = code()
            s2, rec2 = code()
            assert s1 == s2 and torch.equal(rec, rec2)                           # deterministic
            compare_codec(pcc, model, oracle_codec, pts, qc, qf, ("x3", cfg), DEV, exact=False)   # the fp32 codec's stage-by-stage bounds
        # encoder-side and decoder-side latents agree bit for bit in this mode too
        coords4 = torch.cat([torch.zeros((N, 1), device=DEV, dtype=torch.int32), x[:, :3].to(torch.int32)], dim=1)
        feats = torch.cat([torch.ones((N, 1), device=DEV), x[:, 3:6]], dim=1)
        inp = pcc.SparseTensor(feats, coordinate_map=pcc.CoordMap(coords4, 1, nbatch=1))
        Q = pcc.SparseTensor(coordinates=torch.from_numpy(qc).to(DEV), features=torch.from_numpy(qf).to(DEV), device=DEV)
        em = model.entropy_model
        y, _, _ = model.g_a(inp, Q)
        _, strings, shape = em.compress(y)
        y_hat_enc, _, _ = em(y)
        c8 = pcc.CoordMap(y.C, 8, nbatch=1)
        y_hat_dec, _ = em.decompress([c8, c8.down().down()], strings, shape)
        idx = y_hat_dec.map.lookup(y.C).long()
        assert torch.equal(y_hat_dec.F[idx], y_hat_enc.F)
    finally:
        sp.set_infer_x3(False)


def test_training_forward_and_backward_data_in_x3_mode_track_fp32(pcc):
    """PCC_TRAIN_X3 / autograd.set_x3: the differentiable convolution's forward and input gradient through the split
    arithmetic agree with the fp32 kernels to fp32 rounding; the weight gradient (fp32 kernel in both modes) is identical"""
    from pcc_amd import autograd as ag
    torch.manual_seed(5)
    c = shell()
    n = c.shape[0]
    m = pcc.CoordMap(torch.from_numpy(c).to(DEV), 1)
    layer = pcc.MinkowskiConvolution(128, 128, kernel_size=3, stride=1, bias=True, dimension=3).to(DEV)
    x = torch.randn(n, 128, device=DEV, requires_grad=True)
    g = torch.randn(n, 128, device=DEV)
    res = {}
    for mode in (False, True):
        ag.set_x3(mode)
        try:
            out = layer(pcc.SparseTensor(x, coordinate_map=m)).F
            dx, dw = torch.autograd.grad(out, [x, layer.kernel], g)
        finally:
            ag.set_x3(False)
        res[mode] = (out.detach(), dx, dw)
    for a, b in zip(res[False][:2], res[True][:2]):
        assert float((a - b).abs().max()) <= 4e-6 * float(a.abs().max())
        assert not torch.equal(a, b)                                       # the mode really ran
    assert torch.equal(res[False][2], res[True][2])
