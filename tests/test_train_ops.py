"""Training path, operator level: gradients of the HIP convolution against
torch autograd through the CPU oracle's convolution (the reference differentiates through
MinkowskiEngine, train.py:194-206)."""
import numpy as np
import pytest
import torch

from oracle import coords as oc
from oracle import nn as on

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def shell(grid=36, radius=13.0, thick=0.9):
    g = np.stack(np.meshgrid(*[np.arange(grid)] * 3, indexing="ij"), -1).reshape(-1, 3)
    keep = np.abs(np.linalg.norm(g - (grid - 1) / 2, axis=1) - radius) < thick
    return np.concatenate([np.zeros((int(keep.sum()), 1), np.int32), g[keep].astype(np.int32)], axis=1)


def close(a, b, tol=2e-4):
    scale = float(b.abs().max()) + 1e-12
    return float((a - b).abs().max()) <= tol * scale


SHAPES = [(128, 128, 3), (64, 64, 3), (64, 128, 3), (128, 256, 3), (192, 64, 3), (4, 64, 3), (2, 2, 3), (2, 128, 3), (16, 16, 1),
          (128, 128, 1), (64, 3, 3), (128, 2, 3), (32, 3, 3)]


@pytest.mark.parametrize("cin,cout,ksize", SHAPES)
def test_conv_gradients_match_autograd(pcc, cin, cout, ksize):
    torch.manual_seed(cin * 7 + cout)
    c = shell()
    n = c.shape[0]
    layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=ksize, stride=1, bias=True, dimension=3).to(DEV)
    with torch.no_grad():
        layer.kernel.normal_(0, 1.0 / np.sqrt(cin * 10))
        layer.bias.normal_(0, 0.1)
    F = torch.randn(n, cin)
    G = torch.randn(n, cout)
    x = F.to(DEV).requires_grad_(True)
    out = layer(pcc.SparseTensor(x, coordinate_map=pcc.CoordMap(torch.from_numpy(c).to(DEV), 1))).F
    (out * G.to(DEV)).sum().backward()
    Fo = F.clone().requires_grad_(True)
    Wo = layer.kernel.detach().cpu().clone().requires_grad_(True)
    bo = layer.bias.detach().cpu().clone().requires_grad_(True)
    if ksize == 1:
        ref = Fo @ Wo + bo
    else:
        ref = on._apply_conv(Fo, Wo, bo, oc.kernel_map(c, c, ksize, 1), n)
    (ref * G).sum().backward()
    assert close(out.detach().cpu(), ref.detach())
    assert close(x.grad.cpu(), Fo.grad), "dX"
    assert close(layer.kernel.grad.cpu(), Wo.grad), "dW"
    assert close(layer.bias.grad.cpu(), bo.grad), "db"


@pytest.mark.parametrize("kind", ["down", "up3", "up2"])
def test_strided_and_transposed_gradients(pcc, kind):
    torch.manual_seed(3)
    c = shell() * np.array([1, 2, 2, 2], dtype=np.int32)
    n = c.shape[0]
    cin, cout = 64, 128
    if kind == "down":
        layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=3, stride=2, bias=True, dimension=3).to(DEV)
    else:
        layer = pcc.MinkowskiGenerativeConvolutionTranspose(cin, cout, kernel_size=int(kind[-1]), stride=2, bias=True, dimension=3).to(DEV)
    F = torch.randn(n, cin)
    x = F.to(DEV).requires_grad_(True)
    got = layer(pcc.SparseTensor(x, coordinate_map=pcc.CoordMap(torch.from_numpy(c).to(DEV), 2)))
    Fo = F.clone().requires_grad_(True)
    Wo = layer.kernel.detach().cpu().clone().requires_grad_(True)
    bo = layer.bias.detach().cpu().clone().requires_grad_(True)
    xo = on.SparseTensor(c, Fo, 2)
    want = on.conv(xo, Wo, bo, 3, 2) if kind == "down" else on.conv_transpose_generative(xo, Wo, bo, int(kind[-1]))
    idx = torch.from_numpy(oc.lookup(want.C, got.C.cpu().numpy())).long()          # oracle row of every product row
    G = torch.randn(want.F.shape)
    (got.F * G[idx].to(DEV)).sum().backward()
    (want.F * G).sum().backward()
    assert close(got.F.detach().cpu(), want.F.detach()[idx])
    assert close(x.grad.cpu(), Fo.grad), "dX"
    assert close(layer.kernel.grad.cpu(), Wo.grad), "dW"
    assert close(layer.bias.grad.cpu(), bo.grad), "db"


def test_fused_epilogue_terms_and_channel_slice_are_differentiated(pcc):
    """FiLM, activation and residual become torch ops on the training path; the occupancy head's
    channel-0-only evaluation (blocks.py:142) must leave zero gradient in the unused channels"""
    from pcc_amd import sparse as sp
    torch.manual_seed(5)
    c = shell()
    n = c.shape[0]
    layer = pcc.MinkowskiConvolution(64, 64, kernel_size=3, stride=1, bias=True, dimension=3).to(DEV)
    F, film, res = torch.randn(n, 64), torch.randn(n, 128), torch.randn(n, 64)
    x = F.to(DEV).requires_grad_(True)
    fm = film.to(DEV).requires_grad_(True)
    m = pcc.CoordMap(torch.from_numpy(c).to(DEV), 1)
    out = layer(pcc.SparseTensor(x, coordinate_map=m), act=sp.ACT_LRELU, film=fm, residual=res.to(DEV)).F
    out.square().sum().backward()
    Fo, fo = F.clone().requires_grad_(True), film.clone().requires_grad_(True)
    Wo = layer.kernel.detach().cpu().clone().requires_grad_(True)
    bo = layer.bias.detach().cpu().clone().requires_grad_(True)
    base = on._apply_conv(Fo, Wo, bo, oc.kernel_map(c, c, 3, 1), n)
    ref = torch.nn.functional.leaky_relu(base * fo[:, :64] + fo[:, 64:], 0.01) + res
    ref.square().sum().backward()
    assert close(x.grad.cpu(), Fo.grad) and close(fm.grad.cpu(), fo.grad) and close(layer.kernel.grad.cpu(), Wo.grad)
    layer.zero_grad()
    out1 = layer(pcc.SparseTensor(F.to(DEV), coordinate_map=m), out_channels=1).F
    assert out1.shape == (n, 1)
    out1.sum().backward()
    g = layer.kernel.grad
    assert float(g[:, :, 1:].abs().max()) == 0.0 and float(g[:, :, 0].abs().max()) > 0
    assert float(layer.bias.grad[0, 1:].abs().max()) == 0.0


def test_conv_gradients_with_two_batch_items(pcc):
    """two items that overlap in (x, y, z): neighbours never cross items, forward or backward"""
    torch.manual_seed(9)
    a = shell()
    b = shell(radius=10.0)
    b[:, 0] = 1
    c = np.concatenate([a, b], axis=0)
    n = c.shape[0]
    for cin, cout in [(64, 64), (128, 128), (4, 64)]:
        layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, bias=True, dimension=3).to(DEV)
        F, G = torch.randn(n, cin), torch.randn(n, cout)
        x = F.to(DEV).requires_grad_(True)
        out = layer(pcc.SparseTensor(x, coordinate_map=pcc.CoordMap(torch.from_numpy(c).to(DEV), 1))).F
        (out * G.to(DEV)).sum().backward()
        Fo = F.clone().requires_grad_(True)
        Wo = layer.kernel.detach().cpu().clone().requires_grad_(True)
        bo = layer.bias.detach().cpu().clone().requires_grad_(True)
        ref = on._apply_conv(Fo, Wo, bo, oc.kernel_map(c, c, 3, 1), n)
        (ref * G).sum().backward()
        assert close(out.detach().cpu(), ref.detach())
        assert close(x.grad.cpu(), Fo.grad), ("dX", cin, cout)
        assert close(layer.kernel.grad.cpu(), Wo.grad), ("dW", cin, cout)


@pytest.mark.parametrize("cin,cout", [(4, 64), (64, 1), (2, 2), (16, 16), (32, 3)])
def test_scalar_weight_gradient_entry_point(pcc, cin, cout):
    """pcc_conv_wgrad on shapes that are not multiples of 32 (the autograd function pads such shapes onto the
    MFMA kernel; the C-ABI keeps a scalar kernel for direct callers)"""
    from pcc_amd import _lib
    from pcc_amd._lib import check, ptr
    L = pcc.lib()
    torch.manual_seed(cin + 31 * cout)
    c = shell()
    n = c.shape[0]
    X, G = torch.randn(n, cin), torch.randn(n, cout)
    m = pcc.CoordMap(torch.from_numpy(c).to(DEV), 1)
    nbr, _, _ = m.kernel_map(m, 3)
    dw = torch.empty((27, cin, cout), dtype=torch.float32, device=DEV)
    ne = L.pcc_conv_wgrad_scratch_elems(27, cin, cout)
    scratch = torch.empty(ne, dtype=torch.float32, device=DEV)
    xd, gd = X.to(DEV), G.to(DEV)
    check(L.pcc_conv_wgrad(ptr(xd), n, cin, ptr(gd), n, cout, ptr(nbr), None, None, 27, ptr(dw), ptr(scratch), ne, _lib.stream()))
    nb = torch.from_numpy(oc.kernel_map(c, c, 3, 1)).long()
    want = torch.zeros(27, cin, cout)
    for k in range(27):
        ok = nb[:, k] >= 0
        want[k] = X[nb[ok, k]].t() @ G[ok]
    assert close(dw.cpu(), want)


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("with_film,with_res", [(True, True), (True, False), (False, True), (False, False)])
def test_fused_epilogue_equals_the_torch_chain_bitwise(pcc, act, with_film, with_res):
    """csrc/epilogue.hip: out = act(c * beta + gamma) + residual as one kernel forward and one backward — the same
    operation order as the torch ops it replaces on the training path, so values AND gradients are bit-identical"""
    from pcc_amd.autograd import EpilogueFn
    if act == 0 and not with_film and not with_res:
        pytest.skip("identity")
    torch.manual_seed(act * 4 + with_film * 2 + with_res)
    n, ch = 1000, 64
    c = torch.randn(n, ch, device=DEV, requires_grad=True)
    film = torch.randn(n, 2 * ch, device=DEV, requires_grad=True) if with_film else None
    res = torch.randn(n, ch, device=DEV, requires_grad=True) if with_res else None
    g = torch.randn(n, ch, device=DEV)
    u = c * film[:, :ch] + film[:, ch:] if with_film else c
    v = torch.relu(u) if act == 1 else torch.nn.functional.leaky_relu(u, 0.01) if act == 2 else u
    want = v + res if with_res else v
    leaves = [t for t in (c, film, res) if t is not None]
    want_g = torch.autograd.grad(want, leaves, g)
    got = EpilogueFn.apply(c, film, res, act)
    got_g = torch.autograd.grad(got, leaves, g)
    assert torch.equal(got, want)
    for a, b in zip(got_g, want_g):
        assert torch.equal(a, b)


def test_pruning_and_row_scatter_are_differentiated_in_hip(pcc):
    """Training-path row movers (ME.MinkowskiPruning, the q-map scatter of model/transforms.py): forward values and
    gradients of the HIP autograd functions equal torch's boolean indexing / index_add bit for bit"""
    from pcc_amd import sparse as sp
    torch.manual_seed(11)
    n, ch = 5000, 24
    feats = torch.randn(n, ch, device=DEV, requires_grad=True)
    mask = (torch.rand(n, device=DEV) < 0.4)
    coords = torch.randint(0, 100, (n, 4), dtype=torch.int32, device=DEV)
    want = feats[mask]
    g = torch.randn_like(want)
    (want_g,) = torch.autograd.grad(want, feats, g)
    coords_k, got, new_index, m = sp.compact_rows(mask.to(torch.uint8), coords, feats, want_index=True)
    assert got.grad_fn is not None and m == int(mask.sum()) and torch.equal(coords_k, coords[mask])
    assert torch.equal(new_index[mask].long(), torch.arange(m, device=DEV)) and bool((new_index[~mask] == -1).all())
    (got_g,) = torch.autograd.grad(got, feats, g)
    assert torch.equal(got, want) and torch.equal(got_g, want_g)
    # nothing kept / everything kept
    for mk in (torch.zeros(n, dtype=torch.uint8, device=DEV), torch.ones(n, dtype=torch.uint8, device=DEV)):
        _, f2, _, m2 = sp.compact_rows(mk, coords, feats)
        assert m2 == int(mk.sum()) and torch.equal(f2, feats[mk.bool()])
        (g2,) = torch.autograd.grad(f2.sum(), feats)
        assert torch.equal(g2, mk.float().reshape(-1, 1).expand(n, ch))
    # scatter: unique targets, some rows dropped
    n_out = 7000
    idx = torch.randperm(n_out, device=DEV)[:n].to(torch.int32)
    idx[::7] = -1
    ok = idx >= 0
    want = torch.zeros((n_out, ch), device=DEV).index_add(0, idx[ok].long(), feats[ok])
    g = torch.randn_like(want)
    (want_g,) = torch.autograd.grad(want, feats, g)
    got = sp.scatter_rows(feats, idx, n_out)
    (got_g,) = torch.autograd.grad(got, feats, g)
    assert torch.equal(got, want) and torch.equal(got_g, want_g)


@pytest.mark.parametrize("n,ch", [(1, 4), (777, 64), (5000, 100), (100003, 128), (3000, 256), (300, 1024), (0, 64)])
def test_cast_and_column_sums_in_one_pass(pcc, n, ch):
    """pcc_cast_colsum (the convolution backward's single read of dY): the bf16 copy equals torch's conversion bit for bit
    (round to nearest even, infinities, NaN), the column sums equal a float64 sum to fp32 rounding and are deterministic"""
    from pcc_amd import _lib
    from pcc_amd._lib import check, ptr
    L = pcc.lib()
    torch.manual_seed(n + ch)
    x = (torch.randn(n, ch) * torch.exp(torch.randn(n, ch) * 4)).to(DEV)
    if n > 2:
        x[0, 0], x[1, 1], x[2, 2] = float("inf"), float("-inf"), 0.0
        x[2, 3] = torch.tensor(0x3F808000, dtype=torch.int32).view(torch.float32)      # exactly halfway: ties to even
    ne = L.pcc_cast_colsum_scratch_elems(ch)
    scratch = torch.empty(ne, dtype=torch.float32, device=DEV)
    out = torch.empty((n, ch), dtype=torch.bfloat16, device=DEV)
    cs = torch.full((ch,), 7.0, device=DEV)
    check(L.pcc_cast_colsum(ptr(x), n, ch, ptr(out), ptr(cs), ptr(scratch), ne, _lib.stream()))
    assert torch.equal(out.view(torch.int16), x.to(torch.bfloat16).view(torch.int16))
    finite = torch.nan_to_num(x, posinf=0.0, neginf=0.0) if n > 2 else x
    if n > 2:                                                   # the infinities make their columns inf / -inf
        assert cs[0] == float("inf") and cs[1] == -float("inf")
    want = finite.double().sum(0)
    scale = finite.double().abs().sum(0) + 1e-30
    ok = torch.ones(ch, dtype=torch.bool, device=DEV)
    if n > 2:
        ok[:2] = False
    assert bool((((cs.double() - want).abs() / scale)[ok] < 1e-6).all())
    cs2 = torch.empty_like(cs)
    check(L.pcc_cast_colsum(ptr(x), n, ch, None, ptr(cs2), ptr(scratch), ne, _lib.stream()))
    assert torch.equal(cs2[ok], cs[ok])
    if n > 0:
        xn = x.clone()
        xn[0, 0] = float("nan")
        check(L.pcc_cast_colsum(ptr(xn), n, ch, ptr(out), None, None, 0, _lib.stream()))
        assert torch.equal(out.view(torch.int16), xn.to(torch.bfloat16).view(torch.int16))
