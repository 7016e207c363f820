"""16-row MFMA tiles (pcc_conv_fwd16, v_mfma_f32_16x16x4_f32) against the 32-row tiles (pcc_conv_fwd,
v_mfma_f32_32x32x2_f32): BIT FOR BIT — one 16x16x4 runs the same fp32 chain as two chained 32x32x2
(tools/micro/mfma_shapes_bitwise.hip) — on every tile shape the dispatcher picks (row counts from one tile to several
waves of workgroups), dense and mask-diverse sets, strided / transposed maps and every fused epilogue; plus the group
masks per 16 positions against numpy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _inference_mode():
    """these tests are about the INFERENCE kernels: with gradients enabled a layer with trainable parameters takes the
    autograd path (conv_train), which never dispatches to the kernels under test"""
    with torch.no_grad():
        yield


def dev(a):
    return torch.as_tensor(a).to(DEV).contiguous()


def _coords(pcc, kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "shell":
        p = pcc.synthetic.sphere_shell(64, 27.0, 0.9)[:, :3]
    elif kind == "big":                                       # ~100 k rows: the 64 x 128 / 128 x 64 tile shapes
        p = pcc.synthetic.sphere_shell(224, 100.0, 0.5)[:, :3]
    elif kind == "sparse":                                    # a random 35 % of a shell: diverse masks, half-empty tiles
        p = pcc.synthetic.sphere_shell(96, 42.0, 0.9)[:, :3]
        p = p[rng.random(p.shape[0]) < 0.35]
    else:
        p = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [9, 9, 9], [1, 1, 1]], np.float32)
    c = np.concatenate([np.zeros((p.shape[0], 1)), p], axis=1).astype(np.int32)
    return c[rng.permutation(c.shape[0])]


def _both(layer, x, **kw):
    from pcc_amd import sparse as sp
    was = sp.CONV_T16
    try:
        sp.set_conv_t16(True)
        a = layer(x, **kw)
        sp.set_conv_t16(False)
        b = layer(x, **kw)
    finally:
        sp.set_conv_t16(was)
    return a, b


def test_group_masks_per_16_positions(pcc):
    c = _coords(pcc, "sparse")
    m = pcc.CoordMap(dev(c), 1)
    nbr_sorted, order, gmask, _ = m.ordered_kernel_map(m, 3)
    g16 = m.group_mask16(m, 3).cpu().numpy().view(np.uint32)
    rows = ((nbr_sorted.cpu().numpy() >= 0).astype(np.uint32) << np.arange(27, dtype=np.uint32)).sum(axis=1).astype(np.uint32)
    n = rows.shape[0]
    pad = np.zeros((-n) % 16, np.uint32)
    want16 = np.bitwise_or.reduce(np.concatenate([rows, pad]).reshape(-1, 16), axis=1)
    assert np.array_equal(g16, want16)
    pad = np.zeros((-n) % 32, np.uint32)
    want32 = np.bitwise_or.reduce(np.concatenate([rows, pad]).reshape(-1, 32), axis=1)
    assert np.array_equal(gmask.cpu().numpy().view(np.uint32), want32)


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 64), (64, 128), (128, 64), (128, 128), (128, 256), (192, 256), (64, 32), (128, 3)])
@pytest.mark.parametrize("kind", ["shell", "sparse", "five"])
def test_t16_equals_t32_bitwise(pcc, cin, cout, kind):
    from pcc_amd import sparse as sp
    torch.manual_seed(cin * 1000 + cout)
    c = _coords(pcc, kind, seed=cin + cout)
    n = c.shape[0]
    layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, bias=True, dimension=3)
    with torch.no_grad():
        layer.kernel.normal_(0, 1.0 / np.sqrt(cin * 10))
        layer.bias.normal_(0, 0.1)
    layer = layer.to(DEV)
    F, film, res = torch.randn(n, cin), torch.randn(n, 2 * cout), torch.randn(n, cout)
    x = pcc.SparseTensor(dev(F), coordinate_map=pcc.CoordMap(dev(c), 1))
    cases = [{}, dict(act=sp.ACT_RELU, residual=dev(res))]
    if cout > 4:                                             # (narrow heads take no FiLM / residual: another path)
        cases += [dict(film=dev(film)), dict(act=sp.ACT_LRELU, film=dev(film), residual=dev(res))]
    for kw in cases[:1] if cout <= 4 else cases:
        a, b = _both(layer, x, **kw)
        assert torch.equal(a.F, b.F), (kw.keys(), float((a.F - b.F).abs().max()))


@pytest.mark.parametrize("cin,cout", [(128, 128), (64, 64), (128, 256)])
def test_t16_equals_t32_bitwise_on_large_launches(pcc, cin, cout):
    """~100 k rows: the 64 x 128, 128 x 64 and 32 x 128 tile shapes of full launches"""
    torch.manual_seed(5)
    c = _coords(pcc, "big")
    n = c.shape[0]
    layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, bias=True, dimension=3).to(DEV)
    x = pcc.SparseTensor(dev(torch.randn(n, cin)), coordinate_map=pcc.CoordMap(dev(c), 1))
    a, b = _both(layer, x)
    assert torch.equal(a.F, b.F)
    keep = np.random.default_rng(1).random(n) < 0.3           # and a mask-diverse subset of it
    xs = pcc.SparseTensor(dev(torch.randn(int(keep.sum()), cin)), coordinate_map=pcc.CoordMap(dev(c[keep]), 1))
    a, b = _both(layer, xs)
    assert torch.equal(a.F, b.F)


@pytest.mark.parametrize("kind", ["down", "up3", "up2"])
def test_t16_strided_and_transposed_maps(pcc, kind):
    torch.manual_seed(3)
    c = _coords(pcc, "sparse", seed=9) * np.array([1, 2, 2, 2], dtype=np.int32)
    cin, cout = 64, 128
    if kind == "down":
        layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=3, stride=2, bias=True, dimension=3)
    else:
        layer = pcc.MinkowskiGenerativeConvolutionTranspose(cin, cout, kernel_size=int(kind[-1]), stride=2, bias=True, dimension=3)
    layer = layer.to(DEV)
    x = pcc.SparseTensor(dev(torch.randn(c.shape[0], cin)), coordinate_map=pcc.CoordMap(dev(c), 2))
    a, b = _both(layer, x)
    assert torch.equal(a.C, b.C) and torch.equal(a.F, b.F)


def test_t16_codes_a_frame_to_the_same_bytes(pcc):
    from pcc_amd import sparse as sp
    syn = pcc.synthetic
    model = syn.make_model(0, DEV)
    model.update()
    pts = syn.sphere_shell(grid=96, radius=40.0, half_width=0.5)
    qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)

    def run():
        Q = pcc.SparseTensor(coordinates=dev(qc), features=dev(qf), device=DEV)
        strings, shape, k, coords = model.compress(dev(pts), Q)
        return strings, shape, k, model.decompress(coordinates=coords, strings=strings, shape=shape, k=k)

    was = sp.CONV_T16
    try:
        sp.set_conv_t16(False)
        s0, sh0, k0, r0 = run()
        sp.set_conv_t16(True)
        s1, sh1, k1, r1 = run()
    finally:
        sp.set_conv_t16(was)
    assert s0 == s1 and sh0 == sh1 and k0 == k1 and torch.equal(r0, r1)
