"""The small-launch convolution kernel (csrc/conv.hip conv_small_kernel: one 16 x 16 MFMA block per wave, 32 x 32 output
tiles, loader waves that run the LDS-DMAs several steps ahead) against the ordinary tile kernels: BIT FOR BIT, on row counts from one row
to several thousand, every channel-chunk count the codec has, strided / transposed maps, mask-diverse sets and every fused
epilogue — and a frame coded to the same bytes with the kernel on and off."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _inference_mode():
    """the kernels under test are the INFERENCE kernels (with gradients enabled a layer takes the autograd path)"""
    with torch.no_grad():
        yield


def dev(a):
    return torch.as_tensor(a).to(DEV).contiguous()


def _coords(pcc, kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "shell":                                       # 3.6 k rows
        p = pcc.synthetic.sphere_shell(48, 17.0, 0.9)[:, :3]
    elif kind == "sparse":                                    # a random 35 % of a shell: diverse masks, half-empty tiles
        p = pcc.synthetic.sphere_shell(64, 27.0, 0.9)[:, :3]
        p = p[rng.random(p.shape[0]) < 0.35]
    elif kind == "tiny":                                      # 33 rows: one full tile and a tile with one row
        p = pcc.synthetic.sphere_shell(48, 17.0, 0.9)[:33, :3]
    else:
        p = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [9, 9, 9], [1, 1, 1]], np.float32)
    c = np.concatenate([np.zeros((p.shape[0], 1)), p], axis=1).astype(np.int32)
    return c[rng.permutation(c.shape[0])]


def _both(layer, x, **kw):
    """(small-launch kernel, ordinary kernels) on the same input"""
    from pcc_amd import sparse as sp
    was = sp.set_conv_small_max(-1)
    try:
        sp.set_conv_small_max(1 << 20)
        a = layer(x, **kw)
        sp.set_conv_small_max(0)
        b = layer(x, **kw)
    finally:
        sp.set_conv_small_max(was)
    return a, b


def test_threshold_reads_back(pcc):
    from pcc_amd import sparse as sp
    was = sp.set_conv_small_max(-1)
    assert was >= 0
    assert sp.set_conv_small_max(7) == was and sp.set_conv_small_max(was) == 7 and sp.set_conv_small_max(-1) == was


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 64), (64, 128), (96, 64), (128, 64), (128, 128), (128, 256), (192, 256), (256, 128),
                                      (64, 32), (128, 3)])
@pytest.mark.parametrize("kind", ["shell", "sparse", "tiny", "five"])
def test_small_equals_tile_kernels_bitwise(pcc, cin, cout, kind):
    from pcc_amd import sparse as sp
    torch.manual_seed(cin * 1000 + cout)
    c = _coords(pcc, kind, seed=cin + cout)
    n = c.shape[0]
    layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, bias=True, dimension=3)
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


@pytest.mark.parametrize("cin,cout,keep", [(128, 128, 0.05), (128, 128, 0.6), (64, 64, 0.6), (32, 64, 0.6), (96, 128, 0.6)])
def test_every_pipeline_shape_bitwise(pcc, cin, cout, keep):
    """the three pipeline shapes of conv_small_kernel as the dispatcher picks them (csrc/conv.hip launch_small): four chunks
    per step while the launch is at most 256 workgroups and cin % 128 == 0, two chunks per step for even chunk counts, one
    chunk per step (eight stages) for odd ones — each against the tile kernels, bit for bit"""
    from pcc_amd import sparse as sp
    torch.manual_seed(1)
    p = pcc.synthetic.sphere_shell(48, 17.0, 0.9)[:, :3]
    p = p[np.random.default_rng(2).random(p.shape[0]) < keep]
    c = torch.from_numpy(np.concatenate([np.zeros((p.shape[0], 1)), p], axis=1).astype(np.int32)).to(DEV)
    layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, bias=True, dimension=3).to(DEV)
    x = pcc.SparseTensor(torch.randn(c.shape[0], cin, device=DEV), coordinate_map=pcc.CoordMap(c, 1))
    was = sp.set_conv_small_max(-1)
    try:
        with torch.no_grad():
            sp.set_conv_small_max(1 << 20)
            a = layer(x).F.clone()
            sp.set_conv_small_max(0)
            b = layer(x).F
    finally:
        sp.set_conv_small_max(was)
    assert torch.equal(a, b), (cin, cout, float((a - b).abs().max()))


@pytest.mark.parametrize("kind", ["down", "up3", "up2"])
def test_small_strided_and_transposed_maps(pcc, kind):
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


def test_small_kernel_codes_a_frame_to_the_same_bytes(pcc):
    from pcc_amd import sparse as sp
    syn = pcc.synthetic
    model = syn.make_model(0, DEV)
    model.update()
    pts = syn.sphere_shell(**syn.CONFIG1)
    qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)

    def run():
        Q = pcc.SparseTensor(coordinates=dev(qc), features=dev(qf), device=DEV)
        strings, shape, k, coords = model.compress(dev(pts), Q)
        return strings, shape, k, model.decompress(coordinates=coords, strings=strings, shape=shape, k=k)

    was = sp.set_conv_small_max(-1)
    try:
        sp.set_conv_small_max(0)
        s0, sh0, k0, r0 = run()
        sp.set_conv_small_max(1 << 20)                        # every map convolution of the frame
        s1, sh1, k1, r1 = run()
    finally:
        sp.set_conv_small_max(was)
    assert s0 == s1 and sh0 == sh1 and k0 == k1 and torch.equal(r0, r1)
