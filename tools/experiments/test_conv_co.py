"""The compacted-offset convolution (csrc/conv_co.hip, pcc_compact_map / pcc_conv_fwd_co) against (i) the mask-ordered
kernel of csrc/conv.hip — BIT FOR BIT, the property that lets the two be swapped under a decoder — and (ii) the CPU oracle
(oracle/nn.py: per-offset gather -> sgemm -> index_add_) within the fp32 summation-order tolerance of the other convolution
tests.  Shapes: every (cin, cout) class of configs/Ours.yaml that takes the path, same-map / strided / transposed (K = 27
and K = 8) maps, dense shells and sparse random subsets (the sets an untrained decoder keeps), row counts that are not a
multiple of the 256-row group, and one- and five-row sets."""
import numpy as np
import pytest
import torch

from oracle import coords as oc
from oracle import nn as on

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _inference_mode():
    """these tests are about the INFERENCE kernels: with gradients enabled a layer with trainable parameters takes the
    autograd path (conv_train), which never dispatches to the kernels under test"""
    with torch.no_grad():
        yield


def dev(a, dtype=None):
    t = torch.as_tensor(a)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def _coords(pcc, kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "shell":
        p = pcc.synthetic.sphere_shell(48, 20.0, 0.9)[:, :3]
    elif kind == "sparse":                                   # a random 35 % of a shell: ~4 neighbours per row, diverse masks
        p = pcc.synthetic.sphere_shell(64, 27.0, 0.9)[:, :3]
        p = p[rng.random(p.shape[0]) < 0.35]
    elif kind == "block":                                    # a filled block: all 27 offsets almost everywhere
        g = np.arange(14)
        p = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float32) + 3
    elif kind == "five":
        p = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [9, 9, 9], [1, 1, 1]], np.float32)
    else:
        p = np.array([[4, 5, 6]], np.float32)
    c = np.concatenate([np.zeros((p.shape[0], 1)), p], axis=1).astype(np.int32)
    return c[rng.permutation(c.shape[0])]


def _compact_reference(nbr, R=256):
    """numpy restatement of pcc_compact_map: lists per group of R rows and offset; padding entries gather nothing (-1) and
    name the group's first row that is NOT in the list"""
    n, K = nbr.shape
    G = (n + R - 1) // R
    ent_in = np.full((G, K, R), -1, np.int32)
    rows = np.zeros((G, K, R), np.int64)
    cnt = np.zeros((G, 32), np.int16)
    for g in range(G):
        blk = np.full((R, K), -1, np.int64)
        blk[:min(R, n - g * R)] = nbr[g * R:(g + 1) * R]
        for k in range(K):
            r = np.nonzero(blk[:, k] >= 0)[0]
            ent_in[g, k, :r.size] = blk[r, k]
            rows[g, k, :r.size] = r
            if r.size < R:
                rows[g, k, r.size:] = np.nonzero(blk[:, k] < 0)[0][0]
            cnt[g, k] = r.size
    row4 = np.zeros((G, K, 2, 32), np.int64)
    for sp in range(2):
        for i in range(R // 64):
            s0 = 32 * (sp + 2 * i)
            row4[:, :, sp, :] |= rows[:, :, s0:s0 + 32] << (8 * i)
    return ent_in, row4.astype(np.uint32).view(np.int32), cnt


@pytest.mark.parametrize("kind", ["shell", "sparse", "five"])
def test_compact_map_lists(pcc, kind):
    c = _coords(pcc, kind)
    m = pcc.CoordMap(dev(c), 1)
    nbr, _, _ = m.kernel_map(m, 3)
    ent_in, ent_row4, cnt, pairs = m.compact_kernel_map(m, 3)
    w_in, w_row4, w_cnt = _compact_reference(nbr.cpu().numpy())
    assert np.array_equal(cnt.cpu().numpy(), w_cnt)
    assert np.array_equal(ent_in.cpu().numpy(), w_in)
    assert np.array_equal(ent_row4.cpu().numpy(), w_row4)
    assert int(pairs) == int((nbr >= 0).sum())


def _both(pcc, layer, x, **kw):
    """(compacted-offset kernel, mask-ordered kernel) on the same input"""
    from pcc_amd import sparse as sp
    was = sp.CONV_CO
    try:
        sp.set_conv_co(True)
        a = layer(x, **kw)
        sp.set_conv_co(False)
        b = layer(x, **kw)
    finally:
        sp.set_conv_co(was)
    return a, b


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 64), (64, 128), (128, 64), (128, 128), (128, 256), (192, 256), (96, 192)])
@pytest.mark.parametrize("kind", ["shell", "sparse", "block", "five", "one"])
def test_co_equals_mask_ordered_kernel_bitwise_and_oracle(pcc, cin, cout, kind):
    from pcc_amd import sparse as sp
    torch.manual_seed(cin * 1000 + cout)
    c = _coords(pcc, kind, seed=cin + cout)
    n = c.shape[0]
    m = pcc.CoordMap(dev(c), 1)
    layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, bias=True, dimension=3)
    with torch.no_grad():
        layer.kernel.normal_(0, 1.0 / np.sqrt(cin * 10))
        layer.bias.normal_(0, 0.1)
    layer = layer.to(DEV)
    F, film, res = torch.randn(n, cin), torch.randn(n, 2 * cout), torch.randn(n, cout)
    x = pcc.SparseTensor(dev(F), coordinate_map=m)
    base = on._apply_conv(F, layer.kernel.detach().cpu(), layer.bias.detach().cpu(), oc.kernel_map(c, c, 3, 1), n)
    scale = float(base.abs().max())
    for kw, want in (({}, base),
                     (dict(film=dev(film)), base * film[:, :cout] + film[:, cout:]),
                     (dict(act=sp.ACT_RELU, residual=dev(res)), torch.relu(base) + res),
                     (dict(act=sp.ACT_LRELU, film=dev(film), residual=dev(res)),
                      torch.nn.functional.leaky_relu(base * film[:, :cout] + film[:, cout:], 0.01) + res)):
        a, b = _both(pcc, layer, x, **kw)
        assert torch.equal(a.F, b.F), (kw.keys(), float((a.F - b.F).abs().max()))
        assert torch.allclose(a.F.cpu(), want, rtol=1e-4, atol=1e-4 * max(scale, 1.0)), float((a.F.cpu() - want).abs().max())


@pytest.mark.parametrize("kind", ["down", "up3", "up2"])
@pytest.mark.parametrize("geometry", ["shell", "sparse"])
def test_co_strided_and_transposed_maps(pcc, kind, geometry):
    torch.manual_seed(3)
    c = _coords(pcc, geometry, seed=9) * np.array([1, 2, 2, 2], dtype=np.int32)
    n = c.shape[0]
    cin, cout = 64, 128
    F = torch.randn(n, cin)
    if kind == "down":
        layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=3, stride=2, bias=True, dimension=3)
    else:
        layer = pcc.MinkowskiGenerativeConvolutionTranspose(cin, cout, kernel_size=int(kind[-1]), stride=2, bias=True, dimension=3)
    layer = layer.to(DEV)
    x = pcc.SparseTensor(dev(F), coordinate_map=pcc.CoordMap(dev(c), 2))
    a, b = _both(pcc, layer, x)
    assert torch.equal(a.C, b.C) and torch.equal(a.F, b.F)
    x_o = on.SparseTensor(c, F, 2)
    W, bb = layer.kernel.detach().cpu(), layer.bias.detach().cpu()
    want = on.conv(x_o, W, bb, 3, 2) if kind == "down" else on.conv_transpose_generative(x_o, W, bb, int(kind[-1]))
    idx = oc.lookup(want.C, a.C.cpu().numpy())
    assert (idx >= 0).all()
    wf = want.F[torch.from_numpy(idx)]
    assert torch.allclose(a.F.cpu(), wf, rtol=1e-4, atol=2e-5 * float(wf.abs().max()))


def test_co_is_row_order_and_batch_composition_invariant_bitwise(pcc):
    """a row's result depends on its own neighbourhood only: not on where the row sits, which group of 256 it falls into,
    or what else is in the launch (the property the decoder's reproduction of h_s rests on)"""
    torch.manual_seed(11)
    c = _coords(pcc, "shell")
    n = c.shape[0]
    F = torch.randn(n, 128)
    from pcc_amd import sparse as sp
    layer = pcc.MinkowskiConvolution(128, 128, kernel_size=3, stride=1, bias=True, dimension=3).to(DEV)
    was = sp.CONV_CO
    sp.set_conv_co(True)
    try:
        a = layer(pcc.SparseTensor(dev(F), coordinate_map=pcc.CoordMap(dev(c), 1))).F.cpu()
        perm = np.random.default_rng(5).permutation(n)
        b = layer(pcc.SparseTensor(dev(F[torch.from_numpy(perm)]), coordinate_map=pcc.CoordMap(dev(c[perm]), 1))).F.cpu()
        assert torch.equal(a[torch.from_numpy(perm)], b)
        # the same cloud as batch item 1 behind another cloud as item 0: different groups, same rows
        c2 = _coords(pcc, "sparse", seed=4)
        cc = np.concatenate([c2, c + np.array([1, 0, 0, 0], np.int32)])
        FF = torch.cat([torch.randn(c2.shape[0], 128), F])
        d = layer(pcc.SparseTensor(dev(FF), coordinate_map=pcc.CoordMap(dev(cc), 1))).F.cpu()
        assert torch.equal(d[c2.shape[0]:], a)
    finally:
        sp.set_conv_co(was)


def test_co_codes_a_frame_to_the_same_bytes(pcc):
    """the whole codec with the experimental kernel switched on: same streams, same reconstruction as the default kernels"""
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

    was = sp.CONV_CO
    try:
        sp.set_conv_co(False)
        s0, sh0, k0, r0 = run()
        sp.set_conv_co(True)
        s1, sh1, k1, r1 = run()
    finally:
        sp.set_conv_co(was)
    assert s0 == s1 and sh0 == sh1 and k0 == k1 and torch.equal(r0, r1)


def test_co_rejects_what_it_cannot_take(pcc):
    L = pcc.lib()
    from pcc_amd._lib import ptr, stream
    t = torch.zeros(256, device=DEV)
    i = torch.zeros(256, dtype=torch.int32, device=DEV)
    assert L.pcc_conv_fwd_co(ptr(t), 1, 48, ptr(t), None, ptr(i), ptr(i), ptr(i), 27, ptr(t), 1, 64, 0, None, None, stream()) < 0
    assert b"cin" in L.pcc_last_error()
    assert L.pcc_conv_fwd_co(ptr(t), 1, 64, ptr(t), None, ptr(i), ptr(i), ptr(i), 27, ptr(t), 1, 48, 0, None, None, stream()) < 0
    assert b"cout" in L.pcc_last_error()
    assert L.pcc_compact_map(ptr(i), 4, 28, ptr(i), ptr(i), ptr(i), stream()) < 0
