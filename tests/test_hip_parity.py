"""GPU parity tests: every libpcc_hip.so operator against the CPU oracle on the same seeded inputs.

Bit-exact for integer / index / byte work; fp32 convolution within the tolerance written in
each test (the HIP kernels and MKL sgemm sum in different orders).  Run on the MI355X box with
``pytest -m gpu``.
"""
import numpy as np
import pytest
import torch

from oracle import coords as oc
from oracle import nn as on
from oracle import entropy as oe

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def shell_coords(pcc, grid=48, radius=20.0, hw=0.9, batch=1, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for b in range(batch):
        p = pcc.synthetic.sphere_shell(grid, radius - 2 * b, hw)[:, :3]
        c = np.concatenate([np.full((p.shape[0], 1), b), p], axis=1).astype(np.int32)
        out.append(c)
    c = np.concatenate(out, axis=0)
    return c[rng.permutation(c.shape[0])]


def as_set(c):
    return set(map(tuple, np.asarray(c).tolist()))


def dev(a, dtype=None):
    t = torch.as_tensor(a)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


# ---------------------------------------------------------------------------------------------
def test_library_reports_gfx950(pcc):
    L = pcc.lib()
    assert L.pcc_device_count() >= 1
    import ctypes
    buf = ctypes.create_string_buffer(256)
    assert L.pcc_device_name(0, buf, 256) == 0
    assert b"gfx950" in buf.value, buf.value


def test_hash_lookup(pcc):
    c = shell_coords(pcc, batch=2)
    m = pcc.CoordMap(dev(c), 1)
    rng = np.random.default_rng(3)
    q = np.concatenate([c[rng.integers(0, c.shape[0], 2000)],
                        np.concatenate([rng.integers(0, 2, (2000, 1)), rng.integers(-5, 55, (2000, 3))], axis=1)]).astype(np.int32)
    got = m.lookup(dev(q)).cpu().numpy()
    want = oc.lookup(c, q)
    assert (got == want).all()


def test_hash_build_counts_duplicates(pcc):
    c = shell_coords(pcc)
    dup = np.concatenate([c, c[:37]])
    L = pcc.lib()
    from pcc_amd._lib import ptr, check, stream
    cap = L.pcc_hash_capacity(dup.shape[0])
    keys = torch.empty(cap, dtype=torch.int64, device=DEV)
    vals = torch.empty(cap, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    d = dev(dup)
    check(L.pcc_hash_build(ptr(d), dup.shape[0], ptr(keys), ptr(vals), cap, 1, ptr(cnt), stream()))
    assert int(cnt.item()) == 37


@pytest.mark.parametrize("ts", [1, 2, 8])
def test_stride_map(pcc, ts):
    c = shell_coords(pcc, batch=2) * np.array([1, ts, ts, ts], dtype=np.int32)
    c[:5, 1:] -= 3 * ts                                  # negative coordinates: true floor division
    m = pcc.CoordMap(dev(c), ts)
    d = m.down()
    want = oc.stride_map(c, ts)
    got = d.coords.cpu().numpy()
    assert got.shape[0] == want.shape[0] and as_set(got) == as_set(want)
    assert d.stride == 2 * ts
    # the table returned with the map indexes the output rows
    assert (d.lookup(d.coords).cpu().numpy() == np.arange(got.shape[0])).all()


@pytest.mark.parametrize("bad", [(0, 1 << 17, 5, 5), (0, 5, -(1 << 17), 5), (0, 7, 7, 130001), (1023, 1, 2, 3), (0, (1 << 18) + 3, 4, 5)])
def test_coordinates_outside_the_key_range_are_an_error_not_an_alias(pcc, model_for_range, bad):
    """18 bits per coordinate field, 10 for the batch index (csrc/common.h): a coordinate of 2^17 would wrap onto another voxel's key.
    The next coordinate-set construction reports it with the row count (PCC_COUNT_ERR_RANGE) and the Python side raises
    ValueError — from the operator, from ColorModel.compress and from decompress; the limit itself is accepted."""
    c = shell_coords(pcc, grid=24, radius=9.0)
    ok = np.concatenate([c, np.array([[0, 130000, -130000, 130000], [1022, 0, 0, 0], [0, 99999, 99999, 99999]], np.int32)])
    assert pcc.CoordMap(dev(ok), 1).down().n > 0                                  # both limits are inside
    assert pcc.CoordMap(dev(np.array([[0, 129992, -129992, 0], [1, 8, 8, 8]], np.int32)), 8).up(3).n == 54      # children at +-4: inside
    cb = np.concatenate([c, np.array([bad], np.int32)])
    with pytest.raises(ValueError, match="outside the supported range"):
        pcc.CoordMap(dev(cb), 1).down()
    if bad[0] == 0:
        pts = np.concatenate([cb[:, 1:].astype(np.float32), np.full((cb.shape[0], 3), 0.5, np.float32)], axis=1)
        qc, qf = pcc.synthetic.uniform_qmap(pts[:, :3], 0.5, 0.5)
        Q = pcc.SparseTensor(coordinates=dev(qc), features=dev(qf), device=DEV)
        with pytest.raises(ValueError, match="outside the supported range"):
            model_for_range.compress(dev(pts), Q)
        with pytest.raises(ValueError, match="outside the supported range"):
            model_for_range.decompress(coordinates=dev(np.array([[0, 0, 0, 0], list(bad)], np.int32)), strings=[[b""], [b""]],
                                       shape=[1], k=[[1], [1], [1]])


def test_children_beyond_the_key_range_are_reported(pcc):
    """a parent at the limit whose generated children step over it"""
    c = np.array([[0, 129996, 0, 0], [0, 0, 0, 0]], np.int32)
    assert pcc.CoordMap(dev(c), 4).up(2).n == 16                                  # children at +0 / +2: inside
    c[0, 1] = 130000
    with pytest.raises(ValueError, match="outside the supported range"):
        pcc.CoordMap(dev(c), 8).up(3)                                             # a child at 130004


@pytest.fixture(scope="module")
def model_for_range(pcc):
    m = pcc.synthetic.make_model(0, DEV)
    m.update()
    return m


@pytest.mark.parametrize("ksize", [2, 3])
def test_children(pcc, ksize):
    c = shell_coords(pcc, grid=24, radius=9.0) * np.array([1, 8, 8, 8], dtype=np.int32)
    m = pcc.CoordMap(dev(c), 8)
    u = m.up(ksize)
    want = oc.children(c, 8, ksize)
    got = u.coords.cpu().numpy()
    assert got.shape[0] == want.shape[0] and as_set(got) == as_set(want)
    assert u.stride == 4
    assert (u.lookup(u.coords).cpu().numpy() == np.arange(got.shape[0])).all()
    if ksize == 2:      # offset-major order: children of offset k form one contiguous block
        n = c.shape[0]
        offs = oc.kernel_offsets(2) * 4
        for k in range(8):
            assert (got[k * n:(k + 1) * n, 1:] == c[:, 1:] + offs[k]).all()


def _nbr_as_coords(nbr, in_coords):
    """Replace row ids by coordinates so tables over differently ordered maps compare."""
    out = np.full(nbr.shape + (4,), -99999, dtype=np.int64)
    hit = nbr >= 0
    out[hit] = in_coords[nbr[hit]]
    return out


def order_key(masks):
    """the execution-order key of csrc/select.hip: the 27-bit neighbour mask read with the rarest offset as the most
    significant bit (ties: lower offset first), descending"""
    masks = np.asarray(masks, dtype=np.int64) & 0x7FFFFFF
    cnt = np.array([int(((masks >> b) & 1).sum()) for b in range(27)])
    pos = [26 - sum(1 for o in range(27) if cnt[o] < cnt[b] or (cnt[o] == cnt[b] and o < b)) for b in range(27)]
    key = np.zeros(masks.shape[0], dtype=np.int64)
    for b in range(27):
        key |= ((masks >> b) & 1) << pos[b]
    return 0x7FFFFFF - key


@pytest.mark.parametrize("case", ["same", "down", "up3", "up2", "cross"])
def test_kernel_map(pcc, case):
    c = shell_coords(pcc, grid=40, radius=15.0) * np.array([1, 4, 4, 4], dtype=np.int32)
    m = pcc.CoordMap(dev(c), 4)
    if case == "same":
        out, ks, tr, step = m, 3, False, 4
        want_out = c
    elif case == "down":
        out, ks, tr, step = m.down(), 3, False, 4
    elif case == "up3":
        out, ks, tr, step = m.up(3), 3, True, 2
    elif case == "up2":
        out, ks, tr, step = m.up(2), 2, True, 2
    else:   # stride-1 conv evaluated at a different coordinate set of the same stride
        sub = c[::3] + np.array([0, 4, 0, 0], dtype=np.int32)
        out, ks, tr, step = pcc.CoordMap(dev(sub), 4), 3, False, 4
    out_c = out.coords.cpu().numpy()
    nbr, row_mask, pairs = m.kernel_map(out, ks, tr)
    nbr = nbr.cpu().numpy()
    want = oc.kernel_map(c, out_c, ks, step, transposed=tr)
    assert (_nbr_as_coords(nbr, c) == _nbr_as_coords(want, c)).all()
    K = ks ** 3
    # row masks: bit k set iff the row has a neighbour at offset k; pair count = number of hits
    want_mask = ((nbr >= 0).astype(np.int64) << np.arange(K)).sum(axis=1)
    assert (row_mask.cpu().numpy().view(np.uint32).astype(np.int64) == want_mask).all()
    assert int(pairs.item()) == int((nbr >= 0).sum())
    # execution order for the MFMA path: a permutation, the table still by output row, OR-masks per 32 positions; and the
    # permuted copy the weight-gradient kernels take
    for blk in (-1, 3):
        from pcc_amd import sparse as sp
        sp.ORDER_BLOCK_LOG2 = blk
        try:
            nbr_s, order, gm, _ = m.ordered_kernel_map(out, ks, tr)
        finally:
            sp.ORDER_BLOCK_LOG2 = -1
        order = order.cpu().numpy()
        assert sorted(order.tolist()) == list(range(nbr.shape[0]))
        assert (nbr_s.cpu().numpy() == nbr).all()                       # the kernels read row order[p] of the one table
        if blk < 0:
            nbr_p, order_p, _, _ = m.position_ordered_table(out, ks, tr)
            assert (order_p.cpu().numpy() == order).all() and (nbr_p.cpu().numpy() == nbr[order]).all()
        gm = gm.cpu().numpy().view(np.uint32)
        sm = want_mask[order]
        for g in range(gm.shape[0]):
            assert int(gm[g]) == int(np.bitwise_or.reduce(sm[g * 32:(g + 1) * 32]))
        if blk < 0:                                 # sorted by the rare-offsets-first key
            assert (np.diff(order_key(want_mask)[order]) >= 0).all()


CONV_SHAPES = [
    # cin, cout, ksize  (every (C_in, C_out) pair of configs/Ours.yaml)
    (128, 128, 3), (64, 128, 3), (128, 256, 3), (128, 64, 3), (64, 64, 3), (64, 32, 3), (32, 3, 3),
    (128, 2, 3), (64, 2, 3), (192, 256, 3), (128, 1, 3), (128, 128, 1), (16, 16, 1),
    (4, 64, 3), (4, 2, 3), (2, 2, 3), (2, 128, 3), (2, 16, 3), (16, 2, 3), (1, 1, 3),
    # the remaining cin / 32 instantiations of the MFMA kernel (odd chunk counts alternate the LDS buffer parity)
    (96, 64, 3), (160, 32, 3), (224, 128, 3), (256, 128, 3),
    # thin widths of a hyperprior over a narrow tensor (C * 3 / 2: the two-hyperprior variant's q-map model, tests/test_two_hyperprior.py)
    (3, 4, 3), (6, 8, 3), (12, 16, 3), (24, 32, 3), (8, 3, 3),
]


@pytest.mark.parametrize("cin,cout,ksize", CONV_SHAPES)
def test_conv_forward_matches_oracle(pcc, cin, cout, ksize):
    from pcc_amd import sparse as sp
    torch.manual_seed(cin * 1000 + cout)
    c = shell_coords(pcc, grid=40, radius=15.0)
    n = c.shape[0]
    m = pcc.CoordMap(dev(c), 1)
    layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=ksize, stride=1, bias=True, dimension=3)
    with torch.no_grad():
        layer.kernel.normal_(0, 1.0 / np.sqrt(cin * 10))
        layer.bias.normal_(0, 0.1)
    layer = layer.to(DEV)
    F = torch.randn(n, cin)
    film = torch.randn(n, 2 * cout)
    res = torch.randn(n, cout)
    W = layer.kernel.detach().cpu()
    b = layer.bias.detach().cpu()
    if ksize == 1:
        base = F @ W + b
    else:
        nbr = oc.kernel_map(c, c, ksize, 1)
        base = on._apply_conv(F, W, b, nbr, n)
    x = pcc.SparseTensor(dev(F), coordinate_map=m)
    scale = float(base.abs().max())
    # plain
    got = layer(x).F.cpu()
    assert torch.allclose(got, base, rtol=1e-4, atol=2e-5 * scale), float((got - base).abs().max())
    # fused epilogues: FiLM (no activation), ReLU + residual, LeakyReLU
    got = layer(x, film=dev(film)).F.cpu()
    want = base * film[:, :cout] + film[:, cout:]
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-4 * scale)
    got = layer(x, act=sp.ACT_RELU, residual=dev(res)).F.cpu()
    assert torch.allclose(got, torch.relu(base) + res, rtol=1e-4, atol=2e-5 * scale)
    got = layer(x, act=sp.ACT_LRELU).F.cpu()
    assert torch.allclose(got, torch.nn.functional.leaky_relu(base, 0.01), rtol=1e-4, atol=2e-5 * scale)


@pytest.mark.parametrize("kind", ["down", "up3", "up2"])
def test_strided_and_transposed_conv(pcc, kind):
    torch.manual_seed(7)
    c = shell_coords(pcc, grid=40, radius=15.0) * np.array([1, 2, 2, 2], dtype=np.int32)
    n = c.shape[0]
    cin, cout = 64, 128
    F = torch.randn(n, cin)
    x_o = on.SparseTensor(c, F, 2)
    if kind == "down":
        layer = pcc.MinkowskiConvolution(cin, cout, kernel_size=3, stride=2, bias=True, dimension=3)
    else:
        layer = pcc.MinkowskiGenerativeConvolutionTranspose(cin, cout, kernel_size=int(kind[-1]), stride=2, bias=True, dimension=3)
    layer = layer.to(DEV)
    W, b = layer.kernel.detach().cpu(), layer.bias.detach().cpu()
    want = on.conv(x_o, W, b, 3, 2) if kind == "down" else on.conv_transpose_generative(x_o, W, b, int(kind[-1]))
    got = layer(pcc.SparseTensor(dev(F), coordinate_map=pcc.CoordMap(dev(c), 2)))
    assert got.map.stride == want.stride
    gc, gf = got.C.cpu().numpy(), got.F.cpu()
    idx = oc.lookup(want.C, gc)
    assert (idx >= 0).all() and gc.shape[0] == want.C.shape[0]
    wf = want.F[torch.from_numpy(idx)]
    assert torch.allclose(gf, wf, rtol=1e-4, atol=2e-5 * float(wf.abs().max()))


def test_conv_is_row_order_invariant_bitwise(pcc):
    """The property the reference's Sorted* shims exist for (entropy_models.py:12-102): a row's
    result must not depend on where the row sits."""
    torch.manual_seed(11)
    c = shell_coords(pcc, grid=40, radius=15.0)
    n = c.shape[0]
    F = torch.randn(n, 128)
    layer = pcc.MinkowskiConvolution(128, 128, kernel_size=3, stride=1, bias=True, dimension=3).to(DEV)
    a = layer(pcc.SparseTensor(dev(F), coordinate_map=pcc.CoordMap(dev(c), 1))).F.cpu()
    perm = np.random.default_rng(5).permutation(n)
    b = layer(pcc.SparseTensor(dev(F[torch.from_numpy(perm)]), coordinate_map=pcc.CoordMap(dev(c[perm]), 1))).F.cpu()
    assert torch.equal(a[torch.from_numpy(perm)], b)


_PATH_SCRIPT = r"""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
import pcc_amd
torch.manual_seed(3)
g = np.stack(np.meshgrid(*[np.arange(24)] * 3, indexing="ij"), -1).reshape(-1, 3)
keep = np.abs(np.linalg.norm(g - 11.5, axis=1) - 9.0) < 1.2
c = np.concatenate([np.zeros((int(keep.sum()), 1), np.int32), g[keep].astype(np.int32)], 1)
h = hashlib.sha256()
for cin, cout in [(128, 128), (64, 64), (96, 32), (128, 256)]:
    layer = pcc_amd.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, bias=True, dimension=3).to("cuda:0")
    F = torch.randn(c.shape[0], cin, device="cuda:0")
    with torch.no_grad():
        out = layer(pcc_amd.SparseTensor(F, coordinate_map=pcc_amd.CoordMap(torch.from_numpy(c).to("cuda:0"), 1))).F
    h.update(out.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
"""


def test_conv_64bit_addressed_fallback_is_bit_identical(pcc):
    """Operands of 4 GiB and more take conv_mfma_kernel (64-bit LDS-DMA addressing) instead of the
    buffer-addressed kernel; PCC_CONV_PATH=global forces it.  Same accumulation order -> same bits."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for path in ("buffer", "global"):
        env = dict(os.environ, PCC_CONV_PATH=path)
        r = subprocess.run([sys.executable, "-c", _PATH_SCRIPT.format(root=root)], env=env, capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0])
    assert digests[0] == digests[1]


def test_conv_on_operands_beyond_4gib(pcc):
    """9 M rows x 128 channels = 4.6 GB of features: past the 32-bit buffer offsets, so the dispatch must take
    the 64-bit-addressed kernel by itself.  Checked on sampled rows against a direct torch evaluation."""
    rng = np.random.default_rng(33)
    n = 9_000_000
    flat = rng.choice(512 ** 3, size=n, replace=False)
    c = np.stack([np.zeros(n, np.int64), flat // (512 * 512), (flat // 512) % 512, flat % 512], axis=1).astype(np.int32)
    m = pcc.CoordMap(dev(c), 1, nbatch=1)
    torch.manual_seed(8)
    F = torch.randn(n, 128, device=DEV)
    assert F.numel() * 4 > 0xFFFFF000
    layer = pcc.MinkowskiConvolution(128, 64, kernel_size=3, stride=1, bias=True, dimension=3).to(DEV)
    with torch.no_grad():
        out = layer(pcc.SparseTensor(F, coordinate_map=m)).F
        nbr, _, _ = m.kernel_map(m, 3)
        rows = torch.from_numpy(rng.integers(0, n, size=4096)).to(DEV)
        W, b = layer.kernel, layer.bias
        want = b.expand(rows.numel(), -1).clone()
        for k in range(27):
            idx = nbr[rows, k].long()
            ok = (idx >= 0).unsqueeze(1).float()
            want += (F[idx.clamp(min=0)] * ok) @ W[k]
    got = out[rows]
    assert torch.allclose(got, want, rtol=1e-4, atol=2e-5 * float(want.abs().max()))
    del F, out


@pytest.mark.parametrize("cout,K,n_out", [(1, 27, 1000), (3, 27, 777), (2, 27, 256), (4, 27, 513), (1, 8, 300), (1, 27, 1)])
def test_gather_sum_matches_sequential_sum_bitwise(pcc, cout, K, n_out):
    """second half of the narrow-head path (occupancy logit, colour head): out[j, c] = bias[c] + sum over present
    offsets k, ascending, of scores[nbr[j, k], k * cout + c] — fp32 adds in a fixed order, so exact"""
    from pcc_amd._lib import lib, check, ptr, stream
    rng = np.random.default_rng(cout * 100 + K)
    n_in = 900
    nbr = rng.integers(0, n_in, size=(n_out, K)).astype(np.int32)
    nbr[rng.random((n_out, K)) < 0.45] = -1
    nbr[0, :] = -1                                            # a row without neighbours
    scores = rng.standard_normal((n_in, K * cout)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    want = np.zeros((n_out, cout), np.float32)
    for k in range(K):
        got_k = scores[np.maximum(nbr[:, k], 0), k * cout:(k + 1) * cout]
        want = np.where((nbr[:, k] >= 0)[:, None], want + got_k, want).astype(np.float32)
    want = np.maximum(want + bias[None, :], 0).astype(np.float32)           # act 1 = ReLU
    out = torch.empty((n_out, cout), dtype=torch.float32, device=DEV)
    d_scores, d_nbr, d_bias = dev(scores), dev(nbr), dev(bias)
    check(lib().pcc_gather_sum_fwd(ptr(d_scores), K * cout, ptr(d_nbr), K, cout, ptr(d_bias), ptr(out), n_out, 1, stream()))
    assert np.array_equal(out.cpu().numpy(), want)
    # a neighbour table that is only 4-byte aligned (the K = 27 kernel's 16-byte staging loads must not be used)
    flat = torch.empty(n_out * K + 1, dtype=torch.int32, device=DEV)
    flat[1:] = d_nbr.reshape(-1)
    shifted = flat[1:]
    assert shifted.data_ptr() % 16 == 4
    out2 = torch.empty_like(out)
    check(lib().pcc_gather_sum_fwd(ptr(d_scores), K * cout, shifted.data_ptr(), K, cout, ptr(d_bias), ptr(out2), n_out, 1, stream()))
    assert torch.equal(out2, out)


def test_gather_scatter_compact(pcc):
    from pcc_amd import sparse as sp
    rng = np.random.default_rng(2)
    n, cch = 5000, 6
    src = torch.randn(n, cch)
    idx = rng.integers(-1, n, 7000).astype(np.int32)
    got = sp.gather_rows(dev(src), dev(idx)).cpu()
    want = torch.zeros(7000, cch)
    hit = idx >= 0
    want[torch.from_numpy(hit)] = src[torch.from_numpy(idx[hit].astype(np.int64))]
    assert torch.equal(got, want)
    base = torch.randn(7000, cch)
    acc = dev(base.clone())
    sp.gather_rows(dev(src), dev(idx), out=acc, accumulate=True)
    assert torch.equal(acc.cpu(), base + want)
    perm = rng.permutation(n).astype(np.int32)
    sc = sp.scatter_rows(dev(src), dev(perm), n).cpu()
    assert torch.equal(sc[torch.from_numpy(perm.astype(np.int64))], src)
    mask = rng.random(n) < 0.3
    coords = np.concatenate([np.zeros((n, 1)), rng.integers(0, 100, (n, 3))], axis=1).astype(np.int32)
    oc_, of_, ni, m = sp.compact_rows(dev(mask.astype(np.uint8)), dev(coords), dev(src), want_index=True)
    assert m == int(mask.sum())
    assert (oc_.cpu().numpy() == coords[mask]).all() and torch.equal(of_.cpu(), src[torch.from_numpy(mask)])
    ni = ni.cpu().numpy()
    assert (ni[~mask] == -1).all() and (ni[mask] == np.arange(m)).all()
    # empty / full masks
    assert sp.compact_rows(dev(np.zeros(n, np.uint8)), dev(coords), dev(src))[3] == 0
    assert sp.compact_rows(dev(np.ones(n, np.uint8)), dev(coords), dev(src))[3] == n


@pytest.mark.parametrize("nbatch", [1, 3])
def test_topk_mask(pcc, nbatch):
    from pcc_amd import sparse as sp
    from oracle.codec import topk_mask
    rng = np.random.default_rng(4)
    c = shell_coords(pcc, batch=nbatch, seed=9)
    n = c.shape[0]
    logits = rng.normal(size=(n, 5)).astype(np.float32)
    logits[rng.integers(0, n, n // 3), 0] = 0.25            # many exact ties -> coordinate-key tie break
    logits[rng.integers(0, n, 50), 0] = -0.0
    counts = [int((c[:, 0] == b).sum()) for b in range(nbatch)]
    for ks in ([cnt // 3 for cnt in counts], [1] * nbatch, [cnt + 5 for cnt in counts], [0] * nbatch, counts):
        got = sp.topk_mask(dev(logits), dev(c), ks, nbatch).cpu().numpy().astype(bool)
        want = topk_mask(on.SparseTensor(c, torch.from_numpy(logits), 1), ks)
        assert (got == want).all(), (ks, int(got.sum()), int(want.sum()))


@pytest.mark.parametrize("ties", ["some", "none", "all", "two_values"])
def test_topk_mask_of_a_large_single_item(pcc, ties):
    """beyond the one-workgroup size (32,768 rows), one batch item: four fused (histogram + pick) launches over the logit bytes and
    ONE tail launch for the eight bytes of the coordinate tie-break — which only has work when rows with exactly the boundary logit
    straddle the k-th place ("some", "two_values") or every row does ("all")"""
    from pcc_amd import sparse as sp
    from oracle.codec import topk_mask
    rng = np.random.default_rng(12)
    n = 150_001
    flat = rng.choice(200 ** 3, size=n, replace=False)
    c = np.stack([np.zeros(n, np.int64), flat // 40000 - 100, (flat // 200) % 200, flat % 200], axis=1).astype(np.int32)
    logits = rng.normal(size=(n, 2)).astype(np.float32)
    if ties == "some":
        logits[rng.integers(0, n, n // 3), 0] = 0.25
        logits[rng.integers(0, n, 500), 0] = -0.0
        logits[rng.integers(0, n, 500), 0] = 0.0
    elif ties == "all":
        logits[:, 0] = -1.5
    elif ties == "two_values":
        logits[:, 0] = np.where(rng.random(n) < 0.5, 2.0, -2.0).astype(np.float32)
    n_quarter = int((logits[:, 0] == 0.25).sum())
    n_above = int((logits[:, 0] > 0.25).sum())
    for k in (n // 3, 1, n + 5, 0, n, n_above + max(1, n_quarter // 2), n_above, n_above + 1, n - 1):
        got = sp.topk_mask(dev(logits), dev(c), [k], 1).cpu().numpy().astype(bool)
        want = topk_mask(on.SparseTensor(c, torch.from_numpy(logits), 1), [k])
        assert int(got.sum()) == min(k, n), (ties, k, int(got.sum()))
        assert (got == want).all(), (ties, k, int((got != want).sum()))


def test_sort_permutation(pcc):
    c = shell_coords(pcc, batch=3, seed=1)
    perm = pcc.CoordMap(dev(c), 1).sort_permutation().cpu().numpy()
    assert (perm == oc.sort_order(c)).all()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1023, 4097, 16384, 16385, 300_001, 524_288, 524_289])
def test_canonical_sort_all_sizes(pcc, n):
    """the hand-written radix sort behind utils.sort_tensor / sort_points (utils.py:155-204): the single-workgroup
    shape (n <= 16,384), the count / self-prefixed scatter shape (n <= 524,288) and the count / row-scan / scatter shape above it, ragged tails, negative coordinates, several
    batch items; the permutation must equal the oracle's exact lexicographic (b, x, y, z) order"""
    rng = np.random.default_rng(n)
    side = max(4, int(round((4 * n) ** (1 / 3))) + 2)
    flat = rng.choice(side ** 3 * 3, size=n, replace=False)          # distinct (b, x, y, z)
    b, rem = flat // side ** 3, flat % side ** 3
    c = np.stack([b, rem // side ** 2 - side // 2, (rem // side) % side - 3, rem % side], axis=1).astype(np.int32)
    perm = pcc.CoordMap(dev(c), 1).sort_permutation().cpu().numpy()
    assert (perm == oc.sort_order(c)).all()


@pytest.mark.parametrize("n,K", [(5, 27), (257, 27), (16_385, 8), (98_500, 27), (600_003, 27)])
def test_mask_order_is_the_stable_sort_of_the_keys(pcc, n, K):
    """pcc_order_rows_by_mask on synthetic masks (few distinct values -> long runs of equal keys, the real
    distribution): `order` must be THE stable ascending sort of the rare-offsets-first key — ranks come from
    ballots and per-wave counters, never from atomics, so equal keys keep their row order — with the permuted
    table and the 32-position OR-masks consistent with it"""
    from pcc_amd._lib import ptr, check, stream
    L = pcc.lib()
    rng = np.random.default_rng(n + K)
    palette = rng.integers(1, 1 << K, size=37, dtype=np.int64)
    mask = palette[rng.integers(0, 37, size=n)]
    mask[rng.integers(0, n, size=max(1, n // 50))] = rng.integers(0, 1 << K, size=max(1, n // 50))
    nbr = np.where((mask[:, None] >> np.arange(K)) & 1, rng.integers(0, n, size=(n, K)), -1).astype(np.int32)
    d_mask, d_nbr = dev(mask.astype(np.uint32).view(np.int32)), dev(nbr)
    order = torch.empty(n, dtype=torch.int32, device=DEV)
    nbr_s = torch.empty_like(d_nbr)
    gm = torch.empty((n + 31) // 32, dtype=torch.int32, device=DEV)
    nbytes = L.pcc_order_scratch_bytes(n)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    check(L.pcc_order_rows_by_mask(ptr(d_mask), None, n, -1, 1, ptr(order), ptr(gm), ptr(scratch), nbytes, stream()))
    check(L.pcc_permute_map_rows(ptr(d_nbr), ptr(order), n, K, ptr(nbr_s), stream()))      # the training path's permuted copy
    key = order_key(mask)
    want = np.argsort(key, kind="stable")
    got = order.cpu().numpy()
    assert (got == want).all()
    assert (nbr_s.cpu().numpy() == nbr[want]).all()
    g = gm.cpu().numpy().view(np.uint32).astype(np.int64)
    sm = np.concatenate([mask[want], np.zeros((-n) % 32, np.int64)]).reshape(-1, 32)
    assert (g == np.bitwise_or.reduce(sm, axis=1)).all()


def test_count_per_batch(pcc):
    c = shell_coords(pcc, batch=3, seed=1)
    assert pcc.CoordMap(dev(c), 1).count_per_batch() == oc.count_per_batch(c)


# ---------------------------------------------------------------------------------------------
def test_entropy_bottleneck_kernels(pcc, seeded_state_dict, oracle_codec):
    model = pcc.synthetic.make_model(0, DEV)
    model.update()
    eb = model.entropy_model.entropy_bottleneck
    o = oracle_codec.eb
    # tables are built by the same published algorithm on both sides: bit-exact
    cdf, cdf_len, off = eb.tables()
    assert (cdf == o.cdf).all() and (cdf_len == o.cdf_length).all() and (off == o.offset).all()
    torch.manual_seed(0)
    z = torch.randn(300, 128) * 4
    z[0, :] = 40.0          # beyond the table: escape-coded
    z[1, :] = -37.5         # half-way cases: round-half-even
    zin = z.t().unsqueeze(0).contiguous()
    want_hat, want_lik = o.forward_eval(zin)
    got_hat, got_lik = eb(dev(zin))
    assert torch.equal(got_hat.cpu(), want_hat)
    assert torch.allclose(got_lik.cpu(), want_lik, rtol=2e-4, atol=1e-9)
    strings = eb.compress(dev(zin))
    assert strings[0] == o.compress(zin)[0]                   # byte-identical stream
    back = eb.decompress(strings, [300]).cpu()
    assert torch.equal(back, want_hat)


def test_gaussian_conditional_kernels(pcc, oracle_codec):
    model = pcc.synthetic.make_model(0, DEV)
    model.update()
    gc = model.entropy_model.gaussian_conditional
    o = oracle_codec.gc
    cdf, cdf_len, off = gc.tables()
    assert (cdf == o.cdf).all() and (cdf_len == o.cdf_length).all() and (off == o.offset).all()
    torch.manual_seed(1)
    n, c = 700, 128
    y = torch.randn(n, c) * 3
    scales = torch.exp(torch.randn(n, c) * 2)                 # spans the whole table, incl. < 0.11
    scales[0] = -1.0
    scales[1] = torch.tensor(o.scale_table[5].item())          # exactly on a threshold
    means = torch.randn(n, c)
    y[2] = 1e4                                                # long escapes
    params = torch.cat([scales, means], dim=1)
    s3, m3, y3 = (t.t().unsqueeze(0).contiguous() for t in (scales, means, y))
    want_idx = o.build_indexes(s3)
    got_idx = gc.build_indexes(dev(s3)).cpu()
    assert torch.equal(got_idx, want_idx)
    want_hat, want_lik = o.forward_eval(y3, s3, m3)
    got_hat, got_lik = gc(dev(y3), dev(s3), means=dev(m3))
    assert torch.equal(got_hat.cpu(), want_hat)
    # difference of two erfc values: absolute error is a few ulp of 1.0
    assert torch.allclose(got_lik.cpu(), want_lik, rtol=5e-4, atol=3e-7)
    strings = gc.compress_features(dev(y), dev(params))
    assert strings[0] == o.compress(y3, want_idx, m3)[0]
    back = gc.decompress_features(strings, dev(params), c).cpu()
    assert torch.equal(back.t().unsqueeze(0), o.decompress(strings, want_idx, m3))


def test_gaussian_conditional_packed_planes(pcc, oracle_codec):
    """pcc_gc_encode_prep_packed: int16 symbols / uint8 indexes in stream order (rows permuted on the way) equal the
    int32 planes of pcc_gc_encode_prep after index_select; the overflow word reports symbols beyond int16; the packed
    compress path codes the same bytes as the int32 path, and its decode returns the same y_hat"""
    from pcc_amd._lib import ptr, check, stream
    model = pcc.synthetic.make_model(0, DEV)
    model.update()
    gc = model.entropy_model.gaussian_conditional
    L = pcc.lib()
    torch.manual_seed(1)
    n, c = 777, 128
    y = (torch.randn(n, c) * 6).to(DEV)
    params = torch.cat([torch.rand(n, c) * 3 + 0.05, torch.randn(n, c)], dim=1).to(DEV)
    perm = torch.randperm(n).to(torch.int32).to(DEV)
    sym32, idx32 = gc.encode_prep(y, params)
    sym32, idx32 = sym32.index_select(1, perm.long()), idx32.index_select(1, perm.long())
    cn = c * n
    sym16 = torch.empty(cn, dtype=torch.int16, device=DEV)
    idx8 = torch.empty(cn, dtype=torch.uint8, device=DEV)
    flag = torch.ones(1, dtype=torch.int32, device=DEV)
    table = gc.scale_table.to(DEV).contiguous()
    check(L.pcc_gc_encode_prep_packed(ptr(y), ptr(params), n, c, ptr(table), table.numel(), ptr(perm), ptr(sym16), ptr(idx8), ptr(flag),
                                      stream()))
    assert int(flag.item()) == 0
    assert torch.equal(sym16.reshape(c, n).to(torch.int32), sym32) and torch.equal(idx8.reshape(c, n).to(torch.int32), idx32)
    y_big = y.clone()
    y_big[5, 7] = 1e6
    check(L.pcc_gc_encode_prep_packed(ptr(y_big), ptr(params), n, c, ptr(table), table.numel(), ptr(perm), ptr(sym16), ptr(idx8),
                                      ptr(flag), stream()))
    assert int(flag.item()) == 1
    # the two compress paths: same bytes (the big symbol takes the int32 planes inside compress_features_begin)
    for feats in (y, y_big):
        a = gc.compress_features_begin(feats, params, perm)()
        b = gc.compress_features(feats, params, perm)
        assert a == b
        p_sorted = params.index_select(0, perm.long())
        back = gc.decompress_features(a, p_sorted, c)
        want = torch.round(feats.index_select(0, perm.long()) - p_sorted[:, c:]) + p_sorted[:, c:]
        assert torch.equal(back, want)


_THIN_SCRIPT = r"""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
import pcc_amd
from pcc_amd import sparse as sp
torch.manual_seed(5)
torch.set_grad_enabled(False)          # the inference kernels (with gradients on, trainable layers take the autograd path)
rng = np.random.default_rng(2)
p = pcc_amd.synthetic.sphere_shell(64, 27.0, 0.9)[:, :3]
c = np.concatenate([np.zeros((p.shape[0], 1)), p], axis=1).astype(np.int32)
c = c[rng.permutation(c.shape[0])]
sub = c[rng.random(c.shape[0]) < 0.4]
h = hashlib.sha256()
for coords in (c, sub, c[:70], c[:1]):
    n = coords.shape[0]
    m = pcc_amd.CoordMap(torch.from_numpy(coords).cuda(), 1)
    for cin, cout in ((2, 128), (4, 64), (2, 64), (16, 32), (1, 32), (8, 64), (4, 96)):
        layer = pcc_amd.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, bias=True, dimension=3).cuda()
        x = pcc_amd.SparseTensor(torch.randn(n, cin).cuda(), coordinate_map=m)
        film, res = torch.randn(n, 2 * cout).cuda(), torch.randn(n, cout).cuda()
        for kw in ({{}}, dict(act=sp.ACT_RELU, residual=res), dict(act=sp.ACT_LRELU, film=film, residual=res)):
            h.update(layer(x, **kw).F.cpu().numpy().tobytes())
    # strided and transposed maps (K = 27 and K = 8) with thin inputs
    m2 = pcc_amd.CoordMap(torch.from_numpy(coords * np.array([1, 2, 2, 2], np.int32)).cuda(), 2)
    x2 = pcc_amd.SparseTensor(torch.randn(n, 2).cuda(), coordinate_map=m2)
    for layer in (pcc_amd.MinkowskiConvolution(2, 64, kernel_size=3, stride=2, bias=True, dimension=3),
                  pcc_amd.MinkowskiGenerativeConvolutionTranspose(2, 32, kernel_size=2, stride=2, bias=True, dimension=3),
                  pcc_amd.MinkowskiGenerativeConvolutionTranspose(2, 64, kernel_size=3, stride=2, bias=True, dimension=3)):
        h.update(layer.cuda()(x2).F.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
"""


def test_thin_im2col_mfma_equals_thin_kernel_bitwise():
    """thin inputs, wide outputs (2 -> 128, 4 -> 64, ...): im2col + one kernel_size-1 MFMA convolution (the default) against the
    scalar conv_thin_kernel (PCC_THIN_IM2COL=0): the fp32 MFMA is the same fused multiply-add chain as v_fma_f32 and the matrix
    columns are laid out in the order the MFMA loop contracts them, so the same bits — over dense / sparse / 70-row / one-row
    sets, every epilogue, strided and transposed maps"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for flag in ("1", "0"):
        env = dict(os.environ, PCC_THIN_IM2COL=flag)
        r = subprocess.run([sys.executable, "-c", _THIN_SCRIPT.format(root=root)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0])
    assert digests[0] == digests[1]
