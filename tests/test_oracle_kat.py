"""Known-answer tests pinning the CPU oracle (SURVEY.md §8c "golden vectors the build must create").

The reference ships no tests or fixtures for this path; these are the hand-derivable integer
facts and the structural facts the reference itself fixes.
"""
import struct

import numpy as np
import pytest
import torch

from oracle import coords as oc
from oracle import rans as crans
from oracle import rans_py
from oracle.codec import pack_container, unpack_container, count_bits


def C(rows):
    return np.asarray(rows, dtype=np.int32)


def test_pack_orders_lexicographically():
    rng = np.random.default_rng(0)
    c = np.concatenate([rng.integers(0, 3, (500, 1)), rng.integers(-40, 1100, (500, 3))], axis=1)
    order = oc.sort_order(c)
    ref = sorted(range(500), key=lambda i: tuple(c[i]))
    assert [tuple(c[i]) for i in order] == [tuple(c[i]) for i in ref]
    assert (oc.unpack(oc.pack(c)) == c).all()
    # same order as the reference's radix-1e5 int64 key (utils.py:170-171) for in-range coords
    w = np.array([10 ** 15, 10 ** 10, 10 ** 5, 1], dtype=np.int64)
    ref_key = (c.astype(np.int64) * w).sum(1)
    pos = c[:, 1:].min(1) >= 0
    assert (np.argsort(ref_key[pos], kind="stable") == np.argsort(oc.pack(c[pos]), kind="stable")).all()


def test_kernel_offset_enumeration():
    o3 = oc.kernel_offsets(3)
    assert o3.shape == (27, 3)
    assert tuple(o3[0]) == (-1, -1, -1) and tuple(o3[13]) == (0, 0, 0) and tuple(o3[26]) == (1, 1, 1)
    assert tuple(o3[1]) == (0, -1, -1)          # x fastest
    assert tuple(o3[3]) == (-1, 0, -1) and tuple(o3[9]) == (-1, -1, 0)
    o2 = oc.kernel_offsets(2)
    assert [tuple(o) for o in o2] == [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1)]


def test_three_voxel_kernel_map_by_hand():
    c = C([[0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 2, 0]])
    nbr = oc.kernel_map(c, c, 3, 1)
    # row 0 sees itself at the centre (k=13) and row 1 at +x (k=14); nothing else
    assert nbr[0, 13] == 0 and nbr[0, 14] == 1 and (np.delete(nbr[0], [13, 14]) == -1).all()
    assert nbr[1, 13] == 1 and nbr[1, 12] == 0 and (np.delete(nbr[1], [12, 13]) == -1).all()
    assert nbr[2, 13] == 2 and (np.delete(nbr[2], 13) == -1).all()


def test_stride_map_floor_semantics():
    c = C([[0, -1, 0, 3], [0, 1, 1, 2], [0, 2, 5, 7], [1, 1, 1, 1]])
    out = oc.stride_map(c, 1)
    assert sorted(map(tuple, out)) == [(0, -2, 0, 2), (0, 0, 0, 2), (0, 2, 4, 6), (1, 0, 0, 0)]
    out4 = oc.stride_map(C([[0, 4, 8, 12], [0, 6, 8, 12]]), 2)   # stride-2 tensor -> stride 4
    assert sorted(map(tuple, out4)) == [(0, 4, 8, 12)]


def test_children_sets():
    p = C([[0, 8, 8, 8]])
    k2 = oc.children(p, 8, 2)
    assert k2.shape[0] == 8 and set(map(tuple, k2)) == {(0, 8 + 4 * a, 8 + 4 * b, 8 + 4 * c) for a in (0, 1) for b in (0, 1) for c in (0, 1)}
    k3 = oc.children(p, 8, 3)
    assert k3.shape[0] == 27 and k3[:, 1:].min() == 4 and k3[:, 1:].max() == 12
    two = oc.children(C([[0, 8, 8, 8], [0, 16, 8, 8]]), 8, 3)
    assert two.shape[0] == 27 + 27 - 9          # the x = 12 plane is shared
    # transposed kernel map: child c' = c + off_k * half  <=>  parent = c' - off_k * half
    nbr = oc.kernel_map(p, k3, 3, 4, transposed=True)
    for j, cc in enumerate(k3):
        ks = np.nonzero(nbr[j] >= 0)[0]
        assert len(ks) == 1
        off = oc.kernel_offsets(3)[ks[0]] * 4
        assert tuple(p[0, 1:] + off) == tuple(cc[1:])


def test_config1_sphere_sizes(pcc):
    pts = pcc.synthetic.sphere_shell(**pcc.synthetic.CONFIG1)
    assert pts.shape == (4904, 6)
    c = np.concatenate([np.zeros((pts.shape[0], 1)), pts[:, :3]], axis=1).astype(np.int32)
    sizes = []
    for ts in (1, 2, 4, 8, 16):
        c = oc.stride_map(c, ts)
        sizes.append(c.shape[0])
    assert sizes == [1136, 320, 56, 8, 1]


@pytest.mark.parametrize("pmf, expected", [
    ([0.5, 0.25, 0.25], [0, 32768, 49152, 65536]),
    ([1.0, 0.0], [0, 65535, 65536]),                   # zero-width bin repaired by stealing
    ([0.25, 0.0, 0.5, 0.25], [0, 16383, 16384, 49152, 65536]),
])
def test_pmf_to_quantized_cdf_hand_checked(pmf, expected):
    assert crans.pmf_to_quantized_cdf(pmf).tolist() == expected
    assert rans_py.pmf_to_quantized_cdf(pmf) == expected


def _toy_tables():
    cdfs = np.zeros((2, 7), dtype=np.int32)
    cdfs[0, :5] = crans.pmf_to_quantized_cdf([0.6, 0.2, 0.1, 0.1])      # 3 symbols + escape
    cdfs[1, :7] = crans.pmf_to_quantized_cdf([0.1, 0.2, 0.4, 0.2, 0.05, 0.05])
    sizes = np.array([5, 7], dtype=np.int32)
    offsets = np.array([-1, -2], dtype=np.int32)
    return cdfs, sizes, offsets


def test_rans_twins_agree_and_round_trip():
    cdfs, sizes, offsets = _toy_tables()
    rng = np.random.default_rng(1)
    n = 4000
    idx = rng.integers(0, 2, n).astype(np.int32)
    sym = rng.integers(-3, 4, n).astype(np.int32)
    sym[::97] = rng.integers(-70000, 70000, sym[::97].shape)           # force long escapes
    data_c = crans.encode_with_indexes(sym, idx, cdfs, sizes, offsets)
    data_p = rans_py.encode_with_indexes(sym.tolist(), idx.tolist(), cdfs.tolist(), sizes.tolist(), offsets.tolist())
    assert data_c == data_p
    assert len(data_c) % 4 == 0
    assert (crans.decode_with_indexes(data_c, idx, cdfs, sizes, offsets) == sym).all()
    assert rans_py.decode_with_indexes(data_c, idx.tolist(), cdfs.tolist(), sizes.tolist(), offsets.tolist()) == sym.tolist()


def test_rans_empty_and_single():
    cdfs, sizes, offsets = _toy_tables()
    e = np.zeros(0, dtype=np.int32)
    data = crans.encode_with_indexes(e, e, cdfs, sizes, offsets)
    assert data == struct.pack("<II", 1 << 31, 0)                      # flushed initial state only
    assert rans_py.encode_with_indexes([], [], cdfs.tolist(), sizes.tolist(), offsets.tolist()) == data
    one = crans.encode_with_indexes([0], [0], cdfs, sizes, offsets)
    assert crans.decode_with_indexes(one, [0], cdfs, sizes, offsets).tolist() == [0]


def test_container_header_is_28_bytes_big_endian():
    blob = pack_container([1184], b"\x01\x02\x03", [[b"yy"], [b"z"]], [[72752], [265512], [850824]])
    assert blob[:28] == struct.pack(">7i", 1184, 3, 2, 1, 72752, 265512, 850824)     # model.py:243-250
    gp, strings, shape, k = unpack_container(blob)
    assert gp == b"\x01\x02\x03" and strings == [[b"yy"], [b"z"]] and shape == [1184]
    assert k == [[72752], [265512], [850824]]
    assert count_bits(strings) == 24


def test_parameter_count_matches_readme(pcc):
    model = pcc.ColorModel(pcc.synthetic.OURS_CONFIG)
    assert sum(p.numel() for p in model.parameters()) == 31_469_942            # README.md:125 (120.1 MB)
    names = [n for n, _ in model.named_parameters()]
    assert [n for n in names if n.endswith(".quantiles")] == ["entropy_model.entropy_bottleneck.quantiles"]


def test_gaussian_tables(oracle_codec):
    gc = oracle_codec.gc
    assert gc.scale_table.shape[0] == 64
    assert abs(float(gc.scale_table[0]) - 0.11) < 1e-6 and abs(float(gc.scale_table[-1]) - 256.0) < 1e-3
    # pmf_center = ceil(scale * 6.1094...)  -> offsets / lengths are pure math
    mult = 6.109410205
    center = np.ceil(gc.scale_table.numpy() * mult).astype(np.int32)
    assert (gc.offset == -center).all() and (gc.cdf_length == 2 * center + 3).all()
    for i in range(64):
        row = gc.cdf[i, : gc.cdf_length[i]]
        assert row[0] == 0 and row[-1] == 65536 and (np.diff(row) > 0).all()
    idx = gc.build_indexes(torch.tensor([[[0.01, 0.11, 0.12, 255.0, 256.0, 1e9]]]))
    assert idx.reshape(-1).tolist() == [0, 0, 1, 63, 63, 63]
