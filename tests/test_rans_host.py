"""Host entry points of the C-ABI (no GPU needed): the product's range coder against the oracle's.

compressai's `RansEncoder.encode_with_indexes` / `RansDecoder.decode_with_indexes` semantics
(SURVEY.md Appendix B.4); call sites model/entropy_models.py:352-353,372,393,408."""
import numpy as np
import pytest

from oracle import rans as orans
from oracle.entropy import GaussianConditional


def _tables():
    gc = GaussianConditional()
    gc.update()
    return (np.ascontiguousarray(np.asarray(gc.cdf), dtype=np.int32), np.ascontiguousarray(gc.cdf_length, dtype=np.int32),
            np.ascontiguousarray(gc.offset, dtype=np.int32))


def _draw(rng, n, cdf_length, offset, spread):
    idx = rng.integers(0, cdf_length.size, size=n).astype(np.int32)
    half = (cdf_length[idx] - 2) // 2
    sym = np.rint(rng.standard_normal(n) * (half * spread + 0.3)).astype(np.int32)
    return sym, idx


@pytest.mark.parametrize("n,spread,seed", [(1, 0.3, 0), (257, 0.3, 1), (20000, 0.25, 2), (20000, 1.5, 3), (300000, 0.2, 4)])
def test_streams_are_byte_identical_and_round_trip(pcc, n, spread, seed):
    from pcc_amd import entropy as pe
    cdf, cdf_length, offset = _tables()
    rng = np.random.default_rng(seed)
    sym, idx = _draw(rng, n, cdf_length, offset, spread)       # spread 1.5 drives many symbols into the bypass escape
    ours = pe._rans_encode(sym, idx, cdf, cdf_length, offset)
    ref = orans.encode_with_indexes(sym, idx, cdf, cdf_length, offset)
    assert ours == ref
    back = pe._rans_decode(ours, idx, cdf, cdf_length, offset)
    assert np.array_equal(back, sym)
    assert np.array_equal(np.asarray(orans.decode_with_indexes(ours, idx, cdf, cdf_length, offset)), sym)


def test_every_table_every_symbol(pcc):
    """each (table, in-range value) pair once, plus both escape directions per table"""
    from pcc_amd import entropy as pe
    cdf, cdf_length, offset = _tables()
    sym, idx = [], []
    for t in range(cdf_length.size):
        maxv = cdf_length[t] - 2
        vals = np.arange(-3, maxv + 3) + offset[t]
        sym.append(vals)
        idx.append(np.full(vals.size, t))
    sym = np.concatenate(sym).astype(np.int32)
    idx = np.concatenate(idx).astype(np.int32)
    data = pe._rans_encode(sym, idx, cdf, cdf_length, offset)
    assert data == orans.encode_with_indexes(sym, idx, cdf, cdf_length, offset)
    assert np.array_equal(pe._rans_decode(data, idx, cdf, cdf_length, offset), sym)


def test_malformed_inputs_are_reported_not_crashed(pcc):
    from pcc_amd import entropy as pe
    cdf, cdf_length, offset = _tables()
    idx = np.zeros(4, dtype=np.int32)
    with pytest.raises(RuntimeError):
        pe._rans_decode(b"\x00\x01\x02", idx, cdf, cdf_length, offset)          # not a multiple of 4 bytes
    with pytest.raises(RuntimeError):
        pe._rans_decode(b"\x00" * 8, np.array([-1], dtype=np.int32), cdf, cdf_length, offset)


@pytest.mark.parametrize("n,spread,seed", [(1, 0.3, 0), (4099, 0.3, 1), (50000, 1.5, 2)])
def test_packed_planes_code_the_same_bytes(pcc, n, spread, seed):
    """the int16-symbol / uint8-index entry points (what the GPU writes for the y stream) against the int32 ones and
    the oracle: identical bytes, exact round trip; a symbol beyond int16 is reported, not truncated"""
    from pcc_amd import entropy as pe
    cdf, cdf_length, offset = _tables()
    rng = np.random.default_rng(seed)
    sym, idx = _draw(rng, n, cdf_length, offset, spread)
    data = pe._rans_encode_packed(sym.astype(np.int16), idx.astype(np.uint8), cdf, cdf_length, offset)
    assert data == pe._rans_encode(sym, idx, cdf, cdf_length, offset) == orans.encode_with_indexes(sym, idx, cdf, cdf_length, offset)
    out = np.empty(n, dtype=np.int16)
    assert pe._rans_decode_packed(data, idx.astype(np.uint8), cdf, cdf_length, offset, out)
    assert np.array_equal(out.astype(np.int32), sym)
    big = sym.copy()
    big[n // 2] = 40000                                                        # escape-coded, beyond int16
    data = pe._rans_encode(big, idx, cdf, cdf_length, offset)
    assert not pe._rans_decode_packed(data, idx.astype(np.uint8), cdf, cdf_length, offset, out)
    assert np.array_equal(pe._rans_decode(data, idx, cdf, cdf_length, offset), big)
