"""Oracle: ctypes binding of oracle/rans.c (built by oracle/Makefile into oracle/_build/).

Test infrastructure only — see oracle/__init__.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpcc_oracle_rans.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "rans.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "rans"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.pcc_oracle_rans_encode.restype = ctypes.c_long
        L.pcc_oracle_rans_encode.argtypes = [i32p, i32p, ctypes.c_long, i32p, ctypes.c_int, i32p, i32p, u8p, ctypes.c_long]
        L.pcc_oracle_rans_decode.restype = ctypes.c_int
        L.pcc_oracle_rans_decode.argtypes = [u8p, ctypes.c_long, i32p, ctypes.c_long, i32p, ctypes.c_int, i32p, i32p, i32p]
        L.pcc_oracle_pmf_to_quantized_cdf.restype = ctypes.c_int
        L.pcc_oracle_pmf_to_quantized_cdf.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
        _lib = L
    return _lib


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets):
    """Same argument order as compressai's ``encode_with_indexes`` (B.4)."""
    s, sp = _i32(symbols)
    ix, ixp = _i32(indexes)
    c, cp = _i32(cdfs)
    assert c.ndim == 2
    cs, csp = _i32(cdf_sizes)
    of, ofp = _i32(offsets)
    n = s.size
    assert ix.size == n
    cap = 4 * (3 * n + 16) + 64
    while True:
        out = np.empty(cap, dtype=np.uint8)
        rv = lib().pcc_oracle_rans_encode(sp, ixp, n, cp, c.shape[1], csp, ofp,
                                          out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), cap)
        if rv == -2:
            cap *= 4
            continue
        if rv < 0:
            raise MemoryError("rans encode failed")
        return out[:rv].tobytes()


def decode_with_indexes(data, indexes, cdfs, cdf_sizes, offsets):
    ix, ixp = _i32(indexes)
    c, cp = _i32(cdfs)
    cs, csp = _i32(cdf_sizes)
    of, ofp = _i32(offsets)
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(ix.size, dtype=np.int32)
    rv = lib().pcc_oracle_rans_decode(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), len(data), ixp, ix.size,
                                      cp, c.shape[1], csp, ofp, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    if rv != 0:
        raise ValueError("malformed rANS stream")
    return out


def pmf_to_quantized_cdf(pmf, precision=16):
    p = np.ascontiguousarray(pmf, dtype=np.float32)
    cdf = np.empty(p.size + 1, dtype=np.uint32)
    rv = lib().pcc_oracle_pmf_to_quantized_cdf(p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), p.size, precision,
                                               cdf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    if rv != 0:
        raise ValueError("invalid pmf")
    return cdf.astype(np.int64)
