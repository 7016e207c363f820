"""Oracle, order-matched mode: ctypes binding of oracle/chain.c (built by oracle/Makefile into oracle/_build/).

``conv_chain`` is the sparse convolution written as ONE fused multiply-add chain per output element, in the order the
product's kernels document for themselves (offsets ascending, absent neighbours skipped, channels ascending for the thin
kernels and 0, 4, 1, 5, 2, 6, 3, 7 within every 8 for the MFMA kernels) — see the header of chain.c.  oracle/nn.py
switches to it with ``nn.set_order("kernel")``; the default order stays the BLAS one (gather -> sgemm -> index_add_).

Test infrastructure only — see oracle/__init__.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpcc_oracle_chain.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "chain.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "chain"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        f32p, i64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        L.pcc_oracle_conv_chain.restype = ctypes.c_int
        L.pcc_oracle_conv_chain.argtypes = [f32p, ctypes.c_int64, ctypes.c_int32, f32p, i64p, ctypes.c_int64, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_int32, f32p, ctypes.c_int32]
        L.pcc_oracle_conv_chain_scalar.restype = ctypes.c_int
        L.pcc_oracle_conv_chain_scalar.argtypes = [f32p, ctypes.c_int32, f32p, i64p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                   ctypes.c_int32, f32p]
        L.pcc_oracle_gather_sum.restype = ctypes.c_int
        L.pcc_oracle_gather_sum.argtypes = [f32p, ctypes.c_int32, i64p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, f32p]
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i64(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def threads():
    try:
        return max(1, min(len(os.sched_getaffinity(0)), 32))
    except AttributeError:
        return max(1, min(os.cpu_count() or 1, 32))


def conv_chain(fin, w, nbr, n_out, mfma_order, scalar=False):
    """fin [n_in, cin], w [K, cin, cout], nbr int [n_out, K] (-1 = absent) or None (K = 1, identity) -> [n_out, cout], no bias"""
    fin, finp = _f32(fin)
    w, wp = _f32(w)
    assert w.ndim == 3 and fin.ndim == 2 and fin.shape[1] == w.shape[1], (fin.shape, w.shape)
    K, cin, cout = w.shape
    nbr, nbrp = _i64(nbr)
    if nbr is not None:
        assert nbr.shape == (n_out, K), (nbr.shape, n_out, K)
        assert nbr.size == 0 or (int(nbr.max()) < fin.shape[0] and int(nbr.min()) >= -1)
    else:
        assert K == 1 and n_out == fin.shape[0]
    out = np.empty((n_out, cout), dtype=np.float32)
    outp = out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    if scalar:
        rv = lib().pcc_oracle_conv_chain_scalar(finp, cin, wp, nbrp, n_out, K, cout, 1 if mfma_order else 0, outp)
    else:
        rv = lib().pcc_oracle_conv_chain(finp, fin.shape[0], cin, wp, nbrp, n_out, K, cout, 1 if mfma_order else 0, outp, threads())
    if rv != 0:
        raise ValueError("conv_chain: bad arguments")
    return out


def gather_sum(scores, nbr, cout):
    """scores [n_in, K * cout], nbr [n_out, K] -> [n_out, cout]: plain additions, k ascending, absent neighbours skipped"""
    scores, sp = _f32(scores)
    nbr, nbrp = _i64(nbr)
    n_out, K = nbr.shape
    assert scores.shape[1] == K * cout
    out = np.empty((n_out, cout), dtype=np.float32)
    rv = lib().pcc_oracle_gather_sum(sp, scores.shape[1], nbrp, n_out, K, cout, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rv == 0
    return out
