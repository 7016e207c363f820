"""Oracle: integer coordinate algebra of the sparse tensors (numpy, CPU).

Restates the MinkowskiEngine coordinate-manager rules the reference relies on
(ME is imported at /root/reference/model/model.py:3 and
is not vendored):

* coordinates are int32 ``[N, 4]`` rows ``(b, x, y, z)``, batch first
  (model/model.py:117-118, utils.py:438); non-integer inputs are floored.
* stride-2 convolution: output set = unique ``floor(c / 2ts) * 2ts``.
* generative transposed convolution, kernel K in {2, 3}, stride 2:
  output set = unique ``c + off * ts/2``; off in {-1,0,1}^3 (K=3) or {0,1}^3 (K=2).
* kernel index <-> offset: first spatial axis fastest,
  ``k = i_x + K i_y + K^2 i_z`` with ``i = off + 1`` (K=3) or ``i = off`` (K=2).
* canonical bitstream order = lexicographic (b, x, y, z) ascending
  (utils.py:155-204: int64 key with radix 1e5).

Test infrastructure only — see oracle/__init__.py.
"""
import numpy as np

_BITS = 18
_OFF = 1 << (_BITS - 1)  # coordinates must lie in (-2^17, 2^17 - 1); 10 bits of batch index


def pack(coords):
    """int64 key whose ascending order is lexicographic (b, x, y, z).

    Same ordering as the reference's radix-1e5 key (utils.py:170-171,199-200)
    for its coordinates (0 .. 99,999); here the radix is 2^18 with a +2^17 bias (the product's layout, csrc/common.h).
    """
    c = np.asarray(coords).astype(np.int64)
    assert c.ndim == 2 and c.shape[1] == 4
    if c.size:
        assert c[:, 0].min() >= 0 and c[:, 0].max() < 1023, "batch index out of range"
        assert c[:, 1:].min() > -_OFF and c[:, 1:].max() < _OFF - 1, "coordinate out of range"
    u = np.uint64
    return ((c[:, 0].astype(u) << u(3 * _BITS)) | ((c[:, 1] + _OFF).astype(u) << u(2 * _BITS))
            | ((c[:, 2] + _OFF).astype(u) << u(_BITS)) | (c[:, 3] + _OFF).astype(u))          # uint64: batch 512+ sets bit 63


def unpack(keys):
    k = np.asarray(keys, dtype=np.uint64)
    m = np.uint64((1 << _BITS) - 1)
    out = np.empty((k.shape[0], 4), dtype=np.int32)
    out[:, 0] = (k >> np.uint64(3 * _BITS)).astype(np.int64)
    out[:, 1] = ((k >> np.uint64(2 * _BITS)) & m).astype(np.int64) - _OFF
    out[:, 2] = ((k >> np.uint64(_BITS)) & m).astype(np.int64) - _OFF
    out[:, 3] = (k & m).astype(np.int64) - _OFF
    return out


def to_int_coords(coords):
    """ME floors non-int32 coordinates (model/model.py:184-188, utils.py:438)."""
    c = np.asarray(coords)
    if c.dtype.kind == "f":
        c = np.floor(c)
    return c.astype(np.int32)


def sort_order(coords):
    """Row permutation giving the canonical (b,x,y,z) order (utils.py:155-204)."""
    return np.argsort(pack(coords), kind="stable")


def kernel_offsets(ksize):
    """[K^3, 3] integer offsets in kernel-index order (x fastest)."""
    if ksize == 1:
        return np.zeros((1, 3), dtype=np.int64)
    rng = np.arange(ksize) - (1 if ksize == 3 else 0)
    offs = []
    for iz in rng:
        for iy in rng:
            for ix in rng:
                offs.append((ix, iy, iz))
    return np.asarray(offs, dtype=np.int64)


def stride_map(coords, ts):
    """Output coordinates of a stride-2 conv on a tensor of stride ``ts``.

    Unique ``floor(c / 2ts) * 2ts`` (batch column untouched), returned in
    canonical order.  ME's own row order is unspecified and the reference never
    relies on it.
    """
    c = np.asarray(coords).astype(np.int64)
    out = c.copy()
    out[:, 1:] = np.floor_divide(c[:, 1:], 2 * ts) * (2 * ts)
    keys = np.unique(pack(out))
    return unpack(keys)


def children(coords, ts, ksize):
    """Output coordinates of a generative transposed conv (kernel ksize, stride 2).

    blocks.py:84 (K=3), entropy_models.py:286,290 (K=2), :298,302 (K=3 via the
    non-generative ConvTranspose on a fresh coordinate manager, decode order).
    """
    assert ts % 2 == 0
    half = ts // 2
    c = np.asarray(coords).astype(np.int64)
    offs = kernel_offsets(ksize) * half
    cand = np.repeat(c[:, None, :], offs.shape[0], axis=1)
    cand[:, :, 1:] += offs[None, :, :]
    keys = np.unique(pack(cand.reshape(-1, 4)))
    return unpack(keys)


def lookup(table_coords, query_coords):
    """Row index of each query coordinate in ``table_coords`` or -1."""
    tk = pack(table_coords)
    order = np.argsort(tk, kind="stable")
    tks = tk[order]
    qk = pack(query_coords)
    pos = np.searchsorted(tks, qk)
    pos_c = np.minimum(pos, max(len(tks) - 1, 0))
    hit = (len(tks) > 0) & (tks[pos_c] == qk) if len(tks) else np.zeros(len(qk), bool)
    idx = np.where(hit, order[pos_c] if len(tks) else 0, -1)
    return idx.astype(np.int64)


def kernel_map(in_coords, out_coords, ksize, step, transposed=False):
    """Neighbour table ``nbr[N_out, K^3]`` (input row or -1).

    forward conv (stride 1 or 2):  c_in = c_out + off_k * step   (step = ts_in)
    transposed / generative:        c_in = c_out - off_k * step   (step = ts_in/2)
    so that ``out[c_in + off_k*step] += in[c_in] @ W[k]``.
    """
    offs = kernel_offsets(ksize) * step
    if transposed:
        offs = -offs
    oc = np.asarray(out_coords).astype(np.int64)
    K = offs.shape[0]
    nbr = np.empty((oc.shape[0], K), dtype=np.int64)
    tk = pack(in_coords)
    order = np.argsort(tk, kind="stable")
    tks = tk[order]
    for k in range(K):
        q = oc.copy()
        q[:, 1:] += offs[k][None, :]
        # queries may leave the packable range only by |off| <= step: guarded by pack()
        qk = pack(q)
        pos = np.searchsorted(tks, qk)
        pos_c = np.minimum(pos, max(len(tks) - 1, 0))
        if len(tks):
            hit = tks[pos_c] == qk
            nbr[:, k] = np.where(hit, order[pos_c], -1)
        else:
            nbr[:, k] = -1
    return nbr


def count_per_batch(coords):
    """transforms.py:65-71 — number of rows per batch index, ascending batch id."""
    b = np.asarray(coords)[:, 0]
    return [int((b == i).sum()) for i in np.unique(b)]
