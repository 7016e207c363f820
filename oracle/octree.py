"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of this build's latent-coordinate coder.

What it stands in for: the reference writes the stride-8 coordinate list with the external MPEG
G-PCC binary `tmc3` (lossless octree geometry; /root/reference/model/model.py:318-395,
`gpcc_encode` / `gpcc_decode`).  That binary is neither in the reference tree nor in this image, and
its bitstream cannot be reproduced here, so the product ships its own lossless octree coder
("PCO1", learned-compression-..._amd/octree.py + csrc/octree.hip).  This file is the CPU twin the
GPU path is checked against; there is no reference output to pin it to (parity unpinned for the
coordinate payload: only losslessness and GPU == CPU byte equality are claimed, never G-PCC
compatibility).

Format PCO1 (all little-endian):
    magic "PCO1" | u8 depth D | u8 0 | u16 0 | i32 stride | i32 origin[3] | u32 n_points |
    u32 level_counts[D] | u8 table_flag[D] | per flagged level 256 LEB128 frequencies |
    u32 payload_len | payload
Octree: grid g = (c - origin) / stride, D = bits of max(g); child index at every level
= (xbit << 2) | (ybit << 1) | zbit, x most significant; level L (0 = root) holds one occupancy byte
per occupied node of depth L in ascending Morton order.  Payload: one rANS stream (the latent coder,
SURVEY.md Appendix B.4) over symbols byte - 1 in level-major order, table index = level; a level
with fewer than 64 nodes uses the uniform table (257/65536 per symbol).
"""
import struct

import numpy as np

from . import rans as orans

MAGIC = b"PCO1"
MIN_TABLE_NODES = 64


def _spread3(v, depth):
    """bit b of v -> bit 3 b"""
    out = np.zeros(v.shape, dtype=np.uint64)
    for b in range(depth):
        out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
    return out


def morton_keys(g, depth):
    g = g.astype(np.uint64)
    return (_spread3(g[:, 0], depth) << np.uint64(2)) | (_spread3(g[:, 1], depth) << np.uint64(1)) | _spread3(g[:, 2], depth)


def keys_to_grid(keys, depth):
    g = np.zeros((keys.shape[0], 3), dtype=np.int64)
    for b in range(depth):
        g[:, 0] |= (((keys >> np.uint64(3 * b + 2)) & np.uint64(1)) << np.uint64(b)).astype(np.int64)
        g[:, 1] |= (((keys >> np.uint64(3 * b + 1)) & np.uint64(1)) << np.uint64(b)).astype(np.int64)
        g[:, 2] |= (((keys >> np.uint64(3 * b)) & np.uint64(1)) << np.uint64(b)).astype(np.int64)
    return g


def grid_of(points, stride):
    """points int [N,3] -> (origin[3], grid [N,3], depth)"""
    p = np.asarray(points, dtype=np.int64)
    if p.shape[0] == 0:
        return np.zeros(3, dtype=np.int64), p.reshape(0, 3), 0
    origin = p.min(axis=0)
    rel = p - origin
    if (rel % stride).any():
        raise ValueError("coordinates are not multiples of the stride")
    g = rel // stride
    depth = int(g.max()).bit_length()
    return origin, g, depth


def occupancy_levels(keys_sorted, depth):
    """sorted unique leaf keys -> list of uint8 arrays, one per level 0..depth-1"""
    levels = []
    for L in range(depth):
        parent = keys_sorted >> np.uint64(3 * (depth - L))
        child = ((keys_sorted >> np.uint64(3 * (depth - L - 1))) & np.uint64(7)).astype(np.int64)
        uniq, inv = np.unique(parent, return_inverse=True)
        byte = np.zeros(uniq.shape[0], dtype=np.int64)
        np.bitwise_or.at(byte, inv, 1 << child)
        levels.append(byte.astype(np.uint8))
    return levels


def expand_levels(levels, depth):
    """inverse of occupancy_levels -> sorted leaf keys"""
    nodes = np.zeros(1, dtype=np.uint64)
    for L in range(depth):
        b = levels[L].astype(np.int64)
        assert b.shape[0] == nodes.shape[0] and (b > 0).all()
        out = []
        for c in range(8):
            sel = (b >> c) & 1
            out.append(np.stack([np.nonzero(sel)[0], np.full(int(sel.sum()), c)], axis=1))
        pairs = np.concatenate(out, axis=0)
        order = np.lexsort((pairs[:, 1], pairs[:, 0]))
        pairs = pairs[order]
        nodes = (nodes[pairs[:, 0]] << np.uint64(3)) | pairs[:, 1].astype(np.uint64)
    return nodes


def _leb128(values):
    out = bytearray()
    for v in values:
        v = int(v)
        while True:
            byte = v & 0x7F
            v >>= 7
            out.append(byte | (0x80 if v else 0))
            if not v:
                break
    return bytes(out)


def _read_leb128(data, pos, count):
    vals = []
    for _ in range(count):
        v, shift = 0, 0
        while True:
            byte = data[pos]
            pos += 1
            v |= (byte & 0x7F) << shift
            shift += 7
            if not byte & 0x80:
                break
        vals.append(v)
    return vals, pos


def uniform_cdf():
    freq = np.full(256, 257, dtype=np.int64)
    freq[255] = 1                                    # the tail / escape bin, never used
    return np.concatenate([[0], np.cumsum(freq)]).astype(np.int32)


def level_cdf(level_bytes):
    """257-entry quantised CDF (255 symbols + tail) of one level's occupancy bytes"""
    q = orans.pmf_to_quantized_cdf
    counts = np.bincount(level_bytes.astype(np.int64) - 1, minlength=255).astype(np.float32)
    pmf = np.concatenate([counts / np.float32(counts.sum()), np.zeros(1, dtype=np.float32)]).astype(np.float32)
    return np.asarray(q(pmf, 16), dtype=np.int32)


def pack_stream(origin, stride, n_points, depth, levels):
    """levels (list of uint8 arrays) -> PCO1 bytes"""
    enc = orans.encode_with_indexes
    flags, tables, cdfs = [], b"", []
    for lv in levels:
        if lv.shape[0] >= MIN_TABLE_NODES:
            cdf = level_cdf(lv)
            flags.append(1)
            tables += _leb128(np.diff(cdf.astype(np.int64)))
        else:
            cdf = uniform_cdf()
            flags.append(0)
        cdfs.append(cdf)
    head = MAGIC + struct.pack("<BBHi3iI", depth, 0, 0, int(stride), *[int(v) for v in origin], int(n_points))
    head += struct.pack("<%dI" % depth, *[int(lv.shape[0]) for lv in levels])
    head += bytes(flags) + tables
    if depth == 0:
        return head + struct.pack("<I", 0)
    symbols = np.concatenate([lv.astype(np.int32) - 1 for lv in levels])
    indexes = np.concatenate([np.full(lv.shape[0], i, dtype=np.int32) for i, lv in enumerate(levels)])
    cdf_mat = np.stack(cdfs).astype(np.int32)
    sizes = np.full(depth, 257, dtype=np.int32)
    offsets = np.zeros(depth, dtype=np.int32)
    payload = enc(symbols, indexes, cdf_mat, sizes, offsets)
    return head + struct.pack("<I", len(payload)) + payload


def unpack_stream(data):
    dec = orans.decode_with_indexes
    if data[:4] != MAGIC:
        raise ValueError("not a PCO1 coordinate stream")
    depth, _, _, stride, ox, oy, oz, n_points = struct.unpack("<BBHi3iI", data[4:28])
    pos = 28
    counts = list(struct.unpack("<%dI" % depth, data[pos:pos + 4 * depth]))
    pos += 4 * depth
    flags = list(data[pos:pos + depth])
    pos += depth
    cdfs = []
    for f in flags:
        if f:
            freqs, pos = _read_leb128(data, pos, 256)
            cdfs.append(np.concatenate([[0], np.cumsum(freqs)]).astype(np.int32))
        else:
            cdfs.append(uniform_cdf())
    (plen,) = struct.unpack("<I", data[pos:pos + 4])
    pos += 4
    payload = data[pos:pos + plen]
    levels = []
    if depth:
        indexes = np.concatenate([np.full(c, i, dtype=np.int32) for i, c in enumerate(counts)])
        sym = np.asarray(dec(payload, indexes, np.stack(cdfs).astype(np.int32), np.full(depth, 257, dtype=np.int32),
                             np.zeros(depth, dtype=np.int32)), dtype=np.int64)
        o = 0
        for c in counts:
            levels.append((sym[o:o + c] + 1).astype(np.uint8))
            o += c
    return np.array([ox, oy, oz], dtype=np.int64), stride, n_points, depth, levels


def encode_points(points, stride):
    """int [N,3] (multiples of `stride` apart, unique) -> bytes"""
    origin, g, depth = grid_of(points, stride)
    keys = np.sort(morton_keys(g, depth)) if g.shape[0] else np.zeros(0, dtype=np.uint64)
    if keys.shape[0] > 1 and (np.diff(keys.astype(np.int64)) == 0).any():
        raise ValueError("duplicate coordinates")
    return pack_stream(origin, stride, g.shape[0], depth, occupancy_levels(keys, depth))


def decode_points(data):
    """bytes -> int64 [N,3] in ascending Morton order"""
    origin, stride, n_points, depth, levels = unpack_stream(data)
    if n_points == 0:
        return np.zeros((0, 3), dtype=np.int64)
    keys = expand_levels(levels, depth)
    assert keys.shape[0] == n_points
    return keys_to_grid(keys, depth) * stride + origin
