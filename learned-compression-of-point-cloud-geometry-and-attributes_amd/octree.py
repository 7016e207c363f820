"""Lossless coder for the stride-8 latent coordinate list of file mode ("PCO1").

Takes the place of ``ColorModel.gpcc_encode`` / ``gpcc_decode`` (/root/reference/model/model.py:318-395),
which write a PLY and shell out to the external MPEG G-PCC binary ``tmc3``.  ``tmc3`` is not part
of the reference tree and not available here, and its bitstream is not reproduced: PCO1 is this
build's own format, lossless like the G-PCC octree mode the reference configures
(``--mode=0 --positionQuantizationScale=1 ...``), but NOT G-PCC compatible.

Device (csrc/octree.hip): Morton keys, radix sort, one occupancy byte per occupied octree node and
level; and the inverse expansion.  Host: the per-level frequency tables and the range coder shared
with the latents (``pcc_rans_*_with_indexes``), table index = octree level.

Stream (little-endian):
    "PCO1" | u8 depth | u8 0 | u16 0 | i32 stride | i32 origin[3] | u32 n_points |
    u32 nodes_per_level[depth] | u8 has_table[depth] | 256 LEB128 frequencies per level with a table |
    u32 payload_len | payload
A level with fewer than 64 nodes is coded with the flat table (257 / 65536 per byte value).
"""
import struct

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr
from .entropy import _rans_decode, _rans_encode, _to_host

MAGIC = b"PCO1"
_MIN_TABLE_NODES = 64
_PRECISION = 16


def _flat_cdf():
    f = np.full(256, 257, dtype=np.int64)
    f[255] = 1                          # tail bin (the coder's escape symbol; never emitted)
    out = np.zeros(257, dtype=np.int32)
    out[1:] = np.cumsum(f)
    return out


def _quantized_cdf(level_bytes):
    """occupancy bytes of one level -> 257-entry CDF (255 byte values + tail), 16-bit precision"""
    L = _lib.lib()
    counts = np.bincount(level_bytes.astype(np.int64) - 1, minlength=255).astype(np.float32)
    pmf = np.zeros(256, dtype=np.float32)
    pmf[:255] = counts / np.float32(counts.sum())
    cdf = np.zeros(257, dtype=np.int32)
    check(L.pcc_pmf_to_quantized_cdf(ptr(pmf), 256, _PRECISION, ptr(cdf)))
    return cdf


def _write_varints(values):
    out = bytearray()
    for v in values:
        v = int(v)
        while v >= 0x80:
            out.append((v & 0x7F) | 0x80)
            v >>= 7
        out.append(v)
    return bytes(out)


def _read_varints(buf, pos, count):
    out = np.empty(count, dtype=np.int64)
    for i in range(count):
        v = shift = 0
        while True:
            if pos >= len(buf):
                raise ValueError("PCO1: truncated frequency table")
            b = buf[pos]
            pos += 1
            v |= (b & 0x7F) << shift
            shift += 7
            if b < 0x80:
                break
        out[i] = v
    return out, pos


def encode_coordinates(coords, stride):
    """coords: int32 [N,4] on the GPU (batch column ignored; one cloud) -> bytes."""
    L = _lib.lib()
    if coords.device.type != "cuda":
        raise RuntimeError("encode_coordinates: coordinates must live on the GPU")
    coords = coords.to(torch.int32).contiguous()
    n = int(coords.shape[0])
    stride = int(stride)
    if n == 0:
        return MAGIC + struct.pack("<BBHi3iI", 0, 0, 0, stride, 0, 0, 0, 0) + struct.pack("<I", 0)
    lo = coords[:, 1:4].amin(dim=0)
    hi = coords[:, 1:4].amax(dim=0)
    lohi = torch.cat([lo, hi]).cpu().numpy().astype(np.int64)
    origin = np.ascontiguousarray(lohi[:3], dtype=np.int32)
    depth = int(((lohi[3:] - lohi[:3]).max() // stride)).bit_length()
    dev = coords.device
    occ = torch.empty(max(depth, 1) * n, dtype=torch.uint8, device=dev)
    counts = torch.empty(depth + 2, dtype=torch.int32, device=dev)
    nbytes = L.pcc_octree_scratch_bytes(n)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(L.pcc_octree_occupancy(ptr(coords), n, stride, ptr(origin), depth, ptr(occ), ptr(counts), ptr(scratch), nbytes,
                                 _lib.stream()))
    cnt = counts.cpu().numpy().astype(np.int64)
    if cnt[depth + 1] != 0:
        raise ValueError("encode_coordinates: %d coordinates are not on the stride-%d lattice" % (cnt[depth + 1], stride))
    if depth and cnt[depth] != n:
        raise ValueError("encode_coordinates: duplicate coordinates (%d distinct of %d)" % (cnt[depth], n))
    if depth == 0 and n != 1:
        raise ValueError("encode_coordinates: duplicate coordinates")
    if depth:
        picked = torch.cat([occ[l * n:l * n + int(cnt[l])] for l in range(depth)])
        host = _to_host(picked, "octree_occ").copy()
    levels, o = [], 0
    for l in range(depth):
        levels.append(host[o:o + int(cnt[l])])
        o += int(cnt[l])
    head = MAGIC + struct.pack("<BBHi3iI", depth, 0, 0, stride, int(origin[0]), int(origin[1]), int(origin[2]), n)
    head += struct.pack("<%dI" % depth, *[int(c) for c in cnt[:depth]])
    flags, tables = bytearray(), b""
    cdf = np.zeros((max(depth, 1), 257), dtype=np.int32)
    for l, lv in enumerate(levels):
        if lv.shape[0] >= _MIN_TABLE_NODES:
            cdf[l] = _quantized_cdf(lv)
            flags.append(1)
            tables += _write_varints(np.diff(cdf[l].astype(np.int64)))
        else:
            cdf[l] = _flat_cdf()
            flags.append(0)
    head += bytes(flags) + tables
    if depth == 0:
        return head + struct.pack("<I", 0)
    symbols = host.astype(np.int32) - 1
    indexes = np.repeat(np.arange(depth, dtype=np.int32), cnt[:depth])
    payload = _rans_encode(symbols, indexes, cdf, np.full(depth, 257, dtype=np.int32), np.zeros(depth, dtype=np.int32))
    return head + struct.pack("<I", len(payload)) + payload


def decode_coordinates(data, device, batch=0):
    """bytes -> int32 [N,4] = (batch, x, y, z) on ``device``, ascending Morton order."""
    L = _lib.lib()
    data = bytes(data)
    if len(data) < 28 or data[:4] != MAGIC:
        raise ValueError("not a PCO1 coordinate stream")
    depth, _, _, stride, ox, oy, oz, n = struct.unpack("<BBHi3iI", data[4:28])
    if depth > 21 or stride < 1:
        raise ValueError("PCO1: bad header")
    pos = 28
    if len(data) < pos + 5 * depth + 4:
        raise ValueError("PCO1: truncated header")
    cnt = np.array(struct.unpack("<%dI" % depth, data[pos:pos + 4 * depth]), dtype=np.int64)
    pos += 4 * depth
    flags = data[pos:pos + depth]
    pos += depth
    cdf = np.zeros((max(depth, 1), 257), dtype=np.int32)
    for l in range(depth):
        if flags[l]:
            freqs, pos = _read_varints(data, pos, 256)
            if freqs.min() < 1 or freqs.sum() != 1 << _PRECISION:
                raise ValueError("PCO1: invalid frequency table at level %d" % l)
            cdf[l, 1:] = np.cumsum(freqs)
        else:
            cdf[l] = _flat_cdf()
    (plen,) = struct.unpack("<I", data[pos:pos + 4])
    pos += 4
    payload = data[pos:pos + plen]
    if len(payload) != plen:
        raise ValueError("PCO1: truncated payload")
    if n == 0:
        return torch.zeros((0, 4), dtype=torch.int32, device=device)
    out = torch.empty((n, 4), dtype=torch.int32, device=device)
    origin = np.array([ox, oy, oz], dtype=np.int32)
    occ_dev = torch.zeros(1, dtype=torch.uint8, device=device)
    if depth:
        indexes = np.repeat(np.arange(depth, dtype=np.int32), cnt)
        sym = _rans_decode(payload, indexes, cdf, np.full(depth, 257, dtype=np.int32), np.zeros(depth, dtype=np.int32))
        if sym.min() < 0 or sym.max() > 254:
            raise ValueError("PCO1: corrupt payload")
        occ = (sym + 1).astype(np.uint8)
        # every level's popcounts must add up to the size of the next (the last to n_points)
        pop = np.unpackbits(occ[:, None], axis=1).sum(axis=1)
        ends = np.cumsum(cnt)
        want = np.append(cnt[1:], n)
        got = np.add.reduceat(pop, np.concatenate([[0], ends[:-1]]))
        if cnt[0] != 1 or (got != want).any():
            raise ValueError("PCO1: occupancy does not match the level sizes")
        occ_dev = torch.from_numpy(occ).to(device)
    nbytes = L.pcc_octree_scratch_bytes(n)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=device)
    counts64 = np.ascontiguousarray(cnt if depth else np.zeros(1), dtype=np.int64)
    check(L.pcc_octree_expand(ptr(occ_dev), ptr(counts64), depth, stride, ptr(origin), int(batch), n, ptr(out), ptr(scratch),
                              nbytes, _lib.stream()))
    return out
