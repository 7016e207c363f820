"""Oracle: pure-Python twin of oracle/rans.c (slow; small cases only).

Written independently from the C file, directly from the algorithm description
in SURVEY.md Appendix B.4 (ryg_rans rans64 + compressai 1.2.4 rans_interface
semantics; third-party, not under /root/reference).  tests/ require the two to
agree byte for byte.  Test infrastructure only — see oracle/__init__.py.
"""
import struct

_L = 1 << 31
_MASK64 = (1 << 64) - 1


def pmf_to_quantized_cdf(pmf, precision=16):
    import numpy as np
    p = np.asarray(pmf, dtype=np.float32)
    if not np.all(np.isfinite(p)) or np.any(p < 0):
        raise ValueError("invalid pmf")
    # C's roundf: half away from zero (p >= 0 here)
    scaled = (p * np.float32(1 << precision)).astype(np.float32)
    cdf = [0] + [int(np.floor(np.float64(v) + 0.5)) for v in scaled]
    total = sum(cdf)
    if total == 0:
        raise ValueError("pmf sums to zero")
    cdf = [((1 << precision) * c) // total for c in cdf]
    for i in range(1, len(cdf)):
        cdf[i] += cdf[i - 1]
    cdf[-1] = 1 << precision
    n = len(cdf) - 1
    for i in range(n):
        if cdf[i] == cdf[i + 1]:
            best_freq, best = None, -1
            for j in range(n):
                f = cdf[j + 1] - cdf[j]
                if f > 1 and (best_freq is None or f < best_freq):
                    best_freq, best = f, j
            if best < 0:
                raise ValueError("cannot fix zero-width bin")
            if best < i:
                for j in range(best + 1, i + 1):
                    cdf[j] -= 1
            else:
                for j in range(i + 1, best + 1):
                    cdf[j] += 1
    return cdf


def encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets):
    queue = []
    for s, ix in zip(symbols, indexes):
        cdf = cdfs[ix]
        maxv = cdf_sizes[ix] - 2
        v = s - offsets[ix]
        raw = 0
        if v < 0:
            raw, v = -2 * v - 1, maxv
        elif v >= maxv:
            raw, v = 2 * (v - maxv), maxv
        queue.append((cdf[v] & 0xFFFF, (cdf[v + 1] - cdf[v]) & 0xFFFF, False))
        if v == maxv:
            n = 0
            while (raw >> (4 * n)) != 0:
                n += 1
            t = n
            while t >= 15:
                queue.append((15, 16, True))
                t -= 15
            queue.append((t, t + 1, True))
            for j in range(n):
                nib = (raw >> (4 * j)) & 15
                queue.append((nib, nib + 1, True))
    x = _L
    words = []  # emitted back to front
    for start, rng, byp in reversed(queue):
        if not byp:
            xmax = ((_L >> 16) << 32) * rng
            if x >= xmax:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x // rng) << 16) + (x % rng) + start
        else:
            xmax = ((_L >> 16) << 32) * (1 << 12)
            if x >= xmax:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x << 4) | start) & _MASK64
    words.append((x >> 32) & 0xFFFFFFFF)
    words.append(x & 0xFFFFFFFF)
    words.reverse()
    return struct.pack("<%dI" % len(words), *words)


def decode_with_indexes(data, indexes, cdfs, cdf_sizes, offsets):
    nw = len(data) // 4
    words = struct.unpack("<%dI" % nw, data[: nw * 4])
    pos = 2
    x = words[0] | (words[1] << 32)

    def nxt():
        nonlocal pos
        w = words[pos] if pos < nw else 0
        pos += 1
        return w

    def bits():
        nonlocal x
        val = x & 15
        x >>= 4
        if x < _L:
            x = (x << 32) | nxt()
        return val

    out = []
    for ix in indexes:
        cdf = cdfs[ix]
        size = cdf_sizes[ix]
        maxv = size - 2
        cf = x & 0xFFFF
        s = next(j for j in range(size) if cdf[j] > cf) - 1
        x = (cdf[s + 1] - cdf[s]) * (x >> 16) + cf - cdf[s]
        if x < _L:
            x = (x << 32) | nxt()
        value = s
        if value == maxv:
            val = bits()
            n = val
            while val == 15:
                val = bits()
                n += val
            raw = 0
            for j in range(n):
                raw |= bits() << (4 * j)
            value = raw >> 1
            value = -value - 1 if raw & 1 else value + maxv
        out.append(value + offsets[ix])
    return out
