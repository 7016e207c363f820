// Host-side entropy coder of libpcc_hip.so: range-ANS with table indexes, bit-compatible with
// the coder the reference reaches through compressai 1.2.4 (`ans.RansEncoder.encode_with_indexes`
// / `RansDecoder.decode_with_indexes`, called from model/entropy_models.py:352-353,372,393,408)
// and the quantised-CDF builder behind `CompressionModel.update()` (model/model.py:30-36).
//
// rANS is one serial state machine per stream and the reference's bitstream has exactly one
// stream per tensor with symbols in channel-major order, so it stays on the
// host by design: the GPU produces int32 symbol / index planes, this file turns them into bytes.
// This is the shipped coder, not a fallback: there is no GPU variant to fall back from.
//
// Speed (2.6 M symbols per 10-bit frame sit on the critical path of every encode and decode):
//  * encoder: single reverse sweep (no intermediate symbol queue); x / freq and x % freq are
//    replaced by a multiply-high with a per-(table, symbol) reciprocal — the exact-division
//    construction of ryg_rans' Rans64EncSymbol, so the bytes are unchanged;
//  * decoder: a 256-bucket start table per CDF row turns the symbol search into a short forward
//    scan (the published decoder scans the row linearly from its beginning).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/pcc_hip.h"

namespace pcc { void set_error(const char* fmt, ...); }

namespace {

constexpr int kPrecision = 16;
constexpr int kBypassBits = 4;
constexpr uint32_t kBypassMax = 15;
constexpr uint64_t kRansL = 1ull << 31;

struct EncSym {
    uint64_t rcp_freq;   // fixed-point reciprocal of freq
    uint64_t x_max;      // renormalisation threshold
    uint32_t bias;
    uint16_t cmpl_freq;  // 2^16 - freq
    uint16_t rcp_shift;
};

inline uint64_t mul_hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

inline void enc_sym_init(EncSym& s, uint32_t start, uint32_t freq) {
    s.x_max = ((kRansL >> kPrecision) << 32) * freq;
    s.cmpl_freq = (uint16_t)((1u << kPrecision) - freq);
    if (freq < 2) {
        // freq == 1: q = mul_hi(x, ~0) = x - 1 (x > 0), so x + bias + q * (2^16 - 1) = x * 2^16 + start
        s.rcp_freq = ~0ull;
        s.rcp_shift = 0;
        s.bias = start + (1u << kPrecision) - 1;
    } else {
        uint32_t shift = 0;
        while (freq > (1u << shift)) ++shift;
        // rcp_freq = ceil(2^(shift + 63) / freq), via two 64-bit divisions
        uint64_t x0 = freq - 1;
        const uint64_t x1 = 1ull << (shift + 31);
        const uint64_t t1 = x1 / freq;
        x0 += (x1 % freq) << 32;
        const uint64_t t0 = x0 / freq;
        s.rcp_freq = t0 + (t1 << 32);
        s.rcp_shift = (uint16_t)(shift - 1);
        s.bias = start;
    }
}

struct Writer {
    uint32_t* ptr;   // grows downwards
    uint32_t* base;  // lowest writable word (one spare word below it is always writable)
    bool overflow = false;
    inline void emit(uint32_t v) {
        if (ptr > base) *--ptr = v;
        else overflow = true;
    }
};

inline void put_sym(uint64_t& x, Writer& w, const EncSym& s) {
    // branch-free renormalisation (taken for about every other symbol -> unpredictable as a branch):
    // the word is always stored below ptr, and kept only if needed
    const bool need = x >= s.x_max;
    if (w.ptr <= w.base) { if (need) w.overflow = true; }
    else { w.ptr[-1] = (uint32_t)x; w.ptr -= need ? 1 : 0; }
    x = need ? (x >> 32) : x;
    const uint64_t q = mul_hi(x, s.rcp_freq) >> s.rcp_shift;
    x = x + s.bias + q * s.cmpl_freq;
}

inline void put_bits(uint64_t& x, Writer& w, uint32_t val) {
    const uint64_t x_max = ((kRansL >> 16) << 32) * (1ull << (16 - kBypassBits));
    if (x >= x_max) { w.emit((uint32_t)x); x >>= 32; }
    x = (x << kBypassBits) | val;
}

}  // namespace

// the coder proper, on the plane types the caller holds: int32 / int32 (compressai's argument types) or the packed
// int16 symbols / uint8 indexes the GPU writes for the y stream (3 bytes per symbol over PCIe and through the cache)
template <class SymT, class IdxT>
static int64_t rans_encode(const SymT* symbols, const IdxT* indexes, int64_t n, const int32_t* cdfs,
                           int32_t cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets,
                           uint8_t* out, int64_t out_cap) {
    if (n < 0 || !out) { pcc::set_error("pcc_rans_encode_with_indexes: bad arguments"); return PCC_ERR_ARG; }
    // which tables does the stream touch, and what is the highest one?
    int32_t max_ix = -1;
    for (int64_t i = 0; i < n; ++i) {
        if (indexes[i] < 0) { pcc::set_error("pcc_rans_encode_with_indexes: negative table index"); return PCC_ERR_DATA; }
        if (indexes[i] > max_ix) max_ix = indexes[i];
    }
    // per-(table, symbol) encoder entries, rows packed back to back
    std::vector<int64_t> row_off((size_t)max_ix + 2, 0);
    for (int32_t t = 0; t <= max_ix; ++t) row_off[(size_t)t + 1] = row_off[(size_t)t] + (cdf_sizes[t] - 1);
    std::vector<EncSym> table((size_t)row_off[(size_t)max_ix + 1]);
    for (int32_t t = 0; t <= max_ix; ++t) {
        const int32_t* cdf = cdfs + (int64_t)t * cdf_stride;
        EncSym* row = table.data() + row_off[(size_t)t];
        for (int32_t v = 0; v < cdf_sizes[t] - 1; ++v) {
            const uint32_t start = (uint32_t)cdf[v] & 0xFFFFu;
            const uint32_t freq = (uint32_t)(cdf[v + 1] - cdf[v]) & 0xFFFFu;
            if (freq == 0) { row[v].x_max = 0; row[v].rcp_freq = 0; continue; }   // unusable symbol, checked below
            enc_sym_init(row[v], start, freq);
        }
    }
    // words are written downwards from the end of the caller's buffer (no scratch allocation) and
    // moved to its front at the end; the worst case is < 2 words per symbol (16 + 4*9 bits)
    uint8_t* aligned = out + ((4 - (reinterpret_cast<uintptr_t>(out) & 3)) & 3);
    const int64_t cap_words = (out_cap - (aligned - out)) / 4;
    if (cap_words < 2) { pcc::set_error("pcc_rans_encode_with_indexes: output buffer too small"); return PCC_ERR_ARG; }
    uint32_t* wbuf = reinterpret_cast<uint32_t*>(aligned);
    Writer w{wbuf + cap_words, wbuf};
    uint64_t x = kRansL;
    for (int64_t i = n - 1; i >= 0; --i) {
        const int32_t ix = indexes[i];
        const int32_t maxv = cdf_sizes[ix] - 2;
        int32_t v = symbols[i] - offsets[ix];
        if ((uint32_t)v >= (uint32_t)maxv) {      // v < 0 or v >= maxv: escape
            uint32_t raw;
            if (v < 0) raw = (uint32_t)(-2 * v - 1);
            else raw = (uint32_t)(2 * (v - maxv));
            v = maxv;
            int nb = 0;
            while ((raw >> (nb * kBypassBits)) != 0) ++nb;
            // forward order: main, count chunks (15,15,...,rest), nibbles LSB first  => reverse here
            for (int j = nb - 1; j >= 0; --j) put_bits(x, w, (raw >> (j * kBypassBits)) & kBypassMax);
            const int full = nb / (int)kBypassMax, rest = nb % (int)kBypassMax;
            put_bits(x, w, (uint32_t)rest);
            for (int j = 0; j < full; ++j) put_bits(x, w, kBypassMax);
        }
        const EncSym& s = table[(size_t)(row_off[(size_t)ix] + v)];
        if (s.x_max == 0) { pcc::set_error("pcc_rans_encode_with_indexes: zero-frequency symbol at %lld", (long long)i); return PCC_ERR_DATA; }
        put_sym(x, w, s);
    }
    w.emit((uint32_t)(x >> 32));
    w.emit((uint32_t)x);
    if (w.overflow) { pcc::set_error("pcc_rans_encode_with_indexes: output buffer too small (%lld bytes)", (long long)out_cap); return PCC_ERR_ARG; }
    const int64_t nbytes = (int64_t)((wbuf + cap_words) - w.ptr) * 4;
    std::memmove(out, w.ptr, (size_t)nbytes);
    return nbytes;
}

// *narrowed (may be NULL for int32 output) is set when a decoded symbol does not fit SymT: the caller repeats the
// decode with int32 output
template <class SymT, class IdxT>
static int rans_decode(const uint8_t* data, int64_t nbytes, const IdxT* indexes, int64_t n,
                       const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                       const int32_t* offsets, SymT* out_symbols, int32_t* narrowed) {
    if (nbytes < 8 || (nbytes & 3)) { pcc::set_error("pcc_rans_decode_with_indexes: malformed stream length %lld", (long long)nbytes); return PCC_ERR_DATA; }
    int32_t max_ix = -1;
    for (int64_t i = 0; i < n; ++i) {
        if (indexes[i] < 0) { pcc::set_error("pcc_rans_decode_with_indexes: negative table index"); return PCC_ERR_DATA; }
        if (indexes[i] > max_ix) max_ix = indexes[i];
    }
    // Per-table decode state packed back to back (the caller's [n_tables, stride] int32 matrix is
    // ~800 KB, its used prefix a few tens of KB):
    //   lut[b]  = start | freq << 16 | symbol << 32 of the symbol whose interval contains slot b << 8,
    //             bit 47 set when the whole bucket [b << 8, (b + 1) << 8) lies inside that symbol: the
    //             common case needs ONE dependent load between two states of the decoder (the
    //             state -> slot -> table -> state chain is what bounds a serial rANS decode);
    //   sf[s]   = start | freq << 16   (one load instead of cdf[s], cdf[s + 1]) for impure buckets;
    //   cdf[]   = the row itself, for the short forward scan of impure buckets.
    constexpr int kBuckets = 256, kShift = kPrecision - 8;
    struct DecTable { const uint32_t* cdf; const uint32_t* sf; const uint64_t* lut; int32_t maxv; int32_t offset; };
    std::vector<uint64_t> lut((size_t)(max_ix + 1) * kBuckets);
    std::vector<int64_t> row_off((size_t)max_ix + 2, 0);
    for (int32_t t = 0; t <= max_ix; ++t) row_off[(size_t)t + 1] = row_off[(size_t)t] + cdf_sizes[t];
    std::vector<uint32_t> packed((size_t)row_off[(size_t)max_ix + 1] + 1, 0xFFFFFFFFu);   // + sentinel
    std::vector<uint32_t> sfv((size_t)row_off[(size_t)max_ix + 1] + 1, 0u);
    std::vector<DecTable> tabs((size_t)max_ix + 1);
    for (int32_t t = 0; t <= max_ix; ++t) {
        const int32_t* cdf = cdfs + (int64_t)t * cdf_stride;
        uint32_t* row = packed.data() + row_off[(size_t)t];
        uint32_t* sf = sfv.data() + row_off[(size_t)t];
        if (cdf_sizes[t] > 32768) { pcc::set_error("pcc_rans_decode_with_indexes: table %d too long", t); return PCC_ERR_ARG; }
        for (int32_t j = 0; j < cdf_sizes[t]; ++j) row[j] = (uint32_t)cdf[j];
        const int32_t nsym = cdf_sizes[t] - 1;
        for (int32_t j = 0; j < nsym; ++j) sf[j] = (row[j] & 0xFFFFu) | ((row[j + 1] - row[j]) << 16);
        int32_t sidx = 0;
        for (int b = 0; b < kBuckets; ++b) {
            const uint32_t slot = (uint32_t)b << kShift;
            while (sidx + 1 < nsym && row[sidx + 1] <= slot) ++sidx;
            const bool pure = row[sidx + 1] >= slot + (1u << kShift);
            lut[(size_t)t * kBuckets + b] = (uint64_t)sf[sidx] | ((uint64_t)sidx << 32) | (pure ? (1ull << 47) : 0ull);
        }
        tabs[(size_t)t] = DecTable{row, sf, lut.data() + (size_t)t * kBuckets, cdf_sizes[t] - 2, offsets[t]};
    }
    const int64_t nwords = nbytes / 4;
    std::vector<uint32_t> words((size_t)nwords + 4, 0u);   // zero padding: reads past the end yield 0
    std::memcpy(words.data(), data, (size_t)nbytes);
    const uint32_t* p = words.data();
    const uint32_t* const end = p + nwords;
    uint64_t x = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
    p += 2;
    // branch-free renormalisation: the refill happens for roughly every other symbol, so a branch here
    // mispredicts constantly; words[] is zero-padded, and p is clamped so it never leaves the buffer
    auto refill = [&]() {
        const bool need = x < kRansL;
        const uint64_t xn = (x << 32) | *p;
        x = need ? xn : x;
        p += (need && p < end) ? 1 : 0;
    };
    auto get_bits = [&]() -> uint32_t { const uint32_t v = (uint32_t)(x & kBypassMax); x >>= kBypassBits; refill(); return v; };
    for (int64_t i = 0; i < n; ++i) {
        const DecTable& tb = tabs[(size_t)indexes[i]];
        const int32_t maxv = tb.maxv;
        const uint32_t cf = (uint32_t)(x & 0xFFFFu);
        const uint64_t e = tb.lut[cf >> kShift];
        int32_t s = (int32_t)((e >> 32) & 0x7FFFu);
        uint32_t sfe = (uint32_t)e;
        if (!(e >> 47)) {
            // == (first j with cdf[j] > cf) - 1; the row ends with 2^16 > cf, so the scan stops in range
            while (tb.cdf[s + 1] <= cf) ++s;
            sfe = tb.sf[s];
        }
        x = (uint64_t)(sfe >> 16) * (x >> kPrecision) + cf - (sfe & 0xFFFFu);
        refill();
        int32_t value = s;
        if (value == maxv) {
            uint32_t val = get_bits();
            int32_t nb = (int32_t)val;
            while (val == kBypassMax) { val = get_bits(); nb += (int32_t)val; }
            uint32_t raw = 0;
            for (int32_t j = 0; j < nb; ++j) raw |= get_bits() << (j * kBypassBits);
            value = (int32_t)(raw >> 1);
            value = (raw & 1u) ? -value - 1 : value + maxv;
        }
        const int32_t sym = value + tb.offset;
        out_symbols[i] = (SymT)sym;
        if (sizeof(SymT) < 4 && (int32_t)(SymT)sym != sym && narrowed) *narrowed = 1;
    }
    return PCC_OK;
}

extern "C" {

int64_t pcc_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, int64_t n, const int32_t* cdfs,
                                     int32_t cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets,
                                     uint8_t* out, int64_t out_cap) {
    return rans_encode<int32_t, int32_t>(symbols, indexes, n, cdfs, cdf_stride, cdf_sizes, offsets, out, out_cap);
}

int64_t pcc_rans_encode_with_indexes_i16u8(const int16_t* symbols, const uint8_t* indexes, int64_t n, const int32_t* cdfs,
                                           int32_t cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets,
                                           uint8_t* out, int64_t out_cap) {
    return rans_encode<int16_t, uint8_t>(symbols, indexes, n, cdfs, cdf_stride, cdf_sizes, offsets, out, out_cap);
}

int pcc_rans_decode_with_indexes(const uint8_t* data, int64_t nbytes, const int32_t* indexes, int64_t n,
                                 const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                                 const int32_t* offsets, int32_t* out_symbols) {
    return rans_decode<int32_t, int32_t>(data, nbytes, indexes, n, cdfs, cdf_stride, cdf_sizes, offsets, out_symbols, nullptr);
}

int pcc_rans_decode_with_indexes_u8i16(const uint8_t* data, int64_t nbytes, const uint8_t* indexes, int64_t n,
                                       const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                                       const int32_t* offsets, int16_t* out_symbols, int32_t* narrowed) {
    if (!narrowed) { pcc::set_error("pcc_rans_decode_with_indexes_u8i16: the narrowed flag is required"); return PCC_ERR_ARG; }
    *narrowed = 0;
    return rans_decode<int16_t, uint8_t>(data, nbytes, indexes, n, cdfs, cdf_stride, cdf_sizes, offsets, out_symbols, narrowed);
}

int pcc_pmf_to_quantized_cdf(const float* pmf, int32_t n, int32_t precision, int32_t* cdf) {
    if (n < 1 || precision < 1 || precision > 16) { pcc::set_error("pcc_pmf_to_quantized_cdf: bad arguments"); return PCC_ERR_ARG; }
    std::vector<uint32_t> c((size_t)n + 1, 0u);
    uint32_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (!(pmf[i] >= 0.0f) || !std::isfinite(pmf[i])) { pcc::set_error("pcc_pmf_to_quantized_cdf: invalid pmf[%d]", i); return PCC_ERR_DATA; }
        c[i + 1] = (uint32_t)std::round(pmf[i] * (float)(1 << precision));
        total += c[i + 1];
    }
    if (total == 0) { pcc::set_error("pcc_pmf_to_quantized_cdf: pmf sums to zero"); return PCC_ERR_DATA; }
    uint32_t run = 0;
    for (int i = 0; i <= n; ++i) {
        run += (uint32_t)((((uint64_t)1 << precision) * c[i]) / total);
        c[i] = run;
    }
    c[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (c[i] != c[i + 1]) continue;
        uint32_t best_freq = ~0u;
        int best = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t f = c[j + 1] - c[j];
            if (f > 1 && f < best_freq) { best_freq = f; best = j; }
        }
        if (best < 0) { pcc::set_error("pcc_pmf_to_quantized_cdf: cannot repair zero-width bin %d", i); return PCC_ERR_DATA; }
        if (best < i) for (int j = best + 1; j <= i; ++j) --c[j];
        else for (int j = i + 1; j <= best; ++j) ++c[j];
    }
    for (int i = 0; i <= n; ++i) cdf[i] = (int32_t)c[i];
    return PCC_OK;
}

}  // extern "C"
