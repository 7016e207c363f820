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
// Encoder: single reverse sweep over the symbols (no intermediate symbol queue); the escape
// nibbles of an out-of-range symbol are emitted in reverse of their forward order.  Decoder:
// binary search in the (non-decreasing) CDF row instead of the linear scan.
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

struct Writer {
    uint32_t* base;
    uint32_t* ptr;  // grows downwards
    inline bool room() const { return ptr > base; }
};

inline void put_sym(uint64_t& x, Writer& w, uint32_t start, uint32_t freq) {
    const uint64_t x_max = ((kRansL >> kPrecision) << 32) * freq;
    if (x >= x_max) { *--w.ptr = (uint32_t)x; x >>= 32; }
    x = ((x / freq) << kPrecision) + (x % freq) + start;
}

inline void put_bits(uint64_t& x, Writer& w, uint32_t val) {
    const uint64_t x_max = ((kRansL >> 16) << 32) * (1ull << (16 - kBypassBits));
    if (x >= x_max) { *--w.ptr = (uint32_t)x; x >>= 32; }
    x = (x << kBypassBits) | val;
}

}  // namespace

extern "C" {

int64_t pcc_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, int64_t n, const int32_t* cdfs,
                                     int32_t cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets,
                                     uint8_t* out, int64_t out_cap) {
    if (n < 0 || !out) { pcc::set_error("pcc_rans_encode_with_indexes: bad arguments"); return PCC_ERR_ARG; }
    // worst case: every symbol escapes with a 32-bit raw value: 1 + 1 + 8 nibble items + main = < 3 words
    const int64_t cap_words = 3 * n + 4;
    std::vector<uint32_t> buf((size_t)cap_words);
    Writer w{buf.data(), buf.data() + cap_words};
    uint64_t x = kRansL;
    for (int64_t i = n - 1; i >= 0; --i) {
        const int32_t ix = indexes[i];
        const int32_t* cdf = cdfs + (int64_t)ix * cdf_stride;
        const int32_t maxv = cdf_sizes[ix] - 2;
        int32_t v = symbols[i] - offsets[ix];
        uint32_t raw = 0;
        bool esc = false;
        if (v < 0) { raw = (uint32_t)(-2 * v - 1); v = maxv; esc = true; }
        else if (v >= maxv) { raw = (uint32_t)(2 * (v - maxv)); v = maxv; esc = true; }
        if (esc) {
            int nb = 0;
            while ((raw >> (nb * kBypassBits)) != 0) ++nb;
            // forward order: main, count chunks (15,15,...,rest), nibbles LSB first  => reverse here
            for (int j = nb - 1; j >= 0; --j) put_bits(x, w, (raw >> (j * kBypassBits)) & kBypassMax);
            int full = nb / (int)kBypassMax, rest = nb % (int)kBypassMax;
            put_bits(x, w, (uint32_t)rest);
            for (int j = 0; j < full; ++j) put_bits(x, w, kBypassMax);
        }
        const uint32_t start = (uint32_t)cdf[v] & 0xFFFFu;
        const uint32_t freq = (uint32_t)(cdf[v + 1] - cdf[v]) & 0xFFFFu;
        if (freq == 0) { pcc::set_error("pcc_rans_encode_with_indexes: zero-frequency symbol at %lld", (long long)i); return PCC_ERR_DATA; }
        put_sym(x, w, start, freq);
    }
    *--w.ptr = (uint32_t)(x >> 32);
    *--w.ptr = (uint32_t)x;
    const int64_t nbytes = (int64_t)((buf.data() + cap_words) - w.ptr) * 4;
    if (nbytes > out_cap) { pcc::set_error("pcc_rans_encode_with_indexes: output buffer too small (%lld > %lld)", (long long)nbytes, (long long)out_cap); return PCC_ERR_ARG; }
    std::memcpy(out, w.ptr, (size_t)nbytes);
    return nbytes;
}

int pcc_rans_decode_with_indexes(const uint8_t* data, int64_t nbytes, const int32_t* indexes, int64_t n,
                                 const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                                 const int32_t* offsets, int32_t* out_symbols) {
    if (nbytes < 8 || (nbytes & 3)) { pcc::set_error("pcc_rans_decode_with_indexes: malformed stream length %lld", (long long)nbytes); return PCC_ERR_DATA; }
    const int64_t nwords = nbytes / 4;
    std::vector<uint32_t> words((size_t)nwords);
    std::memcpy(words.data(), data, (size_t)nbytes);
    const uint32_t* p = words.data();
    const uint32_t* const end = p + nwords;
    uint64_t x = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
    p += 2;
    auto refill = [&]() { if (x < kRansL) { const uint32_t wv = (p < end) ? *p : 0u; ++p; x = (x << 32) | wv; } };
    auto get_bits = [&]() -> uint32_t { const uint32_t v = (uint32_t)(x & kBypassMax); x >>= kBypassBits; refill(); return v; };
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ix = indexes[i];
        const int32_t* cdf = cdfs + (int64_t)ix * cdf_stride;
        const int32_t size = cdf_sizes[ix];
        const int32_t maxv = size - 2;
        const uint32_t cf = (uint32_t)(x & 0xFFFFu);
        // first j with cdf[j] > cf  (upper bound on a non-decreasing row)
        int32_t lo = 0, hi = size;
        while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if ((uint32_t)cdf[mid] > cf) hi = mid; else lo = mid + 1; }
        if (lo == 0 || lo >= size) { pcc::set_error("pcc_rans_decode_with_indexes: corrupt stream at symbol %lld", (long long)i); return PCC_ERR_DATA; }
        const int32_t s = lo - 1;
        x = (uint64_t)(uint32_t)(cdf[s + 1] - cdf[s]) * (x >> kPrecision) + cf - (uint32_t)cdf[s];
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
        out_symbols[i] = value + offsets[ix];
    }
    return PCC_OK;
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
