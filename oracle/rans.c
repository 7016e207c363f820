/*
 * Oracle: range-ANS coder and quantised-CDF builder (plain C, CPU).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates the published
 * algorithm of the third-party coder the reference calls through
 * compressai==1.2.4 (requirements.txt:9; call sites
 * /root/reference/model/entropy_models.py:352-353,372,393,408):
 * ryg_rans `rans64.h` (64-bit state, lower bound 2^31, 32-bit renormalisation
 * words) driven by compressai's `rans_interface.cpp` (16-bit precision, 4-bit
 * bypass escape for out-of-table symbols) and `pmf_to_quantized_cdf`
 * (SURVEY.md Appendix B.4).  Source of those packages is NOT under
 * /root/reference: parity at this boundary is unpinned; the pure-Python twin
 * oracle/rans_py.py must agree with this file byte for byte.
 *
 * Deliberately simple: the forward pass materialises the (start, range, bypass)
 * queue exactly as described in B.4, the backward pass feeds the rANS state.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRECISION 16
#define BYPASS_PRECISION 4
#define MAX_BYPASS 15
#define RANS_L (1ull << 31)

typedef struct { uint16_t start; uint16_t range; uint8_t bypass; } sym_t;

typedef struct { sym_t *v; long n, cap; } queue_t;

static int q_push(queue_t *q, uint32_t start, uint32_t range, int bypass) {
    if (q->n == q->cap) {
        long nc = q->cap ? q->cap * 2 : 1024;
        sym_t *nv = (sym_t *)realloc(q->v, (size_t)nc * sizeof(sym_t));
        if (!nv) return -1;
        q->v = nv; q->cap = nc;
    }
    q->v[q->n].start = (uint16_t)start;
    q->v[q->n].range = (uint16_t)range;
    q->v[q->n].bypass = (uint8_t)bypass;
    q->n++;
    return 0;
}

/* Returns bytes written, -1 on allocation failure, -2 if out_cap too small. */
long pcc_oracle_rans_encode(const int32_t *symbols, const int32_t *indexes, long n,
                            const int32_t *cdfs, int cdf_stride, const int32_t *cdf_sizes,
                            const int32_t *offsets, uint8_t *out, long out_cap) {
    queue_t q = {0, 0, 0};
    for (long i = 0; i < n; ++i) {
        const int32_t ix = indexes[i];
        const int32_t *cdf = cdfs + (long)ix * cdf_stride;
        const int32_t maxv = cdf_sizes[ix] - 2;
        int32_t v = symbols[i] - offsets[ix];
        uint32_t raw = 0;
        if (v < 0) { raw = (uint32_t)(-2 * v - 1); v = maxv; }
        else if (v >= maxv) { raw = (uint32_t)(2 * (v - maxv)); v = maxv; }
        if (q_push(&q, (uint32_t)cdf[v], (uint32_t)(cdf[v + 1] - cdf[v]), 0)) { free(q.v); return -1; }
        if (v == maxv) {
            int32_t nb = 0;
            while ((raw >> (nb * BYPASS_PRECISION)) != 0) ++nb;
            int32_t t = nb;
            while (t >= MAX_BYPASS) { if (q_push(&q, MAX_BYPASS, MAX_BYPASS + 1, 1)) { free(q.v); return -1; } t -= MAX_BYPASS; }
            if (q_push(&q, (uint32_t)t, (uint32_t)t + 1, 1)) { free(q.v); return -1; }
            for (int32_t j = 0; j < nb; ++j) {
                uint32_t nib = (raw >> (j * BYPASS_PRECISION)) & MAX_BYPASS;
                if (q_push(&q, nib, nib + 1, 1)) { free(q.v); return -1; }
            }
        }
    }
    const long cap_words = q.n + 2;
    uint32_t *buf = (uint32_t *)malloc((size_t)cap_words * sizeof(uint32_t));
    if (!buf) { free(q.v); return -1; }
    uint32_t *ptr = buf + cap_words;
    uint64_t x = RANS_L;
    for (long i = q.n - 1; i >= 0; --i) {
        const sym_t s = q.v[i];
        if (!s.bypass) {
            const uint64_t freq = s.range;
            const uint64_t xmax = ((RANS_L >> PRECISION) << 32) * freq;
            if (x >= xmax) { *--ptr = (uint32_t)x; x >>= 32; }
            x = ((x / freq) << PRECISION) + (x % freq) + s.start;
        } else {
            const uint64_t freq = 1ull << (16 - BYPASS_PRECISION);
            const uint64_t xmax = ((RANS_L >> 16) << 32) * freq;
            if (x >= xmax) { *--ptr = (uint32_t)x; x >>= 32; }
            x = (x << BYPASS_PRECISION) | s.start;
        }
    }
    ptr -= 2;
    ptr[0] = (uint32_t)(x);
    ptr[1] = (uint32_t)(x >> 32);
    const long nbytes = (long)((buf + cap_words) - ptr) * 4;
    long rv = nbytes;
    if (nbytes > out_cap) rv = -2;
    else memcpy(out, ptr, (size_t)nbytes); /* little-endian host: bytes = LE uint32 words in stream order */
    free(buf); free(q.v);
    return rv;
}

static uint32_t get_bits(uint64_t *x, const uint32_t **p, const uint32_t *end) {
    uint32_t val = (uint32_t)(*x & ((1u << BYPASS_PRECISION) - 1));
    *x >>= BYPASS_PRECISION;
    if (*x < RANS_L) { uint32_t w = (*p < end) ? **p : 0; (*p)++; *x = (*x << 32) | w; }
    return val;
}

/* Returns 0 on success, -1 on malformed input (table walk failure). */
int pcc_oracle_rans_decode(const uint8_t *in, long nbytes, const int32_t *indexes, long n,
                           const int32_t *cdfs, int cdf_stride, const int32_t *cdf_sizes,
                           const int32_t *offsets, int32_t *out) {
    if (nbytes < 8 || (nbytes & 3)) return -1;
    const long nwords = nbytes / 4;
    uint32_t *words = (uint32_t *)malloc((size_t)nwords * 4);
    if (!words) return -1;
    memcpy(words, in, (size_t)nbytes);
    const uint32_t *p = words, *end = words + nwords;
    uint64_t x = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
    p += 2;
    for (long i = 0; i < n; ++i) {
        const int32_t ix = indexes[i];
        const int32_t *cdf = cdfs + (long)ix * cdf_stride;
        const int32_t size = cdf_sizes[ix];
        const int32_t maxv = size - 2;
        const uint32_t cf = (uint32_t)(x & 0xFFFFu);
        int32_t s = -1;
        for (int32_t j = 0; j < size; ++j) if ((uint32_t)cdf[j] > cf) { s = j - 1; break; }
        if (s < 0) { free(words); return -1; }
        x = (uint64_t)(cdf[s + 1] - cdf[s]) * (x >> PRECISION) + cf - (uint64_t)cdf[s];
        if (x < RANS_L) { uint32_t w = (p < end) ? *p : 0; p++; x = (x << 32) | w; }
        int32_t value = s;
        if (value == maxv) {
            int32_t val = (int32_t)get_bits(&x, &p, end);
            int32_t nb = val;
            while (val == MAX_BYPASS) { val = (int32_t)get_bits(&x, &p, end); nb += val; }
            uint32_t raw = 0;
            for (int32_t j = 0; j < nb; ++j) { val = (int32_t)get_bits(&x, &p, end); raw |= (uint32_t)val << (j * BYPASS_PRECISION); }
            value = (int32_t)(raw >> 1);
            if (raw & 1) value = -value - 1; else value += maxv;
        }
        out[i] = value + offsets[ix];
    }
    free(words);
    return 0;
}

/* cdf has n+1 entries.  Returns 0, or -1 on invalid pmf. */
int pcc_oracle_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf) {
    for (int i = 0; i < n; ++i) if (!(pmf[i] >= 0.0f) || !isfinite(pmf[i])) return -1;
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)roundf(pmf[i] * (float)(1 << precision));
    uint32_t total = 0;
    for (int i = 0; i <= n; ++i) total += cdf[i];
    if (total == 0) return -1;
    for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)((((uint64_t)1 << precision) * cdf[i]) / total);
    for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
    cdf[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            uint32_t best_freq = ~0u;
            int best = -1;
            for (int j = 0; j < n; ++j) {
                uint32_t f = cdf[j + 1] - cdf[j];
                if (f > 1 && f < best_freq) { best_freq = f; best = j; }
            }
            if (best < 0) return -1;
            if (best < i) { for (int j = best + 1; j <= i; ++j) cdf[j]--; }
            else { for (int j = i + 1; j <= best; ++j) cdf[j]++; }
        }
    }
    return 0;
}
