// Stable LSD radix sort of (key, int32 value) pairs, 8-bit digits (9-bit for the 27-bit execution-order keys: three passes
// instead of four), written for 64-wide wavefronts.
//
// Used for the MFMA execution order of a kernel map (32-bit key: neighbour-mask popcount | 27-bit mask; up to
// 5 M rows per coordinate set), the canonical (b, x, y, z) order of the latents (utils.sort_tensor /
// sort_points, utils.py:155-204; 64-bit voxel key) and the Morton order of the latent-coordinate coder.
// HBM-bound integer work: per pass a key is read twice and written once, a value read and
// written once; nothing here touches the matrix cores.
//
// Ranking.  A wave owns a contiguous chunk of the input and walks it in rounds of 64 keys.  Inside a round the
// lanes holding the same digit are found with 8 ballots (one per digit bit); a key's rank among them is the
// popcount of the lower lanes, and the chunk's running per-digit counter (LDS, wave-private) is advanced by the
// group's lowest lane.  Input order is (wave chunk, round, lane), so ranks are stable by construction and the
// result does not depend on arrival order: no atomics decide a position.
//
// Three shapes (measured on MI355X; what bounds a small sort is launch count and load latency, not bytes):
//   n <= 16 K         ONE launch: a 1024-thread workgroup runs all passes; a wave's chunk of the keys (<= 16 rounds of
//                     64) is loaded into registers once per pass, so the count sweep and the scatter sweep pay the
//                     global-load latency once instead of once per round;
//   n <= 512 K        per pass two launches: per-workgroup digit histograms (2048 keys per workgroup), then the
//                     scatter kernel, which sums the histograms of the workgroups in front of it itself (<= 256 of them);
//   larger            per pass three launches: histograms (4096 keys per workgroup), one scan workgroup per digit
//                     over that digit's row of counters, scatter.
#include "sort_small.h"

namespace pcc {

constexpr int RS_MID_ITEMS = 8;                   // keys per lane: 2048-key workgroups, self-prefixed scatter
constexpr int RS_MID_MAX_UNITS = 64;              // n <= 131,072 (a thread sums its digit's row of unit counters itself: 265 k rows took 150 us this way, 850 k rows 132 us the other)
constexpr int RS_BIG_ITEMS = 16;                  // 4096-key workgroups (48 KB of LDS for 64-bit keys: three per CU), row-scan kernel between count and scatter

template <class K, int ROUNDS, int DB>
__global__ __launch_bounds__(RS_SMALL_THREADS) void radix_sort_small_kernel(K* ka, K* kb, int32_t* va, int32_t* vb, int iota, int n,
                                                                           int begin_bit, int end_bit, int passes) {
    radix_sort_small_body<K, ROUNDS, DB>(ka, kb, va, vb, iota, n, begin_bit, end_bit, passes);
}

// counts[d * nunits + u] = number of keys of workgroup u (256 * ITEMS consecutive keys) with digit d
template <class K, int ITEMS, int DB>
__global__ __launch_bounds__(256) void radix_count_kernel(const K* __restrict__ keys, int64_t n, int shift, unsigned dmask,
                                                          int64_t nunits, int32_t* __restrict__ counts) {
    constexpr int ND = 1 << DB;
    __shared__ int cnt[ND];
#pragma unroll
    for (int d = threadIdx.x; d < ND; d += 256) cnt[d] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * (256 * ITEMS);
    K key[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int64_t i = base + (int64_t)j * 256 + threadIdx.x;
        key[j] = (i < n) ? keys[i] : (K)0;
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (base + (int64_t)j * 256 + threadIdx.x < n) atomicAdd(&cnt[(unsigned)(key[j] >> shift) & dmask], 1);
    __syncthreads();
#pragma unroll
    for (int d = threadIdx.x; d < ND; d += 256) counts[(int64_t)d * nunits + blockIdx.x] = cnt[d];
}

// one workgroup per digit: exclusive scan of the digit's row of `nunits` counters in place, row sum -> totals[digit]
__global__ __launch_bounds__(256) void radix_rowscan_kernel(int32_t* __restrict__ counts, int64_t nunits, int32_t* __restrict__ totals) {
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int32_t* row = counts + (int64_t)blockIdx.x * nunits;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nunits; base += 1024) {
        const int64_t i = base + 4 * (int64_t)t;
        int f[4];
        int v = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[j] = (i + j < nunits) ? row[i + j] : 0;
            v += f[j];
        }
        const int inc = wave_inclusive_scan_i32(v, lane);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int pre = carry_s;
        for (int ww = 0; ww < w; ++ww) pre += wsum[ww];
        int ex = pre + inc - v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i + j < nunits) row[i + j] = ex;
            ex += f[j];
        }
        __syncthreads();
        if (t == 255) carry_s = pre + inc;
        __syncthreads();
    }
    if (t == 0) totals[blockIdx.x] = carry_s;
}

// SELF_PREFIX: `counts` holds the raw histograms; thread d sums its digit's row in front of this workgroup and over all
// workgroups itself (nunits <= RS_MID_MAX_UNITS).  Otherwise `counts` is row-scanned and `totals` holds the row sums.
//
// The workgroup's pairs are first sorted LOCALLY into LDS (digit-major, input order inside a digit: the same stable ranks
// as before, relative to the workgroup instead of to the whole array) and then written out position by position, so that
// consecutive lanes write consecutive global positions: a digit's run of this workgroup (32 pairs on average at 8192 keys)
// leaves as whole 128-byte lines.  Until round 4 every lane wrote its pair straight to its global position — 64 lanes, up
// to 64 different lines per instruction, each (wave, digit) cursor advancing 4 bytes at a time over 32 rounds: with 4096
// waves x 256 cursors x 2 arrays in flight the partially written lines (256 MB) fell out of the 32 MB of L2 long before
// they were full, and a 5.16 M-pair pass took 0.42 ms for 82 MB (0.2 TB/s).
template <class K, int ITEMS, bool SELF_PREFIX, int DB>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const K* __restrict__ src, K* __restrict__ dst, const int32_t* __restrict__ vs,
                                                            int32_t* __restrict__ vd, int iota, int64_t n, int shift, unsigned dmask,
                                                            int64_t nunits, const int32_t* __restrict__ counts,
                                                            const int32_t* __restrict__ totals) {
    constexpr int WAVE_KEYS = 64 * ITEMS, WG_KEYS = 4 * WAVE_KEYS;
    constexpr int ND = 1 << DB, DH = ND / 256;       // digits; digits per thread (digit h * 256 + t, h < DH: ascending with (h, t))
    __shared__ K skey[WG_KEYS];
    __shared__ int32_t sval[WG_KEYS];
    __shared__ int cnt[4][ND];           // per wave: digit counts, then the wave's cursor into the local order
    __shared__ int lbase[ND];            // first local position of a digit
    __shared__ int gbase[ND];            // global position of the digit's first pair of this workgroup
    __shared__ int wsum[4], wsum2[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int i = t; i < 4 * ND; i += 256) (&cnt[0][0])[i] = 0;
    const int64_t base0 = (int64_t)blockIdx.x * WG_KEYS;
    const int64_t wbase = base0 + (int64_t)w * WAVE_KEYS;
    K key[ITEMS];
    int32_t val[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int64_t i = wbase + j * 64 + lane;
        key[j] = (i < n) ? src[i] : (K)0;
        val[j] = (i < n) ? (iota ? (int32_t)i : vs[i]) : 0;
    }
    // digits h * 256 + t: rows in front of this workgroup and the digit's total
    int pre[DH], total[DH];
#pragma unroll
    for (int h = 0; h < DH; ++h) {
        const int d = h * 256 + t;
        if (SELF_PREFIX) {
            const int32_t* row = counts + (int64_t)d * nunits;
            pre[h] = 0;
            total[h] = 0;
            for (int64_t u = 0; u < nunits; ++u) {
                const int c = row[u];
                total[h] += c;
                pre[h] += (u < (int64_t)blockIdx.x) ? c : 0;
            }
        } else {
            pre[h] = counts[(int64_t)d * nunits + blockIdx.x];
            total[h] = totals[d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (wbase + j * 64 + lane < n) atomicAdd(&cnt[w][(unsigned)(key[j] >> shift) & dmask], 1);
    __syncthreads();
    // digit d: its global base for this workgroup (exclusive scan of the totals over the digits + the rows in front); its local
    // base (exclusive scan of the workgroup's digit counts); then each wave's cursor into the local order (waves in input order)
    int gcarry = 0, lcarry = 0;
#pragma unroll
    for (int h = 0; h < DH; ++h) {
        const int d = h * 256 + t;
        const int c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
        const int mine = c0 + c1 + c2 + c3;
        const int inc = wave_inclusive_scan_i32(total[h], lane);
        const int linc = wave_inclusive_scan_i32(mine, lane);
        if (lane == 63) { wsum[w] = inc; wsum2[w] = linc; }
        __syncthreads();
        int run = gcarry + pre[h] + inc - total[h], lrun = lcarry + linc - mine;
        for (int ww = 0; ww < w; ++ww) { run += wsum[ww]; lrun += wsum2[ww]; }
        gcarry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        lcarry += wsum2[0] + wsum2[1] + wsum2[2] + wsum2[3];
        gbase[d] = run;
        lbase[d] = lrun;
        cnt[0][d] = lrun;
        cnt[1][d] = lrun + c0;
        cnt[2][d] = lrun + c0 + c1;
        cnt[3][d] = lrun + c0 + c1 + c2;
        __syncthreads();                                   // wsum / wsum2 are rewritten by the next half
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const bool active = wbase + j * 64 + lane < n;
        const unsigned d = (unsigned)(key[j] >> shift) & dmask;
        const uint64_t same = match_digit<DB>(d, active);
        const int below = __popcll(same & lt);
        int base = 0;
        if (active) {
            base = cnt[w][d];
            skey[base + below] = key[j];
            sval[base + below] = val[j];
        }
        __builtin_amdgcn_wave_barrier();
        if (active && below == 0) cnt[w][d] = base + __popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const int nk = (int)((n - base0 < WG_KEYS) ? (n - base0) : WG_KEYS);
    for (int i = t; i < nk; i += 256) {
        const K k = skey[i];
        const unsigned d = (unsigned)(k >> shift) & dmask;
        const int64_t pos = (int64_t)gbase[d] + (i - lbase[d]);
        dst[pos] = k;
        vd[pos] = sval[i];
    }
}

int radix_sort_passes(int begin_bit, int end_bit, int digit_bits) {
    return end_bit > begin_bit ? (end_bit - begin_bit + digit_bits - 1) / digit_bits : 0;
}

// 27-bit execution-order keys: three 9-bit passes instead of four 8-bit ones — except in the mid shape, whose scatter sums its
// digits' rows of unit counters itself (twice the rows per thread with 512 digits: 9.4 -> 14.6 us per pass, a loss)
int radix_sort_order_digit_bits(int64_t n) {
    const int64_t mid_units = (n + 256 * RS_MID_ITEMS - 1) / (256 * RS_MID_ITEMS);
    return (n > RS_SMALL_N && mid_units <= RS_MID_MAX_UNITS) ? 8 : 9;
}

int64_t radix_sort_counter_bytes(int64_t n) {
    const int64_t mid = (n + 256 * RS_MID_ITEMS - 1) / (256 * RS_MID_ITEMS), big = (n + 256 * RS_BIG_ITEMS - 1) / (256 * RS_BIG_ITEMS);
    const int64_t nunits = mid <= RS_MID_MAX_UNITS ? mid : big;
    return align256(512 * (nunits > 0 ? nunits : 1) * 4) + 2048;       // counters + row totals, for up to 512 digits
}

template <class K, int DB>
static int radix_sort_pairs(K* ka, K* kb, int32_t* va, int32_t* vb, bool iota, int64_t n, int begin_bit, int end_bit, void* counters,
                            hipStream_t st) {
    constexpr int ND = 1 << DB;
    const int passes = radix_sort_passes(begin_bit, end_bit, DB);
    PCC_REQUIRE(passes >= 1 && end_bit <= (int)(8 * sizeof(K)) && begin_bit >= 0, "radix sort: bad bit range [%d, %d)", begin_bit, end_bit);
    PCC_REQUIRE(n < (1ll << 31), "radix sort: too many keys (%lld)", (long long)n);
    if (n <= 0) return PCC_OK;
    if (n <= RS_SMALL_N) {
        const int rounds = (int)(((n + RS_SMALL_WAVES - 1) / RS_SMALL_WAVES + 63) / 64);
#define PCC_RS_SMALL(R)                                                                                                             \
    hipLaunchKernelGGL((radix_sort_small_kernel<K, R, DB>), dim3(1), dim3(RS_SMALL_THREADS), 0, st, ka, kb, va, vb, iota ? 1 : 0, (int)n, \
                       begin_bit, end_bit, passes)
        if (rounds <= 1) PCC_RS_SMALL(1);
        else if (rounds <= 2) PCC_RS_SMALL(2);
        else if (rounds <= 4) PCC_RS_SMALL(4);
        else if (rounds <= 8) PCC_RS_SMALL(8);
        else PCC_RS_SMALL(16);
#undef PCC_RS_SMALL
        PCC_LAUNCH_CHECK();
        return PCC_OK;
    }
    const int64_t mid_units = (n + 256 * RS_MID_ITEMS - 1) / (256 * RS_MID_ITEMS);
    const bool mid = mid_units <= RS_MID_MAX_UNITS;
    const int64_t nunits = mid ? mid_units : (n + 256 * RS_BIG_ITEMS - 1) / (256 * RS_BIG_ITEMS);
    int32_t* counts = reinterpret_cast<int32_t*>(counters);
    int32_t* totals = counts + align256((int64_t)ND * nunits * 4) / 4;
    K* src = ka;
    K* dst = kb;
    int32_t* vs = va;
    int32_t* vd = vb;
    for (int p = 0; p < passes; ++p) {
        const int shift = begin_bit + DB * p;
        const unsigned dmask = (end_bit - shift >= DB) ? (unsigned)(ND - 1) : ((1u << (end_bit - shift)) - 1u);
        const int io = (iota && p == 0) ? 1 : 0;
        if (mid) {
            hipLaunchKernelGGL((radix_count_kernel<K, RS_MID_ITEMS, DB>), dim3((unsigned)nunits), dim3(256), 0, st, src, n, shift, dmask, nunits, counts);
            hipLaunchKernelGGL((radix_scatter_kernel<K, RS_MID_ITEMS, true, DB>), dim3((unsigned)nunits), dim3(256), 0, st, src, dst, vs, vd, io, n,
                               shift, dmask, nunits, counts, totals);
        } else {
            hipLaunchKernelGGL((radix_count_kernel<K, RS_BIG_ITEMS, DB>), dim3((unsigned)nunits), dim3(256), 0, st, src, n, shift, dmask, nunits, counts);
            hipLaunchKernelGGL(radix_rowscan_kernel, dim3(ND), dim3(256), 0, st, counts, nunits, totals);
            hipLaunchKernelGGL((radix_scatter_kernel<K, RS_BIG_ITEMS, false, DB>), dim3((unsigned)nunits), dim3(256), 0, st, src, dst, vs, vd, io, n,
                               shift, dmask, nunits, counts, totals);
        }
        K* tk = src; src = dst; dst = tk;
        int32_t* tv = vs; vs = vd; vd = tv;
    }
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* keys_b, int32_t* vals_a, int32_t* vals_b, bool vals_are_iota, int64_t n, int begin_bit,
                         int end_bit, void* counters, hipStream_t st, int digit_bits) {
    if (digit_bits == 9) return radix_sort_pairs<uint32_t, 9>(keys_a, keys_b, vals_a, vals_b, vals_are_iota, n, begin_bit, end_bit, counters, st);
    return radix_sort_pairs<uint32_t, 8>(keys_a, keys_b, vals_a, vals_b, vals_are_iota, n, begin_bit, end_bit, counters, st);
}

int radix_sort_pairs_u64(uint64_t* keys_a, uint64_t* keys_b, int32_t* vals_a, int32_t* vals_b, bool vals_are_iota, int64_t n, int begin_bit,
                         int end_bit, void* counters, hipStream_t st) {
    return radix_sort_pairs<uint64_t, 8>(keys_a, keys_b, vals_a, vals_b, vals_are_iota, n, begin_bit, end_bit, counters, st);
}

}  // namespace pcc
