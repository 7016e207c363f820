// The one-workgroup radix sort (n <= RS_SMALL_N keys, 1024 threads, all passes in one launch) as a device function: the
// body of radix_sort_small_kernel (sort.hip) and a phase of small_map_kernel (select.hip), which builds a small kernel
// map and its execution order in a single launch.  Every thread of a RS_SMALL_THREADS-thread workgroup must call it.
#pragma once
#include "sort.h"

namespace pcc {

constexpr int RS_SMALL_THREADS = 1024;
constexpr int RS_SMALL_WAVES = RS_SMALL_THREADS / 64;
constexpr int RS_SMALL_ROUNDS = 16;                                       // rounds of 64 keys per wave, held in registers
constexpr int64_t RS_SMALL_N = (int64_t)RS_SMALL_WAVES * 64 * RS_SMALL_ROUNDS;   // 16384

// lanes of this wave that are active and hold the same DB-bit digit as the calling lane
template <int DB = 8>
__device__ __forceinline__ uint64_t match_digit(unsigned d, bool active) {
    uint64_t same = __ballot(active);
#pragma unroll
    for (int b = 0; b < DB; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(active && bit);
        same &= bit ? bal : ~bal;
    }
    return same;
}

__device__ __forceinline__ int wave_inclusive_scan_i32(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// ROUNDS: rounds of 64 keys per wave (compile-time: the sweeps are unrolled over the register-resident keys; a 1.2 k-row
// set runs the 1-round instance instead of 16 mostly predicated-off rounds).  DB: digit bits — 8, or 9 for the 27-bit
// execution-order keys (three passes instead of four: 32 KB of counters instead of 16).
template <class K, int ROUNDS, int DB = 8>
__device__ __forceinline__ void radix_sort_small_body(K* ka, K* kb, int32_t* va, int32_t* vb, int iota, int n, int begin_bit, int end_bit,
                                                      int passes) {
    constexpr int ND = 1 << DB;
    static_assert(ND <= RS_SMALL_THREADS, "one thread per digit");
    __shared__ int cnt[RS_SMALL_WAVES][ND];
    __shared__ int tot[ND];
    __shared__ int wsum[ND / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int chunk = ((n + RS_SMALL_WAVES - 1) / RS_SMALL_WAVES + 63) / 64 * 64;      // <= 64 * ROUNDS
    const int lo = w * chunk < n ? w * chunk : n;
    const int hi = lo + chunk < n ? lo + chunk : n;
    const uint64_t lt = (1ull << lane) - 1ull;
    K* src = ka;
    K* dst = kb;
    int32_t* vs = va;
    int32_t* vd = vb;
    for (int p = 0; p < passes; ++p) {
        const int shift = begin_bit + DB * p;
        const unsigned dmask = (end_bit - shift >= DB) ? (unsigned)(ND - 1) : ((1u << (end_bit - shift)) - 1u);
        // the wave's chunk in registers: one batch of loads per pass
        K key[ROUNDS];
        int32_t val[ROUNDS];
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const int i = lo + j * 64 + lane;
            key[j] = (i < hi) ? src[i] : (K)0;
            val[j] = (i < hi) ? ((iota && p == 0) ? i : vs[i]) : 0;
        }
        for (int i = t; i < RS_SMALL_WAVES * ND; i += RS_SMALL_THREADS) (&cnt[0][0])[i] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j)
            if (lo + j * 64 + lane < hi) atomicAdd(&cnt[w][(unsigned)(key[j] >> shift) & dmask], 1);
        __syncthreads();
        // digit t: chunk bases (exclusive over the waves, input order) and the digit's total
        int v = 0;
        if (t < ND) {
            int run = 0;
#pragma unroll
            for (int ww = 0; ww < RS_SMALL_WAVES; ++ww) {
                const int c = cnt[ww][t];
                cnt[ww][t] = run;
                run += c;
            }
            v = run;
        }
        const int inc = wave_inclusive_scan_i32(v, lane);
        if (lane == 63 && w < ND / 64) wsum[w] = inc;
        __syncthreads();
        if (t < ND) {
            int base = 0;
            for (int ww = 0; ww < w; ++ww) base += wsum[ww];
            tot[t] = base + inc - v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const int i = lo + j * 64 + lane;
            const bool active = i < hi;
            const unsigned d = (unsigned)(key[j] >> shift) & dmask;
            const uint64_t same = match_digit<DB>(d, active);
            const int below = __popcll(same & lt);
            int base = 0;
            if (active) {
                base = cnt[w][d];
                const int pos = tot[d] + base + below;
                dst[pos] = key[j];
                vd[pos] = val[j];
            }
            __builtin_amdgcn_wave_barrier();          // every lane has read its counter before a leader advances it
            if (active && below == 0) cnt[w][d] = base + __popcll(same);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();                               // the pass's output is complete (workgroup scope) before it is read
        K* tk = src; src = dst; dst = tk;
        int32_t* tv = vs; vs = vd; vd = tv;
    }
}

}  // namespace pcc
