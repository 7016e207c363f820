// Stable LSD radix sort of (key, int32 value) pairs, 8-bit digits, written for 64-wide wavefronts.
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
// Two shapes:
//   n <= RS_SMALL_N   one 1024-thread workgroup runs all passes (count sweep, scan in LDS, scatter sweep per
//                     pass, ping-ponging between the two buffers): ONE launch for the 19 k / 72 k-row sets,
//                     where a multi-kernel sort is nothing but launch latency;
//   larger            per pass: count kernel (per-workgroup digit histograms, 8192 keys per workgroup) ->
//                     one-workgroup scan of the 256 x workgroups counters -> scatter kernel (keys of the
//                     workgroup held in registers between its count and its scatter sweep).
#include "sort.h"

namespace pcc {

constexpr int RS_SMALL_THREADS = 1024;
constexpr int RS_SMALL_WAVES = RS_SMALL_THREADS / 64;
constexpr int64_t RS_SMALL_N = 98304;
constexpr int RS_ITEMS = 32;                    // keys per lane in the multi-workgroup kernels
constexpr int RS_WAVE_KEYS = 64 * RS_ITEMS;     // 2048: a wave's contiguous chunk
constexpr int RS_WG_KEYS = 4 * RS_WAVE_KEYS;    // 8192 keys per 256-thread workgroup

// lanes of this wave that are active and hold the same 8-bit digit as the calling lane
__device__ __forceinline__ uint64_t match_digit(unsigned d, bool active) {
    uint64_t same = __ballot(active);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
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

template <class K>
__global__ __launch_bounds__(RS_SMALL_THREADS) void radix_sort_small_kernel(K* ka, K* kb, int32_t* va, int32_t* vb, int iota, int n,
                                                                           int begin_bit, int end_bit, int passes) {
    __shared__ int cnt[RS_SMALL_WAVES][256];
    __shared__ int tot[256];
    __shared__ int wsum[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int chunk = ((n + RS_SMALL_WAVES - 1) / RS_SMALL_WAVES + 63) / 64 * 64;
    const int lo = w * chunk < n ? w * chunk : n;
    const int hi = lo + chunk < n ? lo + chunk : n;
    const uint64_t lt = (1ull << lane) - 1ull;
    K* src = ka;
    K* dst = kb;
    int32_t* vs = va;
    int32_t* vd = vb;
    for (int p = 0; p < passes; ++p) {
        const int shift = begin_bit + 8 * p;
        const unsigned dmask = (end_bit - shift >= 8) ? 255u : ((1u << (end_bit - shift)) - 1u);
        for (int i = t; i < RS_SMALL_WAVES * 256; i += RS_SMALL_THREADS) (&cnt[0][0])[i] = 0;
        __syncthreads();
        for (int i = lo + lane; i < hi; i += 64) atomicAdd(&cnt[w][(unsigned)(src[i] >> shift) & dmask], 1);
        __syncthreads();
        // digit d: chunk bases (exclusive over the waves, input order) and the digit's total
        int v = 0;
        if (t < 256) {
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
        if (lane == 63 && w < 4) wsum[w] = inc;
        __syncthreads();
        if (t < 256) {
            int base = 0;
            for (int ww = 0; ww < w; ++ww) base += wsum[ww];
            tot[t] = base + inc - v;
        }
        __syncthreads();
        for (int i0 = lo; i0 < hi; i0 += 64) {
            const int i = i0 + lane;
            const bool active = i < hi;
            const K key = active ? src[i] : (K)0;
            const unsigned d = (unsigned)(key >> shift) & dmask;
            const uint64_t same = match_digit(d, active);
            const int below = __popcll(same & lt);
            int base = 0;
            if (active) {
                base = cnt[w][d];
                const int pos = tot[d] + base + below;
                dst[pos] = key;
                vd[pos] = (iota && p == 0) ? i : vs[i];
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

// counts[d * nunits + u] = number of keys of workgroup u with digit d
template <class K>
__global__ __launch_bounds__(256) void radix_count_kernel(const K* __restrict__ keys, int64_t n, int shift, unsigned dmask,
                                                          int64_t nunits, int32_t* __restrict__ counts) {
    __shared__ int cnt[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_WG_KEYS;
#pragma unroll 8
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = base + (int64_t)j * 256 + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(unsigned)(keys[i] >> shift) & dmask], 1);
    }
    __syncthreads();
    counts[(int64_t)threadIdx.x * nunits + blockIdx.x] = cnt[threadIdx.x];
}

// in-place exclusive scan of m ints by one workgroup (m = 256 x workgroups of the sort: a few 100 k at most)
__global__ __launch_bounds__(1024) void radix_scan_kernel(int32_t* __restrict__ a, int64_t m) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < m; base += 4096) {
        const int64_t i = base + 4 * (int64_t)t;
        int f[4];
        int v = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[j] = (i + j < m) ? a[i + j] : 0;
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
            if (i + j < m) a[i + j] = ex;
            ex += f[j];
        }
        __syncthreads();
        if (t == 1023) carry_s = pre + inc;
        __syncthreads();
    }
}

template <class K>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const K* __restrict__ src, K* __restrict__ dst, const int32_t* __restrict__ vs,
                                                            int32_t* __restrict__ vd, int iota, int64_t n, int shift, unsigned dmask,
                                                            int64_t nunits, const int32_t* __restrict__ scanned) {
    __shared__ int cnt[4][256];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int i = t; i < 4 * 256; i += 256) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const int64_t wbase = (int64_t)blockIdx.x * RS_WG_KEYS + (int64_t)w * RS_WAVE_KEYS;
    K key[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = wbase + j * 64 + lane;
        key[j] = (i < n) ? src[i] : (K)0;
        if (i < n) atomicAdd(&cnt[w][(unsigned)(key[j] >> shift) & dmask], 1);
    }
    __syncthreads();
    {   // digit t: global base of the workgroup, then of each of its waves (input order)
        int run = scanned[(int64_t)t * nunits + blockIdx.x];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const int c = cnt[ww][t];
            cnt[ww][t] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = wbase + j * 64 + lane;
        const bool active = i < n;
        const unsigned d = (unsigned)(key[j] >> shift) & dmask;
        const uint64_t same = match_digit(d, active);
        const int below = __popcll(same & lt);
        int base = 0;
        if (active) {
            base = cnt[w][d];
            const int64_t pos = (int64_t)base + below;
            dst[pos] = key[j];
            vd[pos] = iota ? (int32_t)i : vs[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (active && below == 0) cnt[w][d] = base + __popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
}

int radix_sort_passes(int begin_bit, int end_bit) { return end_bit > begin_bit ? (end_bit - begin_bit + 7) / 8 : 0; }

int64_t radix_sort_counter_bytes(int64_t n) {
    const int64_t nunits = (n + RS_WG_KEYS - 1) / RS_WG_KEYS;
    return align256(256 * (nunits > 0 ? nunits : 1) * 4);
}

template <class K>
static int radix_sort_pairs(K* ka, K* kb, int32_t* va, int32_t* vb, bool iota, int64_t n, int begin_bit, int end_bit, void* counters,
                            hipStream_t st) {
    const int passes = radix_sort_passes(begin_bit, end_bit);
    PCC_REQUIRE(passes >= 1 && end_bit <= (int)(8 * sizeof(K)) && begin_bit >= 0, "radix sort: bad bit range [%d, %d)", begin_bit, end_bit);
    PCC_REQUIRE(n < (1ll << 31), "radix sort: too many keys (%lld)", (long long)n);
    if (n <= 0) return PCC_OK;
    if (n <= RS_SMALL_N) {
        hipLaunchKernelGGL(radix_sort_small_kernel<K>, dim3(1), dim3(RS_SMALL_THREADS), 0, st, ka, kb, va, vb, iota ? 1 : 0, (int)n, begin_bit,
                           end_bit, passes);
        PCC_LAUNCH_CHECK();
        return PCC_OK;
    }
    const int64_t nunits = (n + RS_WG_KEYS - 1) / RS_WG_KEYS;
    int32_t* counts = reinterpret_cast<int32_t*>(counters);
    K* src = ka;
    K* dst = kb;
    int32_t* vs = va;
    int32_t* vd = vb;
    for (int p = 0; p < passes; ++p) {
        const int shift = begin_bit + 8 * p;
        const unsigned dmask = (end_bit - shift >= 8) ? 255u : ((1u << (end_bit - shift)) - 1u);
        hipLaunchKernelGGL(radix_count_kernel<K>, dim3((unsigned)nunits), dim3(256), 0, st, src, n, shift, dmask, nunits, counts);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(1024), 0, st, counts, 256 * nunits);
        hipLaunchKernelGGL(radix_scatter_kernel<K>, dim3((unsigned)nunits), dim3(256), 0, st, src, dst, vs, vd, (iota && p == 0) ? 1 : 0, n, shift,
                           dmask, nunits, counts);
        K* tk = src; src = dst; dst = tk;
        int32_t* tv = vs; vs = vd; vd = tv;
    }
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* keys_b, int32_t* vals_a, int32_t* vals_b, bool vals_are_iota, int64_t n, int begin_bit,
                         int end_bit, void* counters, hipStream_t st) {
    return radix_sort_pairs<uint32_t>(keys_a, keys_b, vals_a, vals_b, vals_are_iota, n, begin_bit, end_bit, counters, st);
}

int radix_sort_pairs_u64(uint64_t* keys_a, uint64_t* keys_b, int32_t* vals_a, int32_t* vals_b, bool vals_are_iota, int64_t n, int begin_bit,
                         int end_bit, void* counters, hipStream_t st) {
    return radix_sort_pairs<uint64_t>(keys_a, keys_b, vals_a, vals_b, vals_are_iota, n, begin_bit, end_bit, counters, st);
}

}  // namespace pcc
