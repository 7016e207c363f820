// Per-batch top-k occupancy selection (GenerativeUpBlock._topk_prediction, model/blocks.py:130-150)
// and canonical coordinate sort (utils.sort_tensor / sort_points, utils.py:155-204).
//
// top-k = MSB-first radix select over a 96-bit composite key
//     hi32 = order-preserving map of the fp32 logit (NaN on top, as torch.topk)
//     lo64 = ~voxel_key   (exact logit ties -> ascending voxel key wins)
// so the selected SET is a pure function of (logit, coordinate): independent of row order and
// thread arrival order.  No host synchronisation: all passes are enqueued; a batch that has
// resolved early turns the remaining passes into no-ops through its `done` word.
// HBM-bound wavefront-level integer work, no dense contraction.
#include <stdlib.h>

#include "common.h"
#include "sort_small.h"

namespace pcc {

constexpr int TK_STRIDE = 264;  // int32 per batch: krem, done, nbytes, p0, p1, p2, pad, pad, hist[256]
constexpr int TK_LDS_BATCHES = 16;

__device__ __forceinline__ uint32_t float_key(float v) {
    if (v != v) return 0xFFFFFFFFu;
    if (v == 0.0f) v = 0.0f;  // -0 == +0
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Key96 { uint32_t w[3]; };

__device__ __forceinline__ Key96 make_key(float logit, int4 c) {
    const uint64_t lo = ~pack_key(c.x, c.y, c.z, c.w);
    Key96 k;
    k.w[0] = float_key(logit);
    k.w[1] = (uint32_t)(lo >> 32);
    k.w[2] = (uint32_t)lo;
    return k;
}

__device__ __forceinline__ uint32_t top_bytes_mask(int nb, int word) {
    int n = nb - 4 * word;
    n = n < 0 ? 0 : (n > 4 ? 4 : n);
    return n == 0 ? 0u : (0xFFFFFFFFu << (8 * (4 - n)));
}

// compare the top `nb` bytes of key with prefix: -1 / 0 / +1
__device__ __forceinline__ int cmp_prefix(const Key96& k, const int32_t* st, int nb) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t m = top_bytes_mask(nb, j);
        const uint32_t a = k.w[j] & m, b = (uint32_t)st[3 + j] & m;
        if (a != b) return a > b ? 1 : -1;
    }
    return 0;
}

__global__ void topk_init(const int32_t* __restrict__ k, int nbatch, int32_t* __restrict__ state) {
    const int b = blockIdx.x;
    int32_t* st = state + (int64_t)b * TK_STRIDE;
    for (int i = threadIdx.x; i < TK_STRIDE; i += blockDim.x) st[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int kk = k[b];
        st[0] = kk;
        if (kk <= 0) { st[1] = 1; st[2] = 13; }  // select nothing
    }
}

__global__ __launch_bounds__(256) void topk_hist(const float* __restrict__ logits, int ld,
                                                 const int32_t* __restrict__ coords, int64_t n, int nbatch, int pass,
                                                 int32_t* __restrict__ state) {
    __shared__ int lh[TK_LDS_BATCHES * 256];
    const bool use_lds = nbatch <= TK_LDS_BATCHES;
    if (use_lds) {
        // every item resolved (distinct logits resolve within the four passes over the fp32 key): the remaining
        // passes are enqueued all the same (no host sync) and return here without touching the rows
        bool all_done = true;
        for (int b = 0; b < nbatch; ++b) all_done &= state[(int64_t)b * TK_STRIDE + 1] != 0;
        if (all_done) return;
        for (int i = threadIdx.x; i < nbatch * 256; i += 256) lh[i] = 0;
        __syncthreads();
    }
    // one batch item and a pass over the logit bytes: the coordinates (item index, tie-break key) are not needed
    const bool logit_only = nbatch == 1 && pass < 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int4 c = logit_only ? make_int4(0, 0, 0, 0) : reinterpret_cast<const int4*>(coords)[i];
        const int b = c.x;
        if (b < 0 || b >= nbatch) continue;
        const int32_t* st = state + (int64_t)b * TK_STRIDE;
        if (st[1]) continue;
        const Key96 key = make_key(logits[i * ld], c);
        if (cmp_prefix(key, st, pass) != 0) continue;
        const uint32_t digit = (key.w[pass >> 2] >> (8 * (3 - (pass & 3)))) & 0xFFu;
        if (use_lds && nbatch == 1) {
            // Occupancy logits of one frame share their sign and exponent: in the first passes nearly every row of a wave
            // falls into the same one or two bins, and 64 LDS atomics on one address serialise.  The lanes of a wave that
            // hold the same digit are found with ballots (match_digit) and ONE lane adds their count.  (The lanes that skipped
            // this row above take no part in the ballots: the loop runs with the wave's current exec mask.)
            const uint64_t same = match_digit(digit, true);
            if ((same & ((1ull << (threadIdx.x & 63)) - 1ull)) == 0ull) atomicAdd(&lh[digit], __popcll(same));
        } else if (use_lds) {
            atomicAdd(&lh[b * 256 + digit], 1);
        } else {
            atomicAdd(&state[(int64_t)b * TK_STRIDE + 8 + digit], 1);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < nbatch * 256; i += 256) {
            const int v = lh[i];
            if (v) atomicAdd(&state[(int64_t)(i >> 8) * TK_STRIDE + 8 + (i & 255)], v);
        }
    }
}

// One batch item, passes 0-3 (the four bytes of the logit key).  topk_hist ends with every workgroup adding its bins to the SAME 256
// global words: same-address atomics from eight XCDs are performed one after the other at the memory side — ~60 ns each, 512 groups
// = 30-35 us per pass whatever the row count (a pass over 233 k rows took as long as one over 1.26 M).  Here a group STORES its 256
// bins as a row of `partial` and the pick sums the rows.
constexpr int TK1_GROUPS = 256;        // x 1024 threads; 512: the pick's sum over the groups' rows costs more than the pass gains
constexpr int TK1_THREADS = 1024;
__global__ __launch_bounds__(TK1_THREADS) void topk_hist1_kernel(const float* __restrict__ logits, int ld, int64_t n, int pass,
                                                         const int32_t* __restrict__ st, int32_t* __restrict__ partial) {
    __shared__ int lh[256];
    if (st[1] != 0) return;                       // resolved (or k <= 0)
    if (threadIdx.x < 256) lh[threadIdx.x] = 0;
    __syncthreads();
    // four rows per thread and iteration, their loads issued together
    const uint32_t pmask = top_bytes_mask(pass, 0), pref = (uint32_t)st[3] & pmask;
    const int shift = 8 * (3 - pass);
    const int64_t stride = (int64_t)gridDim.x * TK1_THREADS;
    for (int64_t i0 = (int64_t)blockIdx.x * TK1_THREADS + threadIdx.x; i0 < n; i0 += 4 * stride) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            v[u] = (i < n) ? logits[i * ld] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t kw = float_key(v[u]);
            const bool on = (i0 + u * stride < n) && ((kw & pmask) == pref);
            const uint32_t digit = (kw >> shift) & 0xFFu;
            const uint64_t same = match_digit(digit, on);        // (see topk_hist: one LDS atomic per distinct digit and wave)
            if (on && (same & ((1ull << (threadIdx.x & 63)) - 1ull)) == 0ull) atomicAdd(&lh[digit], __popcll(same));
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) partial[(int64_t)blockIdx.x * 256 + threadIdx.x] = lh[threadIdx.x];
}

// topk_pick for one batch item on the groups' bins: 1024 threads, thread (q, d) sums bin d over a quarter of the rows of `partial`
__global__ __launch_bounds__(1024) void topk_pick1_kernel(int pass, int32_t* __restrict__ st, const int32_t* __restrict__ partial, int groups) {
    __shared__ int hq[4][256];
    __shared__ int h[256];
    __shared__ int total_s;
    if (st[1] != 0) return;
    const int d = threadIdx.x & 255, q = threadIdx.x >> 8;
    int sum = 0;
    for (int g = q; g < groups; g += 4) sum += partial[(int64_t)g * 256 + d];
    hq[q][d] = sum;
    __syncthreads();
    if (q == 0) h[d] = hq[0][d] + hq[1][d] + hq[2][d] + hq[3][d];
    __syncthreads();
    const int krem = st[0];
    int above = 0;
    if (q == 0) {
        for (int j = d + 1; j < 256; ++j) above += h[j];
        if (d == 0) total_s = above + h[0];
    }
    __syncthreads();
    if (q != 0) return;
    if (pass == 0 && total_s <= krem) {
        if (d == 0) { st[1] = 1; st[2] = 0; }      // no more than k rows: keep them all
        return;
    }
    if (above < krem && krem <= above + h[d]) {
        st[3] = (int32_t)((uint32_t)st[3] | ((uint32_t)d << (8 * (3 - pass))));
        const int knew = krem - above;
        st[0] = knew;
        if (h[d] == knew) { st[1] = 1; st[2] = pass + 1; }
    }
}

// One batch item, passes 4-11 (the eight bytes of the tie-break key) in ONE workgroup: they run only when rows with EXACTLY the
// boundary logit straddle the k-th place; otherwise this launch returns at once — it replaces sixteen dispatches that did.  The
// slow case walks all rows eight times at one CU's rate (a frame of equal logits: milliseconds instead of microseconds, same mask).
__global__ __launch_bounds__(1024) void topk_tail_kernel(const float* __restrict__ logits, int ld, const int32_t* __restrict__ coords,
                                                         int64_t n, int32_t* __restrict__ state) {
    __shared__ int32_t st[8];
    __shared__ int h[256];
    const int t = threadIdx.x;
    if (state[1] != 0) return;
    if (t < 8) st[t] = state[t];
    __syncthreads();
    for (int pass = 4; pass < 12; ++pass) {
        if (st[1]) break;                              // uniform: st is only written between barriers
        if (t < 256) h[t] = 0;
        __syncthreads();
        for (int64_t i = t; i < n; i += 1024) {
            // the logit alone decides for almost every row: the coordinates are read for the rows on the boundary only
            if (float_key(logits[i * ld]) != (uint32_t)st[3]) continue;
            const int4 c = reinterpret_cast<const int4*>(coords)[i];
            if (c.x != 0) continue;
            const Key96 key = make_key(logits[i * ld], c);
            if (cmp_prefix(key, st, pass) != 0) continue;
            atomicAdd(&h[(key.w[pass >> 2] >> (8 * (3 - (pass & 3)))) & 0xFFu], 1);
        }
        __syncthreads();
        const int krem = st[0];
        int above = 0, mine = 0;
        if (t < 256) {
            mine = h[t];
            for (int j = t + 1; j < 256; ++j) above += h[j];
        }
        __syncthreads();
        if (t < 256 && above < krem && krem <= above + mine) {
            const int word = pass >> 2, shift = 8 * (3 - (pass & 3));
            st[3 + word] = (int32_t)((uint32_t)st[3 + word] | ((uint32_t)t << shift));
            const int knew = krem - above;
            st[0] = knew;
            if (mine == knew || pass == 11) { st[1] = 1; st[2] = pass + 1; }
        }
        __syncthreads();
    }
    if (t < 8) state[t] = st[t];
}

__global__ __launch_bounds__(256) void topk_pick(int pass, int32_t* __restrict__ state) {
    __shared__ int h[256];
    __shared__ int total_s;
    int32_t* st = state + (int64_t)blockIdx.x * TK_STRIDE;
    const int d = threadIdx.x;
    h[d] = st[8 + d];
    __syncthreads();
    // every thread reads the batch state before anyone rewrites it (writes come after the barrier)
    const int done = st[1];
    const int krem = st[0];
    int above = 0;
    for (int j = d + 1; j < 256; ++j) above += h[j];
    if (d == 0) total_s = above + h[0];
    __syncthreads();
    st[8 + d] = 0;
    if (done) return;
    if (pass == 0 && total_s <= krem) {
        // the batch has no more than k rows: keep them all (compare zero prefix bytes)
        if (d == 0) { st[1] = 1; st[2] = 0; }
        return;
    }
    // exactly one digit d satisfies this (its bucket holds the k-th largest key)
    if (above < krem && krem <= above + h[d]) {
        const int word = pass >> 2, shift = 8 * (3 - (pass & 3));
        st[3 + word] = (int32_t)((uint32_t)st[3 + word] | ((uint32_t)d << shift));
        const int knew = krem - above;
        st[0] = knew;
        if (h[d] == knew || pass == 11) { st[1] = 1; st[2] = pass + 1; }
    }
}

__global__ __launch_bounds__(256) void topk_write_mask(const float* __restrict__ logits, int ld,
                                                       const int32_t* __restrict__ coords, int64_t n, int nbatch,
                                                       const int32_t* __restrict__ state, uint8_t* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const int b = c.x;
    uint8_t m = 0;
    if (b >= 0 && b < nbatch) {
        const int32_t* st = state + (int64_t)b * TK_STRIDE;
        const int nb = st[2];
        if (nb <= 12) m = cmp_prefix(make_key(logits[i * ld], c), st, nb) >= 0 ? 1 : 0;
    }
    mask[i] = m;
}

// One batch item, at most TK_SMALL_N rows: init, the twelve (histogram, pick) rounds and the mask in ONE workgroup — 26
// launches otherwise, each a few microseconds of work behind its dispatch (a frame of a few thousand points runs three such
// selections: 72 of its launches).  The same radix select on the same 96-bit keys, the state kept in LDS: the same mask.
constexpr int TK_SMALL_N = 32768;
__global__ __launch_bounds__(1024) void topk_small_kernel(const float* __restrict__ logits, int ld, const int32_t* __restrict__ coords, int n,
                                                          const int32_t* __restrict__ k, uint8_t* __restrict__ mask,
                                                          int32_t* __restrict__ state) {
    __shared__ int32_t st[8];
    __shared__ int h[256];
    __shared__ int total_s;
    const int t = threadIdx.x;
    if (t < 8) st[t] = 0;
    __syncthreads();
    if (t == 0) {
        const int kk = k[0];
        st[0] = kk;
        if (kk <= 0) { st[1] = 1; st[2] = 13; }       // select nothing
    }
    __syncthreads();
    for (int pass = 0; pass < 12; ++pass) {
        if (st[1]) break;                              // uniform: st is only written between barriers
        if (t < 256) h[t] = 0;
        __syncthreads();
        const bool logit_only = pass < 4;              // a pass over the logit bytes: the tie-break key is not needed
        for (int i = t; i < n; i += 1024) {
            const int4 c = logit_only ? make_int4(0, 0, 0, 0) : reinterpret_cast<const int4*>(coords)[i];
            if (c.x != 0) continue;
            const Key96 key = make_key(logits[(int64_t)i * ld], c);
            if (cmp_prefix(key, st, pass) != 0) continue;
            atomicAdd(&h[(key.w[pass >> 2] >> (8 * (3 - (pass & 3)))) & 0xFFu], 1);
        }
        __syncthreads();
        // topk_pick on the LDS state: every thread reads before anyone writes
        const int krem = st[0];
        int above = 0, mine = 0;
        if (t < 256) {
            mine = h[t];
            for (int j = t + 1; j < 256; ++j) above += h[j];
            if (t == 0) total_s = above + mine;
        }
        __syncthreads();
        if (pass == 0 && total_s <= krem) {
            if (t == 0) { st[1] = 1; st[2] = 0; }      // no more than k rows: keep them all
        } else if (t < 256 && above < krem && krem <= above + mine) {
            const int word = pass >> 2, shift = 8 * (3 - (pass & 3));
            st[3 + word] = (int32_t)((uint32_t)st[3 + word] | ((uint32_t)t << shift));
            const int knew = krem - above;
            st[0] = knew;
            if (mine == knew || pass == 11) { st[1] = 1; st[2] = pass + 1; }
        }
        __syncthreads();
    }
    const int nb = st[2];
    for (int i = t; i < n; i += 1024) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        uint8_t m = 0;
        if (c.x == 0 && nb <= 12) m = cmp_prefix(make_key(logits[(int64_t)i * ld], c), st, nb) >= 0 ? 1 : 0;
        mask[i] = m;
    }
    for (int i = t; i < TK_STRIDE; i += 1024) state[i] = i < 8 ? st[i] : 0;      // as the separate launches leave it
}

__global__ __launch_bounds__(256) void coords_to_keys(const int32_t* __restrict__ coords, int64_t n,
                                                      uint64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    keys[i] = pack_key(c.x, c.y, c.z, c.w);
}

// ---- execution order of a kernel map's output rows -------------------------------------------
// Rows are sorted by their 27-bit neighbour mask read with the RAREST offset as the most significant bit, descending: rows
// with equal masks become adjacent, and where masks are too diverse for that (the sparse sets an untrained decoder keeps:
// tens of thousands of distinct masks) the rows of a 32-row MFMA tile at least agree on the rare offsets, which are the
// ones a tile should not have to execute for a single row.  Measured on surface subsets (offline, tools/order_experiment.py):
// issued / useful MFMA rows 2.27 -> 1.80 at 6.8 neighbours per row, 1.13 -> 1.08 on a full surface, against the previous
// key (27 - popcount) << 27 | mask; candidate sets (437 distinct masks) are at 1.00 either way.  Descending = rows holding
// the rare (corner) offsets, the heavy ones, first: tiles are dispatched in key order, so a launch ends on its cheap tiles.
// With spatial blocks the block id goes on top (64-bit key).
// rows per offset.  One ballot per offset and 64 rows (the count of an offset over a wave's rows is the popcount of the
// ballot: a scalar add) instead of one shift-and-add per offset and ROW — 27 x 64 vector operations per 64 rows before.
__global__ __launch_bounds__(256) void mask_bit_counts_kernel(const uint32_t* __restrict__ row_mask, int64_t n,
                                                              uint32_t* __restrict__ counts) {
    __shared__ unsigned c[27];
    if (threadIdx.x < 27) c[threadIdx.x] = 0u;
    __syncthreads();
    unsigned mine[27];                                   // wave-uniform
#pragma unroll
    for (int b = 0; b < 27; ++b) mine[b] = 0u;
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + threadIdx.x;
        const uint32_t m = (i < n) ? row_mask[i] : 0u;
#pragma unroll
        for (int b = 0; b < 27; ++b) mine[b] += (unsigned)__popcll(__ballot((m >> b) & 1u));
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int b = 0; b < 27; ++b)
            if (mine[b]) atomicAdd(&c[b], mine[b]);
    }
    __syncthreads();
    if (threadIdx.x < 27 && c[threadIdx.x]) atomicAdd(&counts[threadIdx.x], c[threadIdx.x]);
}

// bit position of every offset in the sort key: the rarest offset (ties: the lower offset index) takes bit 26
__device__ __forceinline__ void order_bit_positions(const uint32_t* __restrict__ counts, int* pos_s) {
    if (threadIdx.x < 27) {
        const uint32_t mine = counts[threadIdx.x];
        int rank = 0;
        for (int b = 0; b < 27; ++b) {
            const uint32_t o = counts[b];
            rank += (o < mine || (o == mine && b < (int)threadIdx.x)) ? 1 : 0;
        }
        pos_s[threadIdx.x] = 26 - rank;
    }
    __syncthreads();
}

__device__ __forceinline__ uint32_t order_key_of(uint32_t m, const int* pos_s) {
    uint32_t key = 0u;
#pragma unroll
    for (int b = 0; b < 27; ++b) key |= ((m >> b) & 1u) << pos_s[b];
    return 0x7FFFFFFu - key;                         // descending in the permuted mask
}

__global__ __launch_bounds__(256) void order_keys32_kernel(const uint32_t* __restrict__ row_mask, int64_t n,
                                                           const uint32_t* __restrict__ counts, uint32_t* __restrict__ keys) {
    __shared__ int pos_s[27];
    order_bit_positions(counts, pos_s);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t m = row_mask[i] & 0x7FFFFFFu;
    keys[i] = order_key_of(m, pos_s);
}

__global__ __launch_bounds__(256) void order_keys64_kernel(const uint32_t* __restrict__ row_mask,
                                                           const int32_t* __restrict__ coords, int64_t n,
                                                           int block_log2, int ts, const uint32_t* __restrict__ counts,
                                                           uint64_t* __restrict__ keys) {
    __shared__ int pos_s[27];
    order_bit_positions(counts, pos_s);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t key = order_key_of(row_mask[i] & 0x7FFFFFFu, pos_s);
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const uint64_t bx = (uint64_t)(((c.y / ts) + 512) >> block_log2) & 0x3FF;
    const uint64_t by = (uint64_t)(((c.z / ts) + 512) >> block_log2) & 0x3FF;
    const uint64_t bz = (uint64_t)(((c.w / ts) + 512) >> block_log2) & 0x3FF;
    key |= ((((uint64_t)(c.x & 0x3) << 30) | (bx << 20) | (by << 10) | bz) << 32);
    keys[i] = key;
}

// One OR-mask per 32 execution positions (the offsets a 32-row MFMA tile has to execute).  The neighbour table itself stays
// in output-row order: the convolution kernels read row order[pos] of it (csrc/conv.hip) — until round 4 this kernel also
// wrote a second, permuted copy of the table (557 MB on the 5.16 M-row candidate set of a config-2 frame, read once).
__global__ __launch_bounds__(256) void group_masks_kernel(const int32_t* __restrict__ order, const uint32_t* __restrict__ row_mask,
                                                          int64_t n, uint32_t* __restrict__ group_mask32) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t m = (i < n) ? row_mask[order[i]] : 0u;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) m |= __shfl_xor(m, d, 64);
    if (i < n && (threadIdx.x & 31) == 0) group_mask32[i >> 5] = m;
}

// out[pos] = table[order[pos]] (rows of K ints): the permuted copy of a neighbour table that the weight-gradient kernels of
// the training path index by execution position (csrc/conv_bwd.hip).  A block owns 256 consecutive positions; the rows are
// contiguous in the source and the destination is written fully coalesced.
__global__ __launch_bounds__(256) void permute_rows_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ nbr,
                                                           int64_t n, int K, int32_t* __restrict__ nbr_sorted) {
    __shared__ int32_t src_row[256];
    const int64_t r0 = (int64_t)blockIdx.x * 256;
    const int64_t i = r0 + threadIdx.x;
    src_row[threadIdx.x] = (i < n) ? order[i] : 0;
    __syncthreads();
    const int rows = (int)((n - r0 < 256) ? (n - r0) : 256);
    const int total = rows * K;
    int32_t* dst = nbr_sorted + r0 * K;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int lr = e / K;
        const int k = e - lr * K;
        dst[e] = nbr[(int64_t)src_row[lr] * K + k];
    }
}


// Offset counts, keys and the whole sort of a map of at most RS_SMALL_N (16,384) rows in ONE workgroup: what the ordering of
// such a map ran as a memset and three launches (mask_bit_counts_kernel, order_keys32_kernel, radix_sort_small_kernel),
// each a few microseconds of work behind its dispatch.  Same counts, same key, same sort: the same order.
template <int ROUNDS>
__global__ __launch_bounds__(RS_SMALL_THREADS) void order_small_kernel(const uint32_t* __restrict__ row_mask, int n, uint32_t* keys_a,
                                                                      uint32_t* keys_b, int32_t* va, int32_t* vb) {
    __shared__ unsigned cnt27[27];
    __shared__ int pos_s[27];
    const int t = threadIdx.x;
    if (t < 27) cnt27[t] = 0u;
    __syncthreads();
    unsigned mine[27];
#pragma unroll
    for (int b = 0; b < 27; ++b) mine[b] = 0u;
    for (int i = t; i < n; i += RS_SMALL_THREADS) {
        const uint32_t m = row_mask[i];
#pragma unroll
        for (int b = 0; b < 27; ++b) mine[b] += (m >> b) & 1u;
    }
#pragma unroll
    for (int b = 0; b < 27; ++b) {
        unsigned v = mine[b];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if ((t & 63) == 0 && v) atomicAdd(&cnt27[b], v);
    }
    __syncthreads();
    order_bit_positions(cnt27, pos_s);
    for (int i = t; i < n; i += RS_SMALL_THREADS) keys_a[i] = order_key_of(row_mask[i] & 0x7FFFFFFu, pos_s);
    __syncthreads();
    radix_sort_small_body<uint32_t, ROUNDS, 9>(keys_a, keys_b, va, vb, 1, n, 0, 27, 3);      // 27 bits = three 9-bit passes: the result ends in the b-side
}

// ---- a small kernel map and its execution order in ONE launch -----------------------------------------------------------
// Up to SMALL_MAP_MAX output rows: one 1024-thread workgroup runs what pcc_kernel_map + pcc_order_rows_by_mask run as
// several launches (probe, row masks, offset counts, keys, the one-workgroup radix sort, group masks).
// At these sizes every one of those launches is a few microseconds of work behind ~5 us of dispatch.  Same outputs bit for
// bit: the same probe, the same key (rarest offset first), the same stable sort.  Measured (tools/small_map_bench.py, one
// map, host calls included): 56 rows 22 us against 26 for kernel_map + order_small_kernel + order_apply, 300 rows 36
// against 36, 512 rows 66 against 46 — and 433 against 77 at 4,096: the probes (27 per row, ~150 vector instructions
// each) are then ONE CU's work, 360 of the 433 us; hence the 256-row limit.
constexpr int SMALL_MAP_MAX = 256;

// (kernel size as a template parameter: with a runtime K the index arithmetic of a probe — e / K, k % ks, three modulos by the
// parent pitch — is ~250 vector instructions, and ONE CU issues all of them: 190 us of a 4,096-row map)
template <int ROUNDS, int KS>
__global__ __launch_bounds__(RS_SMALL_THREADS) void small_map_kernel(const int32_t* __restrict__ out_coords, int n,
                                                                    const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                                                    uint64_t tmask, int tshift, int step, int pitch,
                                                                    int32_t* nbr, uint32_t* __restrict__ row_mask, int32_t* order,
                                                                    uint32_t* __restrict__ group_mask32, uint32_t* keys_a, uint32_t* keys_b,
                                                                    int32_t* vals_x) {
    constexpr int ks = KS, K = KS * KS * KS;
    __shared__ unsigned rm[SMALL_MAP_MAX];
    __shared__ unsigned cnt27[27];
    __shared__ int pos_s[27];
    const int t = threadIdx.x;
    const bool pow2 = pitch > 0 && (pitch & (pitch - 1)) == 0;
    auto off_grid = [&](int v) { return pow2 ? (v & (pitch - 1)) != 0 : (v % pitch) != 0; };
    for (int i = t; i < n; i += RS_SMALL_THREADS) rm[i] = 0u;
    if (t < 27) cnt27[t] = 0u;
    __syncthreads();
    // probes, (row, offset) pairs offset-fastest like kernel_map_kernel — eight per thread at a time: one workgroup has
    // 16 waves to hide the table's load latency with, so the loads of a batch are issued together (first slot of every
    // key, then the values of the hits; a key whose first slot holds another key — rare — takes the full search)
    const int total = n * K;
    constexpr int U = 8;
    for (int e0 = t; e0 < total; e0 += RS_SMALL_THREADS * U) {
        uint64_t key[U], slot[U], got[U];
        bool live[U];
        int lrs[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int e = e0 + j * RS_SMALL_THREADS;
            const bool valid = e < total;
            const int lr = valid ? e / K : 0, k = valid ? e - lr * K : 0;
            int dx, dy, dz;
            kernel_offset(ks, k, dx, dy, dz);
            const int4 c = reinterpret_cast<const int4*>(out_coords)[lr];
            const int x = c.y + dx * step, y = c.z + dy * step, z = c.w + dz * step;
            const bool on_grid = pitch <= 0 || !(off_grid(x) || off_grid(y) || off_grid(z));
            key[j] = pack_key(c.x, x, y, z);
            slot[j] = table_slot0(key[j], tmask, tshift);
            live[j] = valid && on_grid;
            lrs[j] = lr;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) got[j] = live[j] ? keys[slot[j]] : KEY_EMPTY;
        int val[U];
#pragma unroll
        for (int j = 0; j < U; ++j) val[j] = (live[j] && got[j] == key[j]) ? vals[slot[j]] : -1;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int e = e0 + j * RS_SMALL_THREADS;
            if (e >= total) continue;
            int idx = val[j];
            if (live[j] && got[j] != key[j] && got[j] != KEY_EMPTY) idx = table_find(keys, vals, tmask, tshift, key[j]);
            nbr[e] = idx;
            if (idx >= 0) atomicOr(&rm[lrs[j]], 1u << (e - lrs[j] * K));
        }
    }
    __syncthreads();
    // row masks out; rows per offset
    unsigned mine[27];
#pragma unroll
    for (int b = 0; b < 27; ++b) mine[b] = 0u;
    for (int i = t; i < n; i += RS_SMALL_THREADS) {
        const uint32_t m = rm[i];
        row_mask[i] = m;
#pragma unroll
        for (int b = 0; b < 27; ++b) mine[b] += (m >> b) & 1u;
    }
#pragma unroll
    for (int b = 0; b < 27; ++b) {
        unsigned v = mine[b];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if ((t & 63) == 0 && v) atomicAdd(&cnt27[b], v);
    }
    __syncthreads();
    order_bit_positions(cnt27, pos_s);
    for (int i = t; i < n; i += RS_SMALL_THREADS) keys_a[i] = order_key_of(rm[i] & 0x7FFFFFFu, pos_s);
    __syncthreads();
    // 27 key bits = three 9-bit passes: the sorted values (the rows, from an iota) end in the b-side = `order`
    radix_sort_small_body<uint32_t, ROUNDS, 9>(keys_a, keys_b, vals_x, order, 1, n, 0, 27, 3);
    // group masks (positions are consecutive across a wave's lanes)
    for (int i0 = 0; i0 < n; i0 += RS_SMALL_THREADS) {
        const int i = i0 + t;
        uint32_t m = (i < n) ? rm[order[i]] : 0u;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) m |= __shfl_xor(m, d, 64);
        if (i < n && (t & 31) == 0) group_mask32[i >> 5] = m;
    }
}

}  // namespace pcc

using namespace pcc;

extern "C" {

int64_t pcc_small_map_max(void) {
    static int off = -1;          // PCC_SMALL_MAP=0: always the separate launches (A/B)
    if (off < 0) { const char* e = getenv("PCC_SMALL_MAP"); off = (e && e[0] == '0') ? 1 : 0; }
    return off ? 0 : SMALL_MAP_MAX;
}

int pcc_small_kernel_map(const int32_t* out_coords, int64_t n_out, const uint64_t* in_keys, const int32_t* in_vals, int64_t in_cap,
                         int32_t ksize, int32_t step, int32_t sign, int32_t* nbr, uint32_t* row_mask, int32_t* order,
                         uint32_t* group_mask32, void* scratch, int64_t scratch_bytes, void* stream) {
    PCC_REQUIRE(ksize == 2 || ksize == 3, "pcc_small_kernel_map: kernel size must be 2 or 3");
    PCC_REQUIRE(sign == 1 || sign == -1, "pcc_small_kernel_map: sign must be +1/-1");
    PCC_REQUIRE(in_cap > 0 && (in_cap & (in_cap - 1)) == 0, "pcc_small_kernel_map: bad capacity");
    PCC_REQUIRE(step >= 1, "pcc_small_kernel_map: step must be >= 1");
    PCC_REQUIRE(n_out >= 0 && n_out <= SMALL_MAP_MAX, "pcc_small_kernel_map: %lld rows exceed %d", (long long)n_out, SMALL_MAP_MAX);
    PCC_REQUIRE(scratch_bytes >= pcc_sort_scratch_bytes(n_out), "pcc_small_kernel_map: scratch too small");
    if (n_out <= 0) return PCC_OK;
    const int in_stride = sign > 0 ? step : 2 * step;
    const int pitch = sign > 0 ? 0 : 2 * step;
    char* p = reinterpret_cast<char*>(scratch);
    uint32_t* keys_a = reinterpret_cast<uint32_t*>(p); p += align256(n_out * 8);
    uint32_t* keys_b = reinterpret_cast<uint32_t*>(p); p += align256(n_out * 8);
    int32_t* vals_x = reinterpret_cast<int32_t*>(p);
    const int rounds = (int)(((n_out + RS_SMALL_WAVES - 1) / RS_SMALL_WAVES + 63) / 64);
    hipStream_t st = as_stream(stream);
#define PCC_SMALL_MAP(R, S)                                                                                                          \
    hipLaunchKernelGGL((small_map_kernel<R, S>), dim3(1), dim3(RS_SMALL_THREADS), 0, st, out_coords, (int)n_out, in_keys, in_vals,    \
                       (uint64_t)(in_cap - 1), grid_shift_of(in_stride), sign * step, pitch, nbr, row_mask, order,                   \
                       group_mask32, keys_a, keys_b, vals_x)
#define PCC_SMALL_MAP_R(S)                                                                                                           \
    do {                                                                                                                             \
        if (rounds <= 1) PCC_SMALL_MAP(1, S);                                                                                        \
        else if (rounds <= 2) PCC_SMALL_MAP(2, S);                                                                                   \
        else PCC_SMALL_MAP(4, S);                                                                                                    \
    } while (0)
    if (ksize == 3) PCC_SMALL_MAP_R(3);
    else PCC_SMALL_MAP_R(2);
#undef PCC_SMALL_MAP_R
#undef PCC_SMALL_MAP
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int64_t pcc_topk_state_elems(int32_t nbatch) { return (int64_t)nbatch * TK_STRIDE + (nbatch == 1 ? (int64_t)TK1_GROUPS * 256 : 0); }

int pcc_topk_mask(const float* logits, int32_t ld, const int32_t* coords, int64_t n, int32_t nbatch, const int32_t* k,
                  uint8_t* mask, int32_t* state, void* stream) {
    PCC_REQUIRE(nbatch >= 1 && nbatch <= BATCH_LIMIT + 1, "pcc_topk_mask: bad nbatch %d", nbatch);
    PCC_REQUIRE(ld >= 1, "pcc_topk_mask: bad leading dimension");
    hipStream_t st = as_stream(stream);
    if (small_path_enabled(1) && nbatch == 1 && n > 0 && n <= TK_SMALL_N) {      // PCC_TOPK_SMALL=0: never (A/B)
        hipLaunchKernelGGL(topk_small_kernel, dim3(1), dim3(1024), 0, st, logits, ld, coords, (int)n, k, mask, state);
        PCC_LAUNCH_CHECK();
        return PCC_OK;
    }
    hipLaunchKernelGGL(topk_init, dim3(nbatch), dim3(256), 0, st, k, nbatch, state);
    if (n > 0) {
        // 512 workgroups, not more: every group ends with up to 256 atomic adds on the SAME global bins, which serialise
        // (one item, 5.16 M rows, four passes: 335 us with 2048 groups, 236 with 512, 303 with 256 — fewer groups read the rows slower)
        const unsigned nb = blocks_for(n, 256, 512);
        if (nbatch == 1 && small_path_enabled(1)) {            // PCC_TOPK_SMALL=0: the separate launches (A/B)
            const unsigned groups = blocks_for(n, TK1_THREADS * 4, TK1_GROUPS);
            int32_t* partial = state + TK_STRIDE;
            for (int pass = 0; pass < 4; ++pass) {
                hipLaunchKernelGGL(topk_hist1_kernel, dim3(groups), dim3(TK1_THREADS), 0, st, logits, ld, n, pass, state, partial);
                hipLaunchKernelGGL(topk_pick1_kernel, dim3(1), dim3(1024), 0, st, pass, state, partial, (int)groups);
            }
            hipLaunchKernelGGL(topk_tail_kernel, dim3(1), dim3(1024), 0, st, logits, ld, coords, n, state);
            hipLaunchKernelGGL(topk_write_mask, dim3(blocks_for(n, 256)), dim3(256), 0, st, logits, ld, coords, n, nbatch,
                               state, mask);
            PCC_LAUNCH_CHECK();
            return PCC_OK;
        }
        for (int pass = 0; pass < 12; ++pass) {
            hipLaunchKernelGGL(topk_hist, dim3(nb), dim3(256), 0, st, logits, ld, coords, n, nbatch, pass, state);
            hipLaunchKernelGGL(topk_pick, dim3(nbatch), dim3(256), 0, st, pass, state);
        }
        hipLaunchKernelGGL(topk_write_mask, dim3(blocks_for(n, 256)), dim3(256), 0, st, logits, ld, coords, n, nbatch,
                           state, mask);
    }
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int64_t pcc_order_scratch_bytes(int64_t n) { return pcc_sort_scratch_bytes(n); }

int pcc_order_rows_by_mask(const uint32_t* row_mask, const int32_t* coords, int64_t n, int32_t block_log2,
                           int32_t tensor_stride, int32_t* order, uint32_t* group_mask32, void* scratch, int64_t scratch_bytes,
                           void* stream) {
    if (n <= 0) return PCC_OK;
    PCC_REQUIRE(n < (1ll << 31), "pcc_order_rows_by_mask: too many rows");
    PCC_REQUIRE(block_log2 < 0 || coords != nullptr, "pcc_order_rows_by_mask: block ordering needs coordinates");
    PCC_REQUIRE(block_log2 <= 10, "pcc_order_rows_by_mask: block_log2 out of range");
    PCC_REQUIRE(tensor_stride >= 1, "pcc_order_rows_by_mask: bad tensor stride");
    PCC_REQUIRE(scratch_bytes >= pcc_sort_scratch_bytes(n), "pcc_order_rows_by_mask: scratch too small");
    hipStream_t st = as_stream(stream);
    char* p = reinterpret_cast<char*>(scratch);
    char* keys_a = p; p += align256(n * 8);
    char* keys_b = p; p += align256(n * 8);
    int32_t* vals_x = reinterpret_cast<int32_t*>(p); p += align256(n * 4);
    void* counters = p;
    uint32_t* bit_counts = reinterpret_cast<uint32_t*>(p + radix_sort_counter_bytes(n));       // 27 words in the scratch's 256-byte tail
    if (small_path_enabled(0) && block_log2 < 0 && n <= RS_SMALL_N) {      // PCC_ORDER_SMALL=0: never (A/B)
        // counts + keys + sort in one workgroup (27 key bits = three 9-bit passes: the sorted rows end in the b-side = `order`)
        const int rounds = (int)(((n + RS_SMALL_WAVES - 1) / RS_SMALL_WAVES + 63) / 64);
        uint32_t* ka = reinterpret_cast<uint32_t*>(keys_a);
        uint32_t* kb = reinterpret_cast<uint32_t*>(keys_b);
#define PCC_ORDER_SMALL(R) hipLaunchKernelGGL(order_small_kernel<R>, dim3(1), dim3(RS_SMALL_THREADS), 0, st, row_mask, (int)n, ka, kb, vals_x, order)
        if (rounds <= 1) PCC_ORDER_SMALL(1);
        else if (rounds <= 2) PCC_ORDER_SMALL(2);
        else if (rounds <= 4) PCC_ORDER_SMALL(4);
        else if (rounds <= 8) PCC_ORDER_SMALL(8);
        else PCC_ORDER_SMALL(16);
#undef PCC_ORDER_SMALL
        hipLaunchKernelGGL(group_masks_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, order, row_mask, n, group_mask32);
        PCC_LAUNCH_CHECK();
        return PCC_OK;
    }
    PCC_CHECK_HIP(hipMemsetAsync(bit_counts, 0, 27 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(mask_bit_counts_kernel, dim3(blocks_for(n, 256 * 16, 512)), dim3(256), 0, st, row_mask, n, bit_counts);
    const int begin = 0, end = block_log2 >= 0 ? 64 : 27;
    const int digit_bits = block_log2 >= 0 ? 8 : radix_sort_order_digit_bits(n);      // 27 bits: three 9-bit passes instead of four 8-bit ones
    // the sorted values must land in `order`: they end in the b-side when the pass count is odd
    const bool in_b = radix_sort_result_in_b(begin, end, digit_bits);
    int32_t* va = in_b ? vals_x : order;
    int32_t* vb = in_b ? order : vals_x;
    int rc;
    if (block_log2 < 0) {
        hipLaunchKernelGGL(order_keys32_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, row_mask, n, bit_counts,
                           reinterpret_cast<uint32_t*>(keys_a));
        rc = radix_sort_pairs_u32(reinterpret_cast<uint32_t*>(keys_a), reinterpret_cast<uint32_t*>(keys_b), va, vb, true, n, begin, end,
                                  counters, st, digit_bits);
    } else {
        hipLaunchKernelGGL(order_keys64_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, row_mask, coords, n, block_log2, tensor_stride,
                           bit_counts, reinterpret_cast<uint64_t*>(keys_a));
        rc = radix_sort_pairs_u64(reinterpret_cast<uint64_t*>(keys_a), reinterpret_cast<uint64_t*>(keys_b), va, vb, true, n, begin, end,
                                  counters, st);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(group_masks_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, order, row_mask, n, group_mask32);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_permute_map_rows(const int32_t* nbr, const int32_t* order, int64_t n, int32_t K, int32_t* nbr_sorted, void* stream) {
    PCC_REQUIRE(K >= 1 && K <= 27, "pcc_permute_map_rows: K out of range");
    PCC_REQUIRE(nbr != nullptr && order != nullptr && nbr_sorted != nullptr, "pcc_permute_map_rows: null argument");
    if (n <= 0) return PCC_OK;
    hipLaunchKernelGGL(permute_rows_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), order, nbr, n, K, nbr_sorted);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int64_t pcc_sort_scratch_bytes(int64_t n) {
    if (n < 1) n = 1;
    return 2 * align256(n * 8) + align256(n * 4) + radix_sort_counter_bytes(n) + 256;
}

int pcc_sort_coords(const int32_t* coords, int64_t n, int32_t* perm, void* scratch, int64_t scratch_bytes, void* stream) {
    if (n <= 0) return PCC_OK;
    PCC_REQUIRE(n < (1ll << 31), "pcc_sort_coords: too many rows");
    PCC_REQUIRE(scratch_bytes >= pcc_sort_scratch_bytes(n), "pcc_sort_coords: scratch too small");
    hipStream_t st = as_stream(stream);
    char* p = reinterpret_cast<char*>(scratch);
    uint64_t* keys_a = reinterpret_cast<uint64_t*>(p); p += align256(n * 8);
    uint64_t* keys_b = reinterpret_cast<uint64_t*>(p); p += align256(n * 8);
    int32_t* vals_x = reinterpret_cast<int32_t*>(p); p += align256(n * 4);
    void* counters = p;
    hipLaunchKernelGGL(coords_to_keys, dim3(blocks_for(n, 256)), dim3(256), 0, st, coords, n, keys_a);
    const bool in_b = radix_sort_result_in_b(0, 64);
    return radix_sort_pairs_u64(keys_a, keys_b, in_b ? vals_x : perm, in_b ? perm : vals_x, true, n, 0, 64, counters, st);
}

}  // extern "C"
