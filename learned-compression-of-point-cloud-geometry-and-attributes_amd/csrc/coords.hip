// Coordinate manager kernels: hashed-voxel table, stride map, generative children,
// kernel (neighbour) map, prefix scan, pruning, row gather/scatter.
//
// All of this is HBM-/atomic-bound integer work: the design
// rules that matter are coalesced candidate streams, one 64-bit CAS per insert, and keeping
// results independent of thread arrival order (the winner of a duplicate is the smallest
// candidate index, so outputs are bit-reproducible).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "sort.h"

namespace pcc {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

// ---------------------------------------------------------------------------------------------
// exclusive scan of int32 flags: 3 kernels, 1024 items per 256-thread block
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// returns exclusive prefix of v within the block; total in *block_total (all threads)
__device__ __forceinline__ int block_exclusive_scan(int v, int* block_total) {
    __shared__ int wave_sums[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int inc = wave_inclusive_scan(v, lane);
    if (lane == 63) wave_sums[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        const int s = wave_sums[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *block_total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_block_sums(const int32_t* __restrict__ flags, int64_t m,
                                                              int32_t* __restrict__ block_sums) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < m) v += flags[base + i];
    int tot;
    block_exclusive_scan(v, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// `err` (optional): a device word the producers of the flags set when the set cannot be built (unique_insert: a
// coordinate outside the key range); the count that goes to the host is then COUNT_ERR_RANGE instead of a row count,
// so the error travels with the one value the host reads anyway.
__global__ __launch_bounds__(SCAN_BLOCK) void scan_of_block_sums(int32_t* __restrict__ block_sums, int64_t nb,
                                                                 int64_t* __restrict__ total_out,
                                                                 const int32_t* __restrict__ err) {
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += SCAN_BLOCK) {
        const int64_t i = base + threadIdx.x;
        const int v = (i < nb) ? block_sums[i] : 0;
        int tot;
        const int ex = block_exclusive_scan(v, &tot);
        const int carry = carry_s;
        if (i < nb) block_sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) {
        *total_out = (err && *err) ? COUNT_ERR_RANGE : (int64_t)carry_s;
        __threadfence_system();          // the word may be page-locked host memory that the host polls
    }
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply(const int32_t* __restrict__ flags, int64_t m,
                                                         const int32_t* __restrict__ block_offsets,
                                                         int32_t* __restrict__ pos, int inclusive) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int f[SCAN_ITEMS];
    int v = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        f[i] = (base + i < m) ? flags[base + i] : 0;
        v += f[i];
    }
    int tot;
    int ex = block_exclusive_scan(v, &tot) + block_offsets[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < m) pos[base + i] = inclusive ? ex + f[i] : ex;
        ex += f[i];
    }
}

int64_t scan_block_sums_elems(int64_t m) { return (m + SCAN_TILE - 1) / SCAN_TILE + 16; }

// flags and pos may alias (in-place).  block_sums: scan_block_sums_elems(m) ints.
int scan_flags(const int32_t* flags, int64_t m, int32_t* pos, int32_t* block_sums, int64_t* total, int inclusive,
               hipStream_t st, const int32_t* err) {
    const int64_t nb = (m + SCAN_TILE - 1) / SCAN_TILE;
    if (m <= 0) {
        if (total) PCC_CHECK_HIP(hipMemsetAsync(total, 0, sizeof(int64_t), st));
        return PCC_OK;
    }
    hipLaunchKernelGGL(scan_block_sums, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, flags, m, block_sums);
    hipLaunchKernelGGL(scan_of_block_sums, dim3(1), dim3(SCAN_BLOCK), 0, st, block_sums, nb, total, err);
    hipLaunchKernelGGL(scan_apply, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, flags, m, block_sums, pos, inclusive);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

static int exclusive_scan(const int32_t* flags, int64_t m, int32_t* pos, int32_t* block_sums, int64_t* total,
                          hipStream_t st) {
    return scan_flags(flags, m, pos, block_sums, total, 0, st, nullptr);
}

// ---------------------------------------------------------------------------------------------
// hash table
// ---------------------------------------------------------------------------------------------
__global__ void table_clear(uint64_t* __restrict__ keys, int32_t* __restrict__ vals, int64_t cap, int32_t* __restrict__ err) {
    if (err && blockIdx.x == 0 && threadIdx.x == 0) *err = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = KEY_EMPTY;
        vals[i] = 0x7fffffff;
    }
}

// claim (or find) the slot of `key`; returns slot index.  The key's lane first (every 8th slot); a lane that is full of
// other keys — more than cap / 8 >= a quarter of the candidates in one lane: improbable, not impossible — hands the key
// to slot-by-slot probing from its first slot, which table_find mirrors.  With cap >= 2 * candidates a free slot exists,
// so the second loop always returns; mask + 1 is unreachable and callers still guard it.
__device__ __forceinline__ uint64_t table_claim(uint64_t* keys, uint64_t mask, int shift, uint64_t key) {
    const uint64_t slot0 = table_slot0(key, mask, shift);
    uint64_t slot = slot0;
    for (uint64_t probe = 0; probe <= mask; probe += TABLE_PROBE_STEP) {
        uint64_t cur = keys[slot];
        if (cur == KEY_EMPTY) {
            cur = (uint64_t)atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)KEY_EMPTY,
                                      (unsigned long long)key);
            if (cur == KEY_EMPTY) return slot;
        }
        if (cur == key) return slot;
        slot = (slot + TABLE_PROBE_STEP) & mask;
    }
    for (uint64_t probe = 1; probe <= mask; ++probe) {
        slot = (slot0 + probe) & mask;
        uint64_t cur = keys[slot];
        if (cur == KEY_EMPTY) {
            cur = (uint64_t)atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)KEY_EMPTY,
                                      (unsigned long long)key);
            if (cur == KEY_EMPTY) return slot;
        }
        if (cur == key) return slot;
    }
    return mask + 1;
}

// Candidate generators ------------------------------------------------------------------------
struct GenRows {  // candidate i = row i
    const int32_t* coords;
    __device__ __forceinline__ bool in_range(int64_t) const { return true; }
    __device__ __forceinline__ void get(int64_t i, int& b, int& x, int& y, int& z) const {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        b = c.x; x = c.y; y = c.z; z = c.w;
    }
};
struct GenStride {  // candidate i = floor(row i / 2ts) * 2ts
    const int32_t* coords;
    int s2;  // 2 * ts (power of two in practice, but do a true floor division)
    __device__ __forceinline__ bool in_range(int64_t i) const {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        return coord_in_range(c.x, c.y, c.z, c.w);
    }
    __device__ __forceinline__ int fl(int v) const {
        int q = v / s2;
        if ((v % s2) != 0 && v < 0) --q;
        return q * s2;
    }
    __device__ __forceinline__ void get(int64_t i, int& b, int& x, int& y, int& z) const {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        b = c.x; x = fl(c.y); y = fl(c.z); z = fl(c.w);
    }
};
struct GenChildren {  // ks=3: candidate i = (parent i / 27, offset i % 27); ks=2: (offset i / n, parent i % n)
    const int32_t* coords;
    int64_t n;
    int ks, half;
    __device__ __forceinline__ bool in_range(int64_t) const { return true; }     // a child is within half a stride of its parent
    __device__ __forceinline__ void get(int64_t i, int& b, int& x, int& y, int& z) const {
        int64_t p;
        int k;
        if (ks == 3) { p = i / 27; k = (int)(i - p * 27); }
        else { k = (int)(i / n); p = i - (int64_t)k * n; }
        const int4 c = reinterpret_cast<const int4*>(coords)[p];
        int dx, dy, dz;
        kernel_offset(ks, k, dx, dy, dz);
        b = c.x; x = c.y + dx * half; y = c.z + dy * half; z = c.w + dz * half;
    }
};

template <class Gen>
__global__ __launch_bounds__(256) void unique_insert(Gen gen, int64_t m, uint64_t* __restrict__ keys,
                                                     int32_t* __restrict__ vals, uint64_t mask, int shift,
                                                     int32_t* __restrict__ slot_of, int32_t* __restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int b, x, y, z;
    gen.get(i, b, x, y, z);
    // the source row's own coordinates are covered too: |floor(c / 2ts) 2ts| >= |c| - 2ts + 1 and a child is within ts of
    // its parent, so a source coordinate beyond the limit by more than a stride puts its candidate beyond it as well —
    // and GenStride / GenChildren check the source row directly (in_range)
    if (!coord_in_range(b, x, y, z) || !gen.in_range(i)) {
        *err = 1;
        slot_of[i] = (int32_t)(mask + 1);
        return;
    }
    const uint64_t slot = table_claim(keys, mask, shift, pack_key(b, x, y, z));
    slot_of[i] = (int32_t)slot;
    if (slot <= mask) atomicMin(&vals[slot], (int32_t)i);
}

__global__ __launch_bounds__(256) void unique_flag(int64_t m, const int32_t* __restrict__ vals, uint32_t mask,
                                                   const int32_t* __restrict__ slot_of, int32_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t slot = (uint32_t)slot_of[i];
    flags[i] = (slot <= mask && vals[slot] == (int32_t)i) ? 1 : 0;       // slot > mask: the candidate was rejected (range error)
}

// incl = inclusive scan of the winner flags: candidate i won its slot iff the scan steps at i, and its output row
// is incl[i] - 1.  The winner test reads the scan, never `vals`, so the slot can be rewritten to the row id at once.
template <class Gen>
__global__ __launch_bounds__(256) void unique_finalize(Gen gen, int64_t m, int32_t* __restrict__ vals,
                                                       const int32_t* __restrict__ slot_of,
                                                       const int32_t* __restrict__ incl,
                                                       int32_t* __restrict__ out_coords) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int32_t cur = incl[i], prev = i ? incl[i - 1] : 0;
    if (cur != prev) {
        const int32_t row = cur - 1;
        int b, x, y, z;
        gen.get(i, b, x, y, z);
        reinterpret_cast<int4*>(out_coords)[row] = make_int4(b, x, y, z);
        vals[slot_of[i]] = row;
    }
}

static int g_small_paths = -1;
static int small_paths_value() {
    if (g_small_paths < 0) {
        const char* names[3] = {"PCC_ORDER_SMALL", "PCC_TOPK_SMALL", "PCC_UNIQUE_SMALL"};
        int v = 0;
        for (int b = 0; b < 3; ++b) {
            const char* e = getenv(names[b]);
            if (!(e && e[0] == '0')) v |= 1 << b;
        }
        g_small_paths = v;
    }
    return g_small_paths;
}
bool small_path_enabled(int bit) { return (small_paths_value() >> bit) & 1; }

// A small candidate set (m <= UNIQUE_SMALL_M) in ONE workgroup: table clear, insert, winner flags, their scan, the output
// rows and the count — seven launches otherwise, each a few microseconds of work behind its dispatch, on the critical path
// of a host that waits for the count.  Same rule (the lowest candidate index wins its slot; output rows in candidate
// order): the same set, the same table.  Winner flags are scanned and finalised 1024 candidates at a time; a finalised slot
// holds a row id <= its winner's index < any later candidate's index, so a later duplicate still reads "not me".
constexpr int UNIQUE_SMALL_M = 8192;
template <class Gen>
__global__ __launch_bounds__(1024) void unique_small_kernel(Gen gen, int m, uint64_t* keys, int32_t* vals, int64_t cap, int shift,
                                                            int32_t* __restrict__ slot_of, int32_t* __restrict__ out_coords,
                                                            int64_t* __restrict__ out_count, int32_t* __restrict__ err) {
    __shared__ int err_s, carry_s;
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const uint64_t mask = (uint64_t)(cap - 1);
    for (int64_t i = t; i < cap; i += 1024) {
        keys[i] = KEY_EMPTY;
        vals[i] = 0x7fffffff;
    }
    if (t == 0) { err_s = 0; carry_s = 0; }
    __syncthreads();
    for (int i = t; i < m; i += 1024) {
        int b, x, y, z;
        gen.get(i, b, x, y, z);
        if (!coord_in_range(b, x, y, z) || !gen.in_range(i)) {
            err_s = 1;
            slot_of[i] = (int32_t)(mask + 1);
            continue;
        }
        const uint64_t slot = table_claim(keys, mask, shift, pack_key(b, x, y, z));
        slot_of[i] = (int32_t)slot;
        if (slot <= mask) atomicMin(&vals[slot], (int32_t)i);
    }
    __syncthreads();
    for (int i0 = 0; i0 < m; i0 += 1024) {
        const int i = i0 + t;
        uint32_t slot = 0;
        int flag = 0;
        if (i < m) {
            slot = (uint32_t)slot_of[i];
            // (an agent-scope load: the winners were written by atomics at the L2, behind this CU's vector cache, which no
            // kernel boundary has invalidated in between)
            flag = (slot <= mask && __hip_atomic_load(&vals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int32_t)i) ? 1 : 0;
        }
        const int inc = wave_inclusive_scan(flag, lane);
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        int base = carry_s, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int v = wsum[w];
            if (w < wid) base += v;
            tot += v;
        }
        if (flag) {
            const int32_t row = base + inc - 1;
            int b, x, y, z;
            gen.get(i, b, x, y, z);
            reinterpret_cast<int4*>(out_coords)[row] = make_int4(b, x, y, z);
            vals[slot] = row;
        }
        __syncthreads();
        if (t == 0) carry_s += tot;
        __syncthreads();
    }
    if (t == 0) {
        if (err) *err = err_s;
        *out_count = err_s ? COUNT_ERR_RANGE : (int64_t)carry_s;
        __threadfence_system();          // the word may be page-locked host memory that the host polls
    }
}

template <class Gen>
static int unique_coords(Gen gen, int64_t m, uint64_t* keys, int32_t* vals, int64_t cap, int shift, int32_t* scratch,
                         int32_t* out_coords, int64_t* out_count, hipStream_t st) {
    PCC_REQUIRE(cap > 0 && (cap & (cap - 1)) == 0, "hash capacity %lld is not a power of two", (long long)cap);
    PCC_REQUIRE(cap >= 2 * m, "hash capacity %lld too small for %lld candidates", (long long)cap, (long long)m);
    PCC_REQUIRE(m < (1ll << 31) - 1, "too many candidates (%lld)", (long long)m);
    PCC_REQUIRE(cap <= (1ll << 31), "hash capacity %lld exceeds 2^31 slots", (long long)cap);
    int32_t* slot_of = scratch;
    int32_t* flags = scratch + (m > 0 ? m : 0);
    int32_t* block_sums = scratch + 2 * (m > 0 ? m : 0);
    // the error word lives in the 16 spare ints behind the scan's block sums (pcc_scan_scratch_elems); table_clear zeroes it
    int32_t* err = block_sums + (m > 0 ? (m + SCAN_TILE - 1) / SCAN_TILE : 0) + 8;
    if (small_path_enabled(2) && m > 0 && m <= UNIQUE_SMALL_M && cap <= 8 * UNIQUE_SMALL_M) {      // PCC_UNIQUE_SMALL=0: never (A/B)
        hipLaunchKernelGGL(unique_small_kernel<Gen>, dim3(1), dim3(1024), 0, st, gen, (int)m, keys, vals, cap, shift, slot_of, out_coords,
                           out_count, err);
        PCC_LAUNCH_CHECK();
        return PCC_OK;
    }
    hipLaunchKernelGGL(table_clear, dim3(blocks_for(cap, 256, 4096)), dim3(256), 0, st, keys, vals, cap, err);
    if (m <= 0) {
        PCC_CHECK_HIP(hipMemsetAsync(out_count, 0, sizeof(int64_t), st));
        return PCC_OK;
    }
    const unsigned nb = blocks_for(m, 256);
    hipLaunchKernelGGL(unique_insert<Gen>, dim3(nb), dim3(256), 0, st, gen, m, keys, vals, (uint64_t)(cap - 1), shift, slot_of, err);
    hipLaunchKernelGGL(unique_flag, dim3(nb), dim3(256), 0, st, m, vals, (uint32_t)(cap - 1), slot_of, flags);
    int rc = scan_flags(flags, m, flags, block_sums, out_count, 1, st, err);
    if (rc) return rc;
    hipLaunchKernelGGL(unique_finalize<Gen>, dim3(nb), dim3(256), 0, st, gen, m, vals, slot_of, flags, out_coords);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_insert(const int32_t* __restrict__ coords, int64_t n,
                                                    uint64_t* __restrict__ keys, int32_t* __restrict__ vals,
                                                    uint64_t mask, int shift) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const uint64_t slot = table_claim(keys, mask, shift, pack_key(c.x, c.y, c.z, c.w));
    if (slot <= mask) atomicMin(&vals[slot], (int32_t)i);
}

__global__ __launch_bounds__(256) void build_count_dups(const int32_t* __restrict__ coords, int64_t n,
                                                        const uint64_t* __restrict__ keys,
                                                        const int32_t* __restrict__ vals, uint64_t mask, int shift,
                                                        int32_t* __restrict__ dup_count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    if (table_find(keys, vals, mask, shift, pack_key(c.x, c.y, c.z, c.w)) != (int32_t)i) atomicAdd(dup_count, 1);
}

__global__ __launch_bounds__(256) void lookup_kernel(const uint64_t* __restrict__ keys,
                                                     const int32_t* __restrict__ vals, uint64_t mask, int shift,
                                                     const int32_t* __restrict__ query, int64_t nq,
                                                     int32_t* __restrict__ out_idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int4 c = reinterpret_cast<const int4*>(query)[i];
    out_idx[i] = table_find(keys, vals, mask, shift, pack_key(c.x, c.y, c.z, c.w));
}

// One 256-thread block per group of 64 output rows: 64*K probes, coalesced nbr stores, row masks reduced in LDS.
// Lanes walk (row, offset) pairs offset-fastest, so the three dz probes of a (dx, dy) column sit in one wave
// instruction and — the table keeps a z-run in one line (table_slot0) — are served by one line request.
// Transposed maps (parent = c - d * step must lie on the parent grid of pitch 2 * step) skip the offsets whose
// parity rules the parent out: 1 + (c mod 2 step != 0) candidates per axis instead of 3 (k = 3), 1 instead of 2
// (k = 2) — 3.4 probes per row on average instead of 27.
__global__ __launch_bounds__(256) void kernel_map_kernel(const int32_t* __restrict__ out_coords, int64_t n_out,
                                                         const uint64_t* __restrict__ keys,
                                                         const int32_t* __restrict__ vals, uint64_t mask, int shift,
                                                         int ks, int K, int step, int parent_pitch,
                                                         int32_t* __restrict__ nbr, uint32_t* __restrict__ row_mask,
                                                         unsigned long long* __restrict__ pair_count) {
    __shared__ int4 rows[64];
    __shared__ unsigned rm[64];
    __shared__ unsigned hits_s;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    if (threadIdx.x < 64) {
        const int64_t r = row0 + threadIdx.x;
        rows[threadIdx.x] = (r < n_out) ? reinterpret_cast<const int4*>(out_coords)[r] : make_int4(0, 0, 0, 0);
        rm[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) hits_s = 0u;
    __syncthreads();
    unsigned myhits = 0u;
    const int total = 64 * K;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int lr = e / K, k = e - lr * K;
        const int64_t r = row0 + lr;
        if (r >= n_out) break;
        int dx, dy, dz;
        kernel_offset(ks, k, dx, dy, dz);
        const int4 c = rows[lr];
        const int x = c.y + dx * step, y = c.z + dy * step, z = c.w + dz * step;
        int idx = -1;
        // parent_pitch > 0 (transposed): off-grid targets cannot exist in the parent set — no probe
        const bool on_grid = parent_pitch <= 0 || (((x % parent_pitch) | (y % parent_pitch) | (z % parent_pitch)) == 0);
        if (on_grid) idx = table_find(keys, vals, mask, shift, pack_key(c.x, x, y, z));
        nbr[r * K + k] = idx;
        if (idx >= 0) { atomicOr(&rm[lr], 1u << k); ++myhits; }
    }
    if (myhits) atomicAdd(&hits_s, myhits);
    __syncthreads();
    if (row_mask && threadIdx.x < 64 && row0 + threadIdx.x < n_out) row_mask[row0 + threadIdx.x] = rm[threadIdx.x];
    if (threadIdx.x == 0 && pair_count && !row_mask && hits_s) atomicAdd(pair_count, (unsigned long long)hits_s);
}

// Kernel size 3 (every large map of the codec), same lane mapping as the generic kernel below — (row, offset) pairs
// offset-fastest — with everything the generic kernel computes per probe hoisted out: K = 27 is a compile-time
// constant (e / 27 is a multiply), the 27 offsets come from a table in LDS, and the parity test of transposed maps
// is a mask.  The generic kernel spends ~200 instructions per probe on e / K, k % ks and runtime-pitch modulos.
template <bool POW2>
__global__ __launch_bounds__(256) void kernel_map27_kernel(const int32_t* __restrict__ out_coords, int64_t n_out,
                                                           const uint64_t* __restrict__ keys,
                                                           const int32_t* __restrict__ vals, uint64_t mask, int shift,
                                                           int step, int parent_pitch, int32_t* __restrict__ nbr,
                                                           uint32_t* __restrict__ row_mask,
                                                           unsigned long long* __restrict__ pair_count) {
    // A neighbour's key is the row's key plus a constant: every field of a voxel key stays inside its 18 bits under a step of
    // one stride (COORD_LIMIT's margin, common.h), so the 64-bit sum of the row's key and the offset's packed delta IS
    // pack_key(b, x + dx step, y + dy step, z + dz step) — one add per probe instead of three coordinate adds and a pack.
    __shared__ uint64_t rowkey[64];
    __shared__ uint64_t dkey[27];
    __shared__ unsigned rm[64];
    __shared__ unsigned hits_s;
    __shared__ unsigned short queue[64 * 27];      // pair slots whose first table slot holds another key (phase B)
    __shared__ int queue_n;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    if (threadIdx.x < 64) {
        const int64_t r = row0 + threadIdx.x;
        const int4 c = (r < n_out) ? reinterpret_cast<const int4*>(out_coords)[r] : make_int4(0, 0, 0, 0);
        rowkey[threadIdx.x] = pack_key(c.x, c.y, c.z, c.w);
        rm[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 255) queue_n = 0;
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 27) {
        const int k = threadIdx.x - 64;
        const int64_t dx = (int64_t)(k % 3 - 1) * step, dy = (int64_t)((k / 3) % 3 - 1) * step, dz = (int64_t)(k / 9 - 1) * step;
        dkey[k] = (uint64_t)(dx * (1ll << KEY_X_SHIFT) + dy * (1ll << KEY_Y_SHIFT) + dz);
    }
    if (threadIdx.x == 0) hits_s = 0u;
    __syncthreads();
    const int nrows = (int)((n_out - row0 < 64) ? (n_out - row0) : 64);
    const int total = nrows * 27;
    unsigned myhits = 0u;
    // transposed maps: the parent must lie on the grid of pitch 2 step; COORD_BIAS is a multiple of every power-of-two pitch,
    // so the test reads the biased fields of the key directly
    auto off_grid_key = [&](uint64_t key) {
        if (POW2) {
            // (pitches are <= 2^10: the low bits of the three fields are read straight off the shifted key)
            const uint32_t m = (uint32_t)(parent_pitch - 1);
            return ((((uint32_t)key | (uint32_t)(key >> KEY_Y_SHIFT) | (uint32_t)(key >> KEY_X_SHIFT)) & m) != 0u);
        }
        const int z = (int)key_z(key) - COORD_BIAS, y = (int)key_y(key) - COORD_BIAS, x = (int)key_x(key) - COORD_BIAS;
        return (x % parent_pitch) != 0 || (y % parent_pitch) != 0 || (z % parent_pitch) != 0;
    };
    // Two phases (round 4; profiles/r04_kernel_map_counters.txt).  The one-probe-at-a-time loop this replaces left every lane with
    // a single load in flight: a wave lived 18.5 us and issued 34 loads one after the other, each waiting for the slowest of its
    // ~25 lines (0.5 us: one of them misses the L2 nearly always), 68 % of all wave cycles spent waiting; the same map TRANSPOSED
    // — an eighth of the probes — took the same 1.6 ms on 11.9 M rows.  Batching alone (round 3, and again in round 4) was
    // SLOWER: one probe in five finds another key in its first slot, so in every wave SOME lane needs the lane walk, and the
    // wave pays that dependent chain once per batched probe.
    //   A: the first slot's key and value of all (up to) seven probes of a thread are loaded together; a match or an empty slot
    //      settles the probe, anything else goes to a queue in LDS;
    //   B: the workgroup's queued probes (~ a fifth) are taken up DENSELY, a lane each, with the next four slots of the lane walk
    //      (keys and values) loaded together; what is still open after five slots (~0.5 %) finishes with the sequential search.
    // The table walk is table_find's, slot for slot, so the map is the same.  Measured (tools/coord_bench.py, 11.9 M rows): same-set
    // map 1.74 -> 1.47 ms, transposed 1.58 -> 1.17 ms.  What is left is vector-ALU issue, not memory: ~670 vector instructions
    // per wave (hash with four quarter-rate 32-bit multiplies, 64-bit keys and addresses, the exec-mask bookkeeping of seven
    // unrolled probes) x 4 cycles per wave64 instruction on a 16-lane SIMD x 726 waves per SIMD = 0.93 ms of the 1.17; dropping
    // every value load (an ablation build) took only another 10 % off.  The next step would share one hash between the three dz
    // probes of a (dx, dy) column — a lane per column, results transposed through LDS; not built.
    constexpr int U = 7;                                     // 64 rows x 27 offsets = 1,728 pair slots = 6.75 per thread
    {
        uint64_t key[U], got[U];
        uint32_t slot[U];
        int val[U];
        bool live[U];
        // pair slot e = threadIdx.x + 256 j = 27 lr + k: from one j to the next lr advances by 9 and k by 13 (256 = 9 x 27 + 13)
        int lr = threadIdx.x / 27, k = threadIdx.x - lr * 27;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int e = threadIdx.x + 256 * j;
            const bool valid = e < total;
            key[j] = rowkey[valid ? lr : 0] + dkey[k];
            live[j] = valid && (parent_pitch <= 0 || !off_grid_key(key[j]));
            slot[j] = (uint32_t)table_slot0(key[j], mask, shift);
            lr += 9;
            k += 13;
            if (k >= 27) { k -= 27; ++lr; }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            got[j] = live[j] ? keys[slot[j]] : KEY_EMPTY;
            val[j] = live[j] ? vals[slot[j]] : -1;
        }
        lr = threadIdx.x / 27;
        k = threadIdx.x - lr * 27;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int e = threadIdx.x + 256 * j;
            const int lr_j = lr, k_j = k;
            lr += 9;
            k += 13;
            if (k >= 27) { k -= 27; ++lr; }
            if (e >= total) continue;
            int idx = -1;
            if (live[j] && got[j] != KEY_EMPTY) {
                if (got[j] == key[j]) idx = val[j];
                else {                                       // another key's: the lane walk, in phase B
                    queue[atomicAdd(&queue_n, 1)] = (unsigned short)e;
                    continue;
                }
            }
            nbr[row0 * 27 + e] = idx;
            if (idx >= 0) { atomicOr(&rm[lr_j], 1u << k_j); ++myhits; }
        }
    }
    __syncthreads();
    const int qn = queue_n;
    for (int qi = threadIdx.x; qi < qn; qi += 256) {
        const int e = queue[qi];
        const int lr = e / 27, k = e - lr * 27;
        const uint64_t key = rowkey[lr] + dkey[k];
        const uint64_t slot0 = table_slot0(key, mask, shift);
        uint64_t kk[4];
        int vv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t sl = (slot0 + (uint64_t)(i + 1) * TABLE_PROBE_STEP) & mask;
            kk[i] = keys[sl];
            vv[i] = vals[sl];
        }
        int idx = -1;
        bool open = true;
        const int lane_slots = (int)((mask + TABLE_PROBE_STEP) / TABLE_PROBE_STEP);      // slots of a lane: the walk ends there
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!open) break;
            if (i + 1 >= lane_slots) break;                   // (tables of fewer than 40 slots: leave it to the search below)
            if (kk[i] == key) { idx = vv[i]; open = false; }
            else if (kk[i] == KEY_EMPTY) { idx = -1; open = false; }
        }
        if (open) idx = table_find(keys, vals, mask, shift, key);     // cold: five slots of the lane taken by other keys
        nbr[row0 * 27 + e] = idx;
        if (idx >= 0) { atomicOr(&rm[lr], 1u << k); ++myhits; }
    }
    if (myhits) atomicAdd(&hits_s, myhits);
    __syncthreads();
    if (row_mask && threadIdx.x < nrows) row_mask[row0 + threadIdx.x] = rm[threadIdx.x];
    if (threadIdx.x == 0 && pair_count && !row_mask && hits_s) atomicAdd(pair_count, (unsigned long long)hits_s);
}

// pairs = sum of popcount(row_mask).  A separate grid-stride reduction with a few hundred workgroups: one atomic per
// workgroup of the MAP kernel on the single counter serialises at the L2 — 80 k same-address atomics were 1.0 of the
// 1.35 ms of the 5.16 M-row map (a transposed map with an eighth of the probes took the same time).
__global__ __launch_bounds__(256) void pair_count_kernel(const uint32_t* __restrict__ row_mask, int64_t n,
                                                         unsigned long long* __restrict__ pair_count) {
    __shared__ unsigned long long ws[4];
    unsigned long long v = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) v += __popc(row_mask[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = ws[0] + ws[1] + ws[2] + ws[3];
        if (t) atomicAdd(pair_count, t);
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int c,
                                                          const int32_t* __restrict__ idx, int64_t n,
                                                          float* __restrict__ out, int accumulate) {
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / c;
        const int col = (int)(e - r * c);
        const int32_t s = idx[r];
        const float v = (s >= 0) ? src[(int64_t)s * c + col] : 0.0f;
        out[e] = accumulate ? out[e] + v : v;
    }
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, int c,
                                                           const int32_t* __restrict__ idx, int64_t n,
                                                           float* __restrict__ out) {
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / c;
        const int col = (int)(e - r * c);
        const int32_t d = idx[r];
        if (d >= 0) out[(int64_t)d * c + col] = src[e];
    }
}

// out[idx[r], :] += src[r, :]: the backward of a row gather (training path).  Float atomics: the sum order is
// unspecified only where several source rows hit the same target (repeated queries).
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, int c,
                                                               const int32_t* __restrict__ idx, int64_t n,
                                                               float* __restrict__ out) {
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / c;
        const int col = (int)(e - r * c);
        const int32_t d = idx[r];
        if (d >= 0) atomicAdd(&out[(int64_t)d * c + col], src[e]);
    }
}

__global__ __launch_bounds__(256) void mask_to_flags(const uint8_t* __restrict__ mask, int64_t n,
                                                     int32_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = mask[i] ? 1 : 0;
}

__global__ __launch_bounds__(256) void compact_index_kernel(const uint8_t* __restrict__ mask,
                                                            const int32_t* __restrict__ pos, int64_t n,
                                                            const int32_t* __restrict__ coords,
                                                            int32_t* __restrict__ out_coords,
                                                            int32_t* __restrict__ new_index) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool keep = mask[i] != 0;
    const int32_t p = pos[i];
    if (new_index) new_index[i] = keep ? p : -1;
    if (keep && coords) reinterpret_cast<int4*>(out_coords)[p] = reinterpret_cast<const int4*>(coords)[i];
}

// kept rows move as 16-byte pieces: one thread per (row, 4 channels); dropped rows cost one mask byte per piece
__global__ __launch_bounds__(256) void compact_feats_kernel(const uint8_t* __restrict__ mask,
                                                            const int32_t* __restrict__ pos, int64_t n, int c,
                                                            const float* __restrict__ feats,
                                                            float* __restrict__ out_feats, int vec) {
    if (vec) {
        const int c4 = c >> 2;
        const int64_t total = n * c4;
        const float4* src = reinterpret_cast<const float4*>(feats);
        float4* dst = reinterpret_cast<float4*>(out_feats);
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
            const int64_t r = e / c4;
            if (mask[r]) dst[(int64_t)pos[r] * c4 + (e - r * c4)] = src[e];
        }
        return;
    }
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / c;
        if (mask[r]) out_feats[(int64_t)pos[r] * c + (e - r * c)] = feats[e];
    }
}

// rows per batch item.  Batch indices are few and rows of one item are contiguous in practice, so a
// global atomic per row serialises on a handful of addresses (4 ms for 5 M rows); count runs inside the
// wave first: one atomic per (wave, batch value present in the wave).
__global__ __launch_bounds__(256) void count_batch_kernel(const int32_t* __restrict__ coords, int64_t n, int nbatch,
                                                          int32_t* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int b = (i < n) ? coords[i * 4] : -1;
    if (b >= nbatch) b = -1;
    while (true) {
        const unsigned long long todo = __ballot(b >= 0);
        if (!todo) break;
        const int leader = __ffsll(todo) - 1;
        const int v = __shfl(b, leader);
        const unsigned long long same = __ballot(b == v);
        if ((threadIdx.x & 63) == leader) atomicAdd(&counts[v], __popcll(same));
        if (b == v) b = -1;
    }
}

}  // namespace pcc

using namespace pcc;

extern "C" {

int pcc_version(void) { return 1; }
const char* pcc_last_error(void) { return pcc::last_error(); }

int pcc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pcc_device_name(int dev, char* out, int out_len) {
    hipDeviceProp_t p;
    PCC_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    snprintf(out, (size_t)out_len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return PCC_OK;
}

int64_t pcc_hash_capacity(int64_t n) {
    int64_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    return cap;
}

int64_t pcc_scan_scratch_elems(int64_t m) { return 2 * m + (m + SCAN_TILE - 1) / SCAN_TILE + 16; }

int pcc_hash_build(const int32_t* coords, int64_t n, uint64_t* keys, int32_t* vals, int64_t cap, int32_t tensor_stride,
                   int32_t* dup_count, void* stream) {
    hipStream_t st = as_stream(stream);
    PCC_REQUIRE(tensor_stride >= 1, "pcc_hash_build: tensor stride must be >= 1");
    const int shift = grid_shift_of(tensor_stride);
    PCC_REQUIRE(cap > 0 && (cap & (cap - 1)) == 0 && cap >= 2 * n, "pcc_hash_build: bad capacity %lld for n=%lld",
                (long long)cap, (long long)n);
    hipLaunchKernelGGL(table_clear, dim3(blocks_for(cap, 256, 4096)), dim3(256), 0, st, keys, vals, cap, (int32_t*)nullptr);
    if (dup_count) PCC_CHECK_HIP(hipMemsetAsync(dup_count, 0, sizeof(int32_t), st));
    if (n > 0) {
        hipLaunchKernelGGL(build_insert, dim3(blocks_for(n, 256)), dim3(256), 0, st, coords, n, keys, vals,
                           (uint64_t)(cap - 1), shift);
        if (dup_count)
            hipLaunchKernelGGL(build_count_dups, dim3(blocks_for(n, 256)), dim3(256), 0, st, coords, n, keys, vals,
                               (uint64_t)(cap - 1), shift, dup_count);
    }
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_hash_lookup(const uint64_t* keys, const int32_t* vals, int64_t cap, int32_t tensor_stride, const int32_t* query,
                    int64_t nq, int32_t* out_idx, void* stream) {
    PCC_REQUIRE(cap > 0 && (cap & (cap - 1)) == 0, "pcc_hash_lookup: bad capacity");
    PCC_REQUIRE(tensor_stride >= 1, "pcc_hash_lookup: tensor stride must be >= 1");
    if (nq <= 0) return PCC_OK;
    hipLaunchKernelGGL(lookup_kernel, dim3(blocks_for(nq, 256)), dim3(256), 0, as_stream(stream), keys, vals,
                       (uint64_t)(cap - 1), grid_shift_of(tensor_stride), query, nq, out_idx);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_stride_map(const int32_t* coords, int64_t n, int32_t ts, uint64_t* keys, int32_t* vals, int64_t cap,
                   int32_t* scratch, int32_t* out_coords, int64_t* out_count, void* stream) {
    PCC_REQUIRE(ts >= 1, "pcc_stride_map: tensor stride must be >= 1");
    GenStride gen{coords, 2 * ts};
    return unique_coords(gen, n, keys, vals, cap, grid_shift_of(2 * ts), scratch, out_coords, out_count, as_stream(stream));
}

int pcc_children(const int32_t* coords, int64_t n, int32_t ts, int32_t ksize, uint64_t* keys, int32_t* vals,
                 int64_t cap, int32_t* scratch, int32_t* out_coords, int64_t* out_count, void* stream) {
    PCC_REQUIRE(ksize == 2 || ksize == 3, "pcc_children: kernel size must be 2 or 3");
    PCC_REQUIRE(ts >= 2 && (ts % 2) == 0, "pcc_children: tensor stride %d is not even", ts);
    const int K = ksize * ksize * ksize;
    GenChildren gen{coords, n, ksize, ts / 2};
    return unique_coords(gen, n * K, keys, vals, cap, grid_shift_of(ts / 2), scratch, out_coords, out_count, as_stream(stream));
}

int32_t pcc_small_paths(int32_t mask) {
    const int before = pcc::small_paths_value();
    if (mask >= 0) pcc::g_small_paths = mask & 7;
    return before;
}

int pcc_kernel_map(const int32_t* out_coords, int64_t n_out, const uint64_t* in_keys, const int32_t* in_vals,
                   int64_t in_cap, int32_t ksize, int32_t step, int32_t sign, int32_t* nbr, uint32_t* row_mask,
                   int64_t* pair_count, void* stream) {
    PCC_REQUIRE(ksize >= 1 && ksize <= 3, "pcc_kernel_map: kernel size must be 1..3");
    PCC_REQUIRE(sign == 1 || sign == -1, "pcc_kernel_map: sign must be +1/-1");
    PCC_REQUIRE(in_cap > 0 && (in_cap & (in_cap - 1)) == 0, "pcc_kernel_map: bad capacity");
    if (pair_count) PCC_CHECK_HIP(hipMemsetAsync(pair_count, 0, sizeof(int64_t), as_stream(stream)));
    if (n_out <= 0) return PCC_OK;
    const int K = ksize * ksize * ksize;
    PCC_REQUIRE(step >= 1, "pcc_kernel_map: step must be >= 1");
    // the input set's grid: pitch `step` for a (strided) convolution, 2 * step for a transposed one
    const int in_stride = sign > 0 ? step : 2 * step;
    const int pitch = sign > 0 ? 0 : 2 * step;
    const bool pow2 = pitch == 0 || (pitch & (pitch - 1)) == 0;
    unsigned long long* pc = reinterpret_cast<unsigned long long*>(pair_count);
    const uint64_t tmask = (uint64_t)(in_cap - 1);
    const int tshift = grid_shift_of(in_stride);
    const unsigned nb = blocks_for(n_out, 64);
    hipStream_t st = as_stream(stream);
    if (ksize == 3) {      // offset-fastest lanes, K = 27 at compile time; kernel sizes 1 and 2 take the generic kernel
        if (pow2) hipLaunchKernelGGL(kernel_map27_kernel<true>, dim3(nb), dim3(256), 0, st, out_coords, n_out, in_keys, in_vals, tmask, tshift,
                                     sign * step, pitch, nbr, row_mask, pc);
        else hipLaunchKernelGGL(kernel_map27_kernel<false>, dim3(nb), dim3(256), 0, st, out_coords, n_out, in_keys, in_vals, tmask, tshift,
                                sign * step, pitch, nbr, row_mask, pc);
    } else {
        hipLaunchKernelGGL(kernel_map_kernel, dim3(nb), dim3(256), 0, st, out_coords, n_out, in_keys, in_vals, tmask, tshift, ksize, K,
                           sign * step, pitch, nbr, row_mask, pc);
    }
    if (pair_count && row_mask)
        hipLaunchKernelGGL(pair_count_kernel, dim3(blocks_for(n_out, 256 * 16, 512)), dim3(256), 0, st, row_mask, n_out, pc);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_pair_count(const uint32_t* row_mask, int64_t n_out, int64_t* pair_count, void* stream) {
    PCC_REQUIRE(pair_count != nullptr, "pcc_pair_count: no output");
    hipStream_t st = as_stream(stream);
    PCC_CHECK_HIP(hipMemsetAsync(pair_count, 0, sizeof(int64_t), st));
    if (n_out <= 0) return PCC_OK;
    hipLaunchKernelGGL(pair_count_kernel, dim3(blocks_for(n_out, 256 * 16, 512)), dim3(256), 0, st, row_mask, n_out,
                       reinterpret_cast<unsigned long long*>(pair_count));
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_gather_rows(const float* src, int32_t c, const int32_t* idx, int64_t n, float* out, int32_t accumulate,
                    void* stream) {
    if (n <= 0 || c <= 0) return PCC_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_for(n * c, 256, 65536)), dim3(256), 0, as_stream(stream), src, c,
                       idx, n, out, accumulate);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_scatter_rows(const float* src, int32_t c, const int32_t* idx, int64_t n, float* out, void* stream) {
    if (n <= 0 || c <= 0) return PCC_OK;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(blocks_for(n * c, 256, 65536)), dim3(256), 0, as_stream(stream), src,
                       c, idx, n, out);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_scatter_add_rows(const float* src, int32_t c, const int32_t* idx, int64_t n, float* out, void* stream) {
    if (n <= 0 || c <= 0) return PCC_OK;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(blocks_for(n * c, 256, 65536)), dim3(256), 0, as_stream(stream), src,
                       c, idx, n, out);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_compact_rows(const uint8_t* mask, int64_t n, const int32_t* coords, int32_t* out_coords, const float* feats,
                     int32_t c, float* out_feats, int32_t* new_index, int32_t* scratch, int64_t* out_count,
                     void* stream) {
    hipStream_t st = as_stream(stream);
    if (n <= 0) {
        PCC_CHECK_HIP(hipMemsetAsync(out_count, 0, sizeof(int64_t), st));
        return PCC_OK;
    }
    int32_t* flags = scratch;
    int32_t* block_sums = scratch + n;
    hipLaunchKernelGGL(mask_to_flags, dim3(blocks_for(n, 256)), dim3(256), 0, st, mask, n, flags);
    int rc = exclusive_scan(flags, n, flags, block_sums, out_count, st);
    if (rc) return rc;
    if (coords || new_index)
        hipLaunchKernelGGL(compact_index_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, mask, flags, n, coords,
                           out_coords, new_index);
    if (feats && c > 0)
    {
        const int vec = (c & 3) == 0 && ((reinterpret_cast<uintptr_t>(feats) | reinterpret_cast<uintptr_t>(out_feats)) & 15) == 0;
        hipLaunchKernelGGL(compact_feats_kernel, dim3(blocks_for(vec ? n * (c / 4) : n * c, 256, 65536)), dim3(256), 0, st,
                           mask, flags, n, c, feats, out_feats, vec);
    }
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_count_per_batch(const int32_t* coords, int64_t n, int32_t nbatch, int32_t* counts, void* stream) {
    hipStream_t st = as_stream(stream);
    PCC_REQUIRE(nbatch >= 1, "pcc_count_per_batch: nbatch must be >= 1");
    PCC_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * nbatch, st));
    if (n > 0) hipLaunchKernelGGL(count_batch_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, coords, n, nbatch, counts);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

}  // extern "C"
