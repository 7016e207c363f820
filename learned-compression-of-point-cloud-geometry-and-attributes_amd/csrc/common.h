// Shared helpers for libpcc_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcc_hip.h"

namespace pcc {

void set_error(const char* fmt, ...);

#define PCC_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            pcc::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return PCC_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define PCC_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            pcc::set_error(__VA_ARGS__);            \
            return PCC_ERR_ARG;                     \
        }                                           \
    } while (0)

#define PCC_LAUNCH_CHECK() PCC_CHECK_HIP(hipGetLastError())

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline unsigned blocks_for(int64_t n, int per_block, unsigned cap = 0x7fffffffu) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (int64_t)cap) b = cap;
    return (unsigned)b;
}

// ---- voxel key: (b << 54) | (x+2^17) << 36 | (y+2^17) << 18 | (z+2^17): 10 bits of batch index, 18 bits per coordinate.
// Ascending key order == lexicographic (b, x, y, z): the reference's canonical order (utils.py:170-171).  Valid for
// |coord| <= COORD_LIMIT and 0 <= b <= BATCH_LIMIT — the margin below 2^17 keeps every neighbour probe (coordinate +- one step
// of a tensor stride <= 512) inside the field, so a key never wraps onto another voxel's.  The range covers the reference's
// radix-1e5 keys (model/blocks.py:118, utils.py:170: coordinates 0 .. 99,999) and as much again on the negative side (rounds
// 1-3 had 16-bit fields: |coord| <= 32,000); a coordinate outside it is an ERROR reported with the row count of the next
// coordinate-set construction (coords.hip: unique_insert -> COUNT_ERR_RANGE), never an aliased key.
constexpr uint64_t KEY_EMPTY = 0xFFFFFFFFFFFFFFFFull;
constexpr int KEY_FIELD_BITS = 18;
constexpr uint32_t KEY_FIELD_MASK = (1u << KEY_FIELD_BITS) - 1u;
constexpr int KEY_Y_SHIFT = KEY_FIELD_BITS, KEY_X_SHIFT = 2 * KEY_FIELD_BITS, KEY_B_SHIFT = 3 * KEY_FIELD_BITS;
constexpr int COORD_BIAS = 1 << (KEY_FIELD_BITS - 1);      // a multiple of every power-of-two grid pitch (kernel_map27's on-grid test)
constexpr int COORD_LIMIT = 130000;
constexpr int BATCH_LIMIT = 1022;
constexpr int64_t COUNT_ERR_RANGE = -2;      // written instead of a row count (PCC_COUNT_ERR_RANGE in pcc_hip.h)

__host__ __device__ __forceinline__ bool coord_in_range(int b, int x, int y, int z) {
    return (unsigned)b <= (unsigned)BATCH_LIMIT && (unsigned)(x + COORD_LIMIT) <= 2u * COORD_LIMIT &&
           (unsigned)(y + COORD_LIMIT) <= 2u * COORD_LIMIT && (unsigned)(z + COORD_LIMIT) <= 2u * COORD_LIMIT;
}

__host__ __device__ __forceinline__ uint64_t pack_key(int b, int x, int y, int z) {
    return ((uint64_t)((uint32_t)b & 0x3FFu) << KEY_B_SHIFT) | ((uint64_t)((uint32_t)(x + COORD_BIAS) & KEY_FIELD_MASK) << KEY_X_SHIFT) |
           ((uint64_t)((uint32_t)(y + COORD_BIAS) & KEY_FIELD_MASK) << KEY_Y_SHIFT) | (uint64_t)((uint32_t)(z + COORD_BIAS) & KEY_FIELD_MASK);
}

// the three biased coordinate fields of a key
__host__ __device__ __forceinline__ uint32_t key_z(uint64_t key) { return (uint32_t)key & KEY_FIELD_MASK; }
__host__ __device__ __forceinline__ uint32_t key_y(uint64_t key) { return (uint32_t)(key >> KEY_Y_SHIFT) & KEY_FIELD_MASK; }
__host__ __device__ __forceinline__ uint32_t key_x(uint64_t key) { return (uint32_t)(key >> KEY_X_SHIFT) & KEY_FIELD_MASK; }

__device__ __forceinline__ uint64_t hash_key(uint64_t k) {
    // splitmix64 finaliser
    k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
    k ^= k >> 27; k *= 0x94d049bb133111ebull;
    k ^= k >> 31;
    return k;
}

// Slot function of the hashed-voxel table.  The 8 voxels of an aligned run along z (8 grid steps of the set's
// tensor stride, `shift` = log2(stride)) share the hash of their run and differ in the low 3 slot bits, so they
// sit in ONE 64-byte line of `keys` (and one 32-byte sector of `vals`): the 3 dz probes of a kernel offset
// column, and the probes of neighbouring rows, hit the same lines instead of 27 random ones.  Probing advances
// by 8 slots — each of the 8 lanes is an ordinary linear-probing table over runs, so dense runs do not lengthen
// the unsuccessful probes the way slot-by-slot probing through a spatial block does.
__device__ __forceinline__ uint64_t table_slot0(uint64_t key, uint64_t mask, int shift) {
    const uint32_t z = key_z(key);
    // 32-bit mix of the run (the key without the 3 + shift low bits of z: bits of the z-run, y and x in `lo`, the rest of y, x
    // and b in `hi`): the probe kernels are bound by VALU issue as much as by memory, and a 64-bit splitmix is ~55 issue slots
    // on CDNA (four quarter-rate 32-bit multiplies per 64-bit product) against ~25 here
    const uint32_t lo = (uint32_t)(key >> (shift + 3)), hi = (uint32_t)(key >> 32);
    uint32_t h = lo * 0x9E3779B1u + hi * 0x85EBCA77u;
    h ^= h >> 15; h *= 0x2C1B3C6Du;
    h ^= h >> 12; h *= 0x297A2D39u;
    h ^= h >> 15;
    // the run's lanes are rotated by 3 hash bits the bucket index does not use (cap <= 2^31 slots = 2^28 buckets): a
    // degenerate set (a plane of constant z, a wrong `shift`) still spreads evenly over the 8 lanes
    return (((uint64_t)h << 3) | (((z >> shift) + (h >> 29)) & 7u)) & mask;
}
constexpr uint64_t TABLE_PROBE_STEP = 8;

__host__ __device__ __forceinline__ int grid_shift_of(int tensor_stride) {      // log2 for powers of two, else 0
    int s = 0;
    if (tensor_stride > 0 && (tensor_stride & (tensor_stride - 1)) == 0)
        while ((1 << s) < tensor_stride) ++s;
    return s;
}

// Probe the table.  Returns row id or -1.  A key's lane (every 8th slot from its first one) is searched first; only if that
// whole lane holds other keys — which needs more than a quarter of a set's voxels in one lane: adversarial input — the
// search goes on slot by slot from the first slot, exactly as table_claim (coords.hip) placed the key.  Lanes only ever
// fill, so "the lane is full" reads the same for the insert and for every later lookup.
__device__ __forceinline__ int table_find(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                          uint64_t mask, int shift, uint64_t key) {
    const uint64_t slot0 = table_slot0(key, mask, shift);
    uint64_t slot = slot0;
    for (uint64_t probe = 0; probe <= mask; probe += TABLE_PROBE_STEP) {
        const uint64_t k = keys[slot];
        if (k == key) return vals[slot];
        if (k == KEY_EMPTY) return -1;
        slot = (slot + TABLE_PROBE_STEP) & mask;
    }
    for (uint64_t probe = 1; probe <= mask; ++probe) {          // cold: the key's lane is full
        slot = (slot0 + probe) & mask;
        const uint64_t k = keys[slot];
        if (k == key) return vals[slot];
        if (k == KEY_EMPTY) return -1;
    }
    return -1;
}

// Kernel offset of index k for kernel size ks (x fastest; ks=3 centred, ks=2 un-centred).
__device__ __forceinline__ void kernel_offset(int ks, int k, int& dx, int& dy, int& dz) {
    if (ks == 1) { dx = dy = dz = 0; return; }
    const int ix = k % ks, iy = (k / ks) % ks, iz = k / (ks * ks);
    const int c = (ks == 3) ? 1 : 0;
    dx = ix - c; dy = iy - c; dz = iz - c;
}

// One-workgroup forms of the small per-map chains (bit 0: execution order <= 16,384 rows, bit 1: top-k <= 32,768 rows, bit 2:
// coordinate sets <= 8,192 candidates): on unless PCC_ORDER_SMALL=0 / PCC_TOPK_SMALL=0 / PCC_UNIQUE_SMALL=0, or switched at run
// time through pcc_small_paths (coords.hip; A/B measurements in one process).
bool small_path_enabled(int bit);

}  // namespace pcc
