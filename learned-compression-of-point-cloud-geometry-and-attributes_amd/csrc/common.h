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

// ---- voxel key: (b << 48) | (x+2^15) << 32 | (y+2^15) << 16 | (z+2^15).
// Ascending key order == lexicographic (b, x, y, z): the reference's canonical order
// (utils.py:170-171).  Valid for |coord| < 32767, 0 <= b < 32767.
constexpr uint64_t KEY_EMPTY = 0xFFFFFFFFFFFFFFFFull;
constexpr int COORD_BIAS = 1 << 15;

__host__ __device__ __forceinline__ uint64_t pack_key(int b, int x, int y, int z) {
    return ((uint64_t)(uint16_t)b << 48) | ((uint64_t)(uint16_t)(x + COORD_BIAS) << 32) |
           ((uint64_t)(uint16_t)(y + COORD_BIAS) << 16) | (uint64_t)(uint16_t)(z + COORD_BIAS);
}

__device__ __forceinline__ uint64_t hash_key(uint64_t k) {
    // splitmix64 finaliser
    k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
    k ^= k >> 27; k *= 0x94d049bb133111ebull;
    k ^= k >> 31;
    return k;
}

// Probe an open-addressing table (linear probing).  Returns row id or -1.
__device__ __forceinline__ int table_find(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                          uint64_t mask, uint64_t key) {
    uint64_t slot = hash_key(key) & mask;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        const uint64_t k = keys[slot];
        if (k == key) return vals[slot];
        if (k == KEY_EMPTY) return -1;
        slot = (slot + 1) & mask;
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

}  // namespace pcc
