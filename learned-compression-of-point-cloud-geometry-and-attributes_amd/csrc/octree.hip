// Lossless octree serialisation of a coordinate list (the stride-8 latent coordinates of file mode).
//
// Stands where the reference shells out to MPEG G-PCC `tmc3` (model/model.py:318-395, gpcc_encode /
// gpcc_decode).  tmc3 is an external binary that is not part of the reference tree; this is the
// build's own coder ("PCO1", see ../octree.py for the container; tests/ check the kernels against a
// CPU twin).  It is NOT G-PCC compatible.
//
// Device side = everything that scales with the number of points: Morton keys, radix sort, one
// occupancy byte per occupied node and level (encode); level-by-level expansion of the occupancy
// bytes back to coordinates (decode).  The bytes then go through the same host range coder as the
// latents (rans_host.cpp).  All of it is HBM/latency-bound integer work: 8 B key + 1 B output per
// node and level.
#include "common.h"
#include "sort.h"

namespace pcc {

__device__ __forceinline__ uint64_t spread3(uint32_t v) {       // bit b -> bit 3 b (21 bits)
    uint64_t x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__device__ __forceinline__ uint32_t compact3(uint64_t x) {     // inverse of spread3
    x &= 0x1249249249249249ull;
    x = (x | (x >> 2)) & 0x10c30c30c30c30c3ull;
    x = (x | (x >> 4)) & 0x100f00f00f00f00full;
    x = (x | (x >> 8)) & 0x1f0000ff0000ffull;
    x = (x | (x >> 16)) & 0x1f00000000ffffull;
    x = (x | (x >> 32)) & 0x1fffffull;
    return (uint32_t)x;
}

// child index at every level = (xbit << 2) | (ybit << 1) | zbit
__global__ __launch_bounds__(256) void octree_keys_kernel(const int32_t* __restrict__ coords, int64_t n, int stride, int ox, int oy,
                                                          int oz, int depth, uint64_t* __restrict__ keys, int32_t* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int rx = coords[4 * i + 1] - ox, ry = coords[4 * i + 2] - oy, rz = coords[4 * i + 3] - oz;
    const int gx = rx / stride, gy = ry / stride, gz = rz / stride;
    const int lim = 1 << depth;
    const bool ok = rx >= 0 && ry >= 0 && rz >= 0 && gx * stride == rx && gy * stride == ry && gz * stride == rz && gx < lim &&
                    gy < lim && gz < lim;
    if (!ok) atomicAdd(bad, 1);
    keys[i] = ok ? ((spread3((uint32_t)gx) << 2) | (spread3((uint32_t)gy) << 1) | spread3((uint32_t)gz)) : 0ull;
}

// head[i] = 1 when the depth-L prefix of sorted key i differs from its predecessor's
__global__ __launch_bounds__(256) void octree_heads_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift, int32_t* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t p = shift >= 64 ? 0ull : keys[i] >> shift;
    const uint64_t q = i == 0 ? ~p : (shift >= 64 ? 0ull : keys[i - 1] >> shift);
    head[i] = p != q;
}

__global__ __launch_bounds__(256) void octree_or_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ rank, int64_t n,
                                                        int child_shift, uint32_t* __restrict__ words, int32_t* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    atomicOr(&words[rank[i] - 1], 1u << (int)((keys[i] >> child_shift) & 7));
    if (i == n - 1) *count = rank[i];
}

__global__ __launch_bounds__(256) void octree_pack_kernel(const uint32_t* __restrict__ words, const int32_t* __restrict__ count,
                                                          int64_t cap, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cap || i >= *count) return;
    out[i] = (uint8_t)words[i];
}

__global__ __launch_bounds__(256) void octree_last_kernel(const int32_t* __restrict__ rank, int64_t n, int32_t* __restrict__ count) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *count = rank[n - 1];
}

__global__ __launch_bounds__(256) void octree_popc_kernel(const uint8_t* __restrict__ bytes, int64_t n, int32_t* __restrict__ pc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) pc[i] = __popc((unsigned)bytes[i]);
}

__global__ __launch_bounds__(256) void octree_children_kernel(const uint64_t* __restrict__ nodes, const uint8_t* __restrict__ bytes,
                                                              const int32_t* __restrict__ incl, int64_t n, int64_t cap,
                                                              uint64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned b = bytes[i];
    int64_t o = incl[i] - __popc(b);
    const uint64_t base = nodes[i] << 3;
    while (b) {
        const int c = __ffs(b) - 1;
        b &= b - 1;
        if (o < cap) out[o] = base | (uint64_t)c;       // a malformed stream cannot write past the caller's buffer
        ++o;
    }
}

__global__ __launch_bounds__(256) void octree_coords_kernel(const uint64_t* __restrict__ keys, int64_t n, int stride, int ox, int oy,
                                                            int oz, int batch, int32_t* __restrict__ coords) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];
    coords[4 * i + 0] = batch;
    coords[4 * i + 1] = (int)compact3(k >> 2) * stride + ox;
    coords[4 * i + 2] = (int)compact3(k >> 1) * stride + oy;
    coords[4 * i + 3] = (int)compact3(k) * stride + oz;
}

// scratch behind the fixed arrays: radix-sort counters, or the block sums of a scan (never both at once)
static int64_t octree_temp_bytes(int64_t n) {
    const int64_t a = radix_sort_counter_bytes(n), b = align256(scan_block_sums_elems(n) * 4);
    return a > b ? a : b;
}

}  // namespace pcc

using namespace pcc;

extern "C" {

int64_t pcc_octree_scratch_bytes(int64_t n) {
    if (n < 1) n = 1;
    return octree_temp_bytes(n) + 2 * align256(n * 8) + 3 * align256(n * 4) + 256;
}

int pcc_octree_occupancy(const int32_t* coords, int64_t n, int32_t stride, const int32_t* origin, int32_t depth,
                         uint8_t* occupancy, int32_t* level_counts, void* scratch, int64_t scratch_bytes, void* stream) {
    PCC_REQUIRE(n >= 1 && n < (1ll << 31), "pcc_octree_occupancy: n out of range");
    PCC_REQUIRE(stride >= 1 && depth >= 0 && depth <= 21, "pcc_octree_occupancy: bad stride / depth");
    PCC_REQUIRE(scratch_bytes >= pcc_octree_scratch_bytes(n), "pcc_octree_occupancy: scratch too small");
    hipStream_t st = as_stream(stream);
    char* p = reinterpret_cast<char*>(scratch);
    uint64_t* keys_in = reinterpret_cast<uint64_t*>(p); p += align256(n * 8);
    uint64_t* keys = reinterpret_cast<uint64_t*>(p); p += align256(n * 8);
    int32_t* head = reinterpret_cast<int32_t*>(p); p += align256(n * 4);
    int32_t* rank = reinterpret_cast<int32_t*>(p); p += align256(n * 4);
    uint32_t* words = reinterpret_cast<uint32_t*>(p); p += align256(n * 4);
    int32_t* temp = reinterpret_cast<int32_t*>(p);
    const unsigned nb = blocks_for(n, 256);
    PCC_CHECK_HIP(hipMemsetAsync(level_counts, 0, (size_t)(depth + 2) * sizeof(int32_t), st));
    hipLaunchKernelGGL(octree_keys_kernel, dim3(nb), dim3(256), 0, st, coords, n, stride, origin[0], origin[1], origin[2], depth,
                       keys_in, level_counts + depth + 1);
    if (depth > 0) {
        // keys only matter; the values (head / rank double as the value ping-pong) are overwritten below
        const int rc = radix_sort_pairs_u64(keys_in, keys, head, rank, true, n, 0, 3 * depth, temp, st);
        if (rc) return rc;
        if (!radix_sort_result_in_b(0, 3 * depth))
            PCC_CHECK_HIP(hipMemcpyAsync(keys, keys_in, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
    } else {
        PCC_CHECK_HIP(hipMemcpyAsync(keys, keys_in, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
    }
    for (int L = 0; L < depth; ++L) {
        hipLaunchKernelGGL(octree_heads_kernel, dim3(nb), dim3(256), 0, st, keys, n, 3 * (depth - L), head);
        { const int rc = scan_flags(head, n, rank, temp, nullptr, 1, st); if (rc) return rc; }
        PCC_CHECK_HIP(hipMemsetAsync(words, 0, (size_t)n * 4, st));
        hipLaunchKernelGGL(octree_or_kernel, dim3(nb), dim3(256), 0, st, keys, rank, n, 3 * (depth - L - 1), words, level_counts + L);
        hipLaunchKernelGGL(octree_pack_kernel, dim3(nb), dim3(256), 0, st, words, level_counts + L, n, occupancy + (int64_t)L * n);
    }
    // number of distinct leaves (== n unless the input holds duplicates)
    hipLaunchKernelGGL(octree_heads_kernel, dim3(nb), dim3(256), 0, st, keys, n, 0, head);
    { const int rc = scan_flags(head, n, rank, temp, nullptr, 1, st); if (rc) return rc; }
    hipLaunchKernelGGL(octree_last_kernel, dim3(1), dim3(256), 0, st, rank, n, level_counts + depth);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_octree_expand(const uint8_t* occupancy, const int64_t* level_counts, int32_t depth, int32_t stride, const int32_t* origin,
                      int32_t batch, int64_t n_points, int32_t* coords_out, void* scratch, int64_t scratch_bytes, void* stream) {
    PCC_REQUIRE(n_points >= 1 && n_points < (1ll << 31), "pcc_octree_expand: n_points out of range");
    PCC_REQUIRE(stride >= 1 && depth >= 0 && depth <= 21, "pcc_octree_expand: bad stride / depth");
    PCC_REQUIRE(scratch_bytes >= pcc_octree_scratch_bytes(n_points), "pcc_octree_expand: scratch too small");
    PCC_REQUIRE(depth == 0 || level_counts[0] == 1, "pcc_octree_expand: the root level must hold one node");
    for (int L = 0; L < depth; ++L)
        PCC_REQUIRE(level_counts[L] >= 1 && level_counts[L] <= n_points, "pcc_octree_expand: level %d holds %lld nodes for %lld points", L,
                    (long long)level_counts[L], (long long)n_points);
    hipStream_t st = as_stream(stream);
    char* p = reinterpret_cast<char*>(scratch);
    uint64_t* a = reinterpret_cast<uint64_t*>(p); p += align256(n_points * 8);
    uint64_t* b = reinterpret_cast<uint64_t*>(p); p += align256(n_points * 8);
    int32_t* pc = reinterpret_cast<int32_t*>(p); p += align256(n_points * 4);
    int32_t* incl = reinterpret_cast<int32_t*>(p); p += 2 * align256(n_points * 4);
    int32_t* temp = reinterpret_cast<int32_t*>(p);
    PCC_CHECK_HIP(hipMemsetAsync(a, 0, 8, st));     // the root: prefix 0
    int64_t off = 0;
    for (int L = 0; L < depth; ++L) {
        const int64_t nl = level_counts[L];
        const int64_t next = (L + 1 < depth) ? level_counts[L + 1] : n_points;
        const unsigned nb = blocks_for(nl, 256);
        hipLaunchKernelGGL(octree_popc_kernel, dim3(nb), dim3(256), 0, st, occupancy + off, nl, pc);
        { const int rc = scan_flags(pc, nl, incl, temp, nullptr, 1, st); if (rc) return rc; }
        hipLaunchKernelGGL(octree_children_kernel, dim3(nb), dim3(256), 0, st, a, occupancy + off, incl, nl, next, b);
        uint64_t* t = a; a = b; b = t;
        off += nl;
    }
    hipLaunchKernelGGL(octree_coords_kernel, dim3(blocks_for(n_points, 256)), dim3(256), 0, st, a, n_points, stride, origin[0], origin[1],
                       origin[2], batch, coords_out);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

}  // extern "C"
