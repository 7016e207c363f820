// Training-path epilogue of a sparse convolution, forward and backward, one kernel each.
//
// On the inference path FiLM (x * beta + gamma, model/blocks.py:37-40), the activation and the residual add
// (blocks.py:49-52) are fused into the convolution kernel's epilogue.  The training path needs the convolution output
// itself for the backward pass, so there the epilogue is its own differentiable operator: these two kernels replace the
// 4 forward and ~8 backward elementwise torch launches per layer (slice, mul, add, relu, add and their autograd nodes).
// Arithmetic order is the torch graph's (mul, then add; no contraction: -ffp-contract=off), so values and gradients are
// the ones the torch ops produce, bit for bit.  HBM-bound elementwise work: 16-B accesses, no reuse.
#include "common.h"

namespace pcc {

__device__ __forceinline__ float ep_act(float u, int act) {
    if (act == PCC_ACT_RELU) return u > 0.0f ? u : 0.0f;
    if (act == PCC_ACT_LEAKY_RELU) return u > 0.0f ? u : 0.01f * u;
    return u;
}
__device__ __forceinline__ float ep_dact(float u, int act) {
    if (act == PCC_ACT_RELU) return u > 0.0f ? 1.0f : 0.0f;
    if (act == PCC_ACT_LEAKY_RELU) return u > 0.0f ? 1.0f : 0.01f;
    return 1.0f;
}

// one thread per (row, 4 channels); C % 4 == 0
__global__ __launch_bounds__(256) void epilogue_fwd_kernel(const float4* __restrict__ c, const float4* __restrict__ film,
                                                           const float4* __restrict__ residual, int64_t n, int c4, int act,
                                                           float4* __restrict__ out) {
    const int64_t total = n * c4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / c4;
        const int q = (int)(e - row * c4);
        float4 u = c[e];
        if (film) {
            const float4 b = film[row * 2 * c4 + q], g = film[row * 2 * c4 + c4 + q];
            u.x = u.x * b.x; u.y = u.y * b.y; u.z = u.z * b.z; u.w = u.w * b.w;
            u.x = u.x + g.x; u.y = u.y + g.y; u.z = u.z + g.z; u.w = u.w + g.w;
        }
        float4 v = make_float4(ep_act(u.x, act), ep_act(u.y, act), ep_act(u.z, act), ep_act(u.w, act));
        if (residual) {
            const float4 r = residual[e];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        out[e] = v;
    }
}

__global__ __launch_bounds__(256) void epilogue_bwd_kernel(const float4* __restrict__ dout, const float4* __restrict__ c,
                                                           const float4* __restrict__ film, int64_t n, int c4, int act,
                                                           float4* __restrict__ dc, float4* __restrict__ dfilm) {
    const int64_t total = n * c4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / c4;
        const int q = (int)(e - row * c4);
        const float4 cv = c[e];
        float4 u = cv, b = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        if (film) {
            b = film[row * 2 * c4 + q];
            const float4 g = film[row * 2 * c4 + c4 + q];
            u.x = u.x * b.x; u.y = u.y * b.y; u.z = u.z * b.z; u.w = u.w * b.w;
            u.x = u.x + g.x; u.y = u.y + g.y; u.z = u.z + g.z; u.w = u.w + g.w;
        }
        const float4 d = dout[e];
        const float4 du = make_float4(d.x * ep_dact(u.x, act), d.y * ep_dact(u.y, act), d.z * ep_dact(u.z, act), d.w * ep_dact(u.w, act));
        if (film) {
            dc[e] = make_float4(du.x * b.x, du.y * b.y, du.z * b.z, du.w * b.w);
            dfilm[row * 2 * c4 + q] = make_float4(du.x * cv.x, du.y * cv.y, du.z * cv.z, du.w * cv.w);
            dfilm[row * 2 * c4 + c4 + q] = du;
        } else {
            dc[e] = du;
        }
    }
}

// ---- dY of a convolution read ONCE: its bf16 copy (what the bf16 weight-gradient and backward-data kernels gather; round to
// nearest even, NaN -> 0x7FC0: torch's conversion) and its column sums (the bias gradient).  A block owns a contiguous range of
// rows; thread (row lane, 4 columns) walks it, the row lanes are folded in a fixed order through LDS, and a second kernel adds
// the per-block partials in block order: deterministic, unlike an atomic reduction.
constexpr int CS_MAX_BLOCKS = 1024;
constexpr int CS_FINISH_COLS = 16;      // columns per workgroup of the second stage: 16 lanes of blocks per column

__device__ __forceinline__ unsigned short bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ __launch_bounds__(256) void cast_colsum_kernel(const float4* __restrict__ x, int64_t n, int c4, int64_t rows_per_block,
                                                          ushort4* __restrict__ out_bf16, float4* __restrict__ partial) {
    __shared__ float4 red[256];
    const int lanes = 256 / c4;
    const int t = threadIdx.x, rl = t / c4, q = t - rl * c4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (rl < lanes) {
        for (int64_t r = r0 + rl; r < r1; r += lanes) {
            const float4 v = x[r * c4 + q];
            if (out_bf16) out_bf16[r * c4 + q] = make_ushort4(bf16_rne(v.x), bf16_rne(v.y), bf16_rne(v.z), bf16_rne(v.w));
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (!partial) return;
    red[t] = acc;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < lanes; ++j) {
            const float4 o = red[j * c4 + q];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        partial[(int64_t)blockIdx.x * c4 + q] = acc;
    }
}

// colsum[col] = sum over blocks: a workgroup owns CS_FINISH_COLS columns, its 256 / CS_FINISH_COLS block lanes each add a
// contiguous piece of the block range in ascending order, and the pieces are folded in piece order — a fixed order, and
// c / 16 workgroups with 64 loads per thread instead of two workgroups with 512 (the first version: 100 us per call)
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, int nblocks, int c,
                                                            float* __restrict__ colsum) {
    __shared__ float red[256];
    const int cols = c < CS_FINISH_COLS ? c : CS_FINISH_COLS;
    const int lanes = 256 / cols;
    const int t = threadIdx.x, pl = t / cols, cl = t - pl * cols;
    const int col = blockIdx.x * cols + cl;
    float acc = 0.0f;
    if (pl < lanes && col < c) {
        const int per = (nblocks + lanes - 1) / lanes;
        const int b0 = pl * per, b1 = (b0 + per < nblocks) ? b0 + per : nblocks;
        for (int b = b0; b < b1; ++b) acc += partial[(int64_t)b * c + col];
    }
    red[t] = acc;
    __syncthreads();
    if (pl == 0 && col < c) {
        for (int j = 1; j < lanes; ++j) acc += red[j * cols + cl];
        colsum[col] = acc;
    }
}

}  // namespace pcc

using namespace pcc;

extern "C" {

int pcc_epilogue_fwd(const float* c, const float* film, const float* residual, int64_t n, int32_t channels, int32_t act, float* out,
                     void* stream) {
    PCC_REQUIRE(channels >= 4 && channels % 4 == 0, "pcc_epilogue_fwd: channel count %d must be a multiple of 4", channels);
    PCC_REQUIRE(act >= 0 && act <= 2, "pcc_epilogue_fwd: bad activation %d", act);
    PCC_REQUIRE(((reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(film) | reinterpret_cast<uintptr_t>(residual) |
                  reinterpret_cast<uintptr_t>(out)) & 15) == 0, "pcc_epilogue_fwd: tensors must be 16-byte aligned");
    if (n <= 0) return PCC_OK;
    hipLaunchKernelGGL(epilogue_fwd_kernel, dim3(blocks_for(n * (channels / 4), 256, 16384)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(c), reinterpret_cast<const float4*>(film), reinterpret_cast<const float4*>(residual), n,
                       channels / 4, act, reinterpret_cast<float4*>(out));
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_epilogue_bwd(const float* dout, const float* c, const float* film, int64_t n, int32_t channels, int32_t act, float* dc,
                     float* dfilm, void* stream) {
    PCC_REQUIRE(channels >= 4 && channels % 4 == 0, "pcc_epilogue_bwd: channel count %d must be a multiple of 4", channels);
    PCC_REQUIRE(act >= 0 && act <= 2, "pcc_epilogue_bwd: bad activation %d", act);
    PCC_REQUIRE((film == nullptr) == (dfilm == nullptr), "pcc_epilogue_bwd: dfilm goes with film");
    PCC_REQUIRE(((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(film) |
                  reinterpret_cast<uintptr_t>(dc) | reinterpret_cast<uintptr_t>(dfilm)) & 15) == 0,
                "pcc_epilogue_bwd: tensors must be 16-byte aligned");
    if (n <= 0) return PCC_OK;
    hipLaunchKernelGGL(epilogue_bwd_kernel, dim3(blocks_for(n * (channels / 4), 256, 16384)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(dout), reinterpret_cast<const float4*>(c), reinterpret_cast<const float4*>(film), n,
                       channels / 4, act, reinterpret_cast<float4*>(dc), reinterpret_cast<float4*>(dfilm));
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int64_t pcc_cast_colsum_scratch_elems(int32_t channels) { return (int64_t)CS_MAX_BLOCKS * channels; }

int pcc_cast_colsum(const float* x, int64_t n, int32_t channels, uint16_t* out_bf16, float* colsum, float* scratch,
                    int64_t scratch_elems, void* stream) {
    PCC_REQUIRE(channels >= 4 && channels % 4 == 0 && channels <= 1024, "pcc_cast_colsum: channel count %d must be a multiple of 4, at most 1024", channels);
    PCC_REQUIRE(out_bf16 != nullptr || colsum != nullptr, "pcc_cast_colsum: nothing to do");
    PCC_REQUIRE(((reinterpret_cast<uintptr_t>(x) & 15) | (reinterpret_cast<uintptr_t>(out_bf16) & 7) |
                 (reinterpret_cast<uintptr_t>(scratch) & 15)) == 0, "pcc_cast_colsum: x / scratch must be 16-byte and out_bf16 8-byte aligned");
    hipStream_t st = as_stream(stream);
    if (n <= 0) {
        if (colsum) PCC_CHECK_HIP(hipMemsetAsync(colsum, 0, (size_t)channels * sizeof(float), st));
        return PCC_OK;
    }
    const int c4 = channels / 4;
    const int lanes = 256 / c4;
    int64_t nb = (n + 4 * lanes - 1) / (4 * lanes);            // at least four passes of the row lanes per block
    if (nb > CS_MAX_BLOCKS) nb = CS_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    const int64_t rpb = (n + nb - 1) / nb;
    nb = (n + rpb - 1) / rpb;
    if (colsum) PCC_REQUIRE(scratch != nullptr && scratch_elems >= nb * channels, "pcc_cast_colsum: scratch too small");
    hipLaunchKernelGGL(cast_colsum_kernel, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const float4*>(x), n, c4, rpb,
                       reinterpret_cast<ushort4*>(out_bf16), colsum ? reinterpret_cast<float4*>(scratch) : nullptr);
    if (colsum) {
        const int cols = channels < CS_FINISH_COLS ? channels : CS_FINISH_COLS;
        hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((channels + cols - 1) / cols)), dim3(256), 0, st, scratch, (int)nb,
                           channels, colsum);
    }
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

}  // extern "C"
