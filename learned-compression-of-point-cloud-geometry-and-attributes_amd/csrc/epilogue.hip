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

}  // extern "C"
