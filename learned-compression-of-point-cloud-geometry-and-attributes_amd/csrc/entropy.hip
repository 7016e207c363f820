// Entropy-model device kernels: quantisation, table-index build and likelihood evaluation of the
// factorized bottleneck (z) and the Gaussian conditional (y).  Wavefront-level ALU work with no
// dense contraction; formulas: SURVEY.md Appendix B.2 / B.3
// (compressai 1.2.4 EntropyBottleneck / GaussianConditional as used at
// model/entropy_models.py:313,330,352-353,371-372,393,407-408).
//
// Feature matrices are [N, C] row-major; symbol / index / likelihood planes are [C, N]
// (channel-major) — the flattening order of the reference's (1, C, N) tensors, hence the symbol
// order inside each rANS stream.
#include "common.h"

namespace pcc {

constexpr float LIKELIHOOD_BOUND = 1e-9f;
constexpr float SCALE_BOUND = 0.11f;

__global__ __launch_bounds__(256) void eb_quantize_kernel(const float* __restrict__ z, int64_t n, int c,
                                                          const float* __restrict__ med, int32_t* __restrict__ sym,
                                                          float* __restrict__ zhat) {
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / c;
        const int ch = (int)(e - row * c);
        const float m = med[ch];
        const float q = rintf(z[e] - m);
        if (sym) sym[(int64_t)ch * n + row] = (int32_t)q;
        if (zhat) zhat[e] = q + m;
    }
}

__global__ __launch_bounds__(256) void eb_dequantize_kernel(const int32_t* __restrict__ sym, int64_t n, int c,
                                                            const float* __restrict__ med, float* __restrict__ zhat) {
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / c;
        const int ch = (int)(e - row * c);
        zhat[e] = (float)sym[(int64_t)ch * n + row] + med[ch];
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// logits_cumulative for one scalar through the 1-3-3-3-3-1 chain; p = 58 per-channel floats
__device__ __forceinline__ float eb_logits(const float* __restrict__ p, float v) {
    float a[3], b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float t = p[i] * v + p[3 + i];
        a[i] = t + p[6 + i] * tanhf(t);
    }
    const float* q = p + 9;
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float t = q[3 * i] * a[0] + q[3 * i + 1] * a[1] + q[3 * i + 2] * a[2] + q[9 + i];
            b[i] = t + q[12 + i] * tanhf(t);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) a[i] = b[i];
        q += 15;
    }
    return q[0] * a[0] + q[1] * a[1] + q[2] * a[2] + q[3];
}

__global__ __launch_bounds__(256) void eb_likelihood_kernel(const float* __restrict__ zhat, int64_t n, int c,
                                                            const float* __restrict__ params,
                                                            float* __restrict__ lik) {
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / c;
        const int ch = (int)(e - row * c);
        const float* p = params + ch * 58;
        const float v = zhat[e];
        const float lo = eb_logits(p, v - 0.5f);
        const float up = eb_logits(p, v + 0.5f);
        const float sum = lo + up;
        const float s = sum > 0.0f ? -1.0f : (sum < 0.0f ? 1.0f : 0.0f);
        float L = fabsf(sigmoidf_(s * up) - sigmoidf_(s * lo));
        lik[(int64_t)ch * n + row] = fmaxf(L, LIKELIHOOD_BOUND);
    }
}

__global__ __launch_bounds__(256) void gc_encode_prep_kernel(const float* __restrict__ y,
                                                             const float* __restrict__ params, int64_t n, int c,
                                                             const float* __restrict__ table, int levels,
                                                             int32_t* __restrict__ sym, int32_t* __restrict__ idx) {
    __shared__ float tb[256];
    for (int i = threadIdx.x; i < levels; i += 256) tb[i] = table[i];
    __syncthreads();
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / c;
        const int ch = (int)(e - row * c);
        const float scale = fmaxf(params[row * 2 * c + ch], SCALE_BOUND);
        const float mean = params[row * 2 * c + c + ch];
        int ix = levels - 1;
        for (int i = 0; i < levels - 1; ++i) ix -= (scale <= tb[i]) ? 1 : 0;
        const int64_t o = (int64_t)ch * n + row;
        idx[o] = ix;
        if (sym) sym[o] = (int32_t)rintf(y[e] - mean);
    }
}

__global__ __launch_bounds__(256) void gc_dequantize_kernel(const int32_t* __restrict__ sym,
                                                            const float* __restrict__ params, int64_t n, int c,
                                                            float* __restrict__ yhat) {
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / c;
        const int ch = (int)(e - row * c);
        yhat[e] = (float)sym[(int64_t)ch * n + row] + params[row * 2 * c + c + ch];
    }
}

// The same preparation writing the planes the host coder reads directly, in stream order: rows permuted into the
// canonical (b, x, y, z) order on the way (`perm`: output column j takes row perm[j]; the reference sorts the tensors,
// utils.py:155-180), symbols as int16 and table indexes as uint8 — 3 bytes per symbol over PCIe instead of 8, one kernel
// instead of prepare + two index_selects + a stack.  A symbol outside int16 raises *overflow (the caller then takes the
// int32 path; escape coding makes such symbols legal, they just do not occur with sane scales).
__global__ __launch_bounds__(256) void gc_encode_prep_packed_kernel(const float* __restrict__ y,
                                                                    const float* __restrict__ params, int64_t n, int c,
                                                                    const float* __restrict__ table, int levels,
                                                                    const int32_t* __restrict__ perm,
                                                                    int16_t* __restrict__ sym, uint8_t* __restrict__ idx,
                                                                    int32_t* __restrict__ overflow) {
    __shared__ float tb[256];
    for (int i = threadIdx.x; i < levels; i += 256) tb[i] = table[i];
    __syncthreads();
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t j = e / c;
        const int ch = (int)(e - j * c);
        const int64_t row = perm ? perm[j] : j;
        const float scale = fmaxf(params[row * 2 * c + ch], SCALE_BOUND);
        const float mean = params[row * 2 * c + c + ch];
        int ix = levels - 1;
        for (int i = 0; i < levels - 1; ++i) ix -= (scale <= tb[i]) ? 1 : 0;
        const int64_t o = (int64_t)ch * n + j;
        idx[o] = (uint8_t)ix;
        if (sym) {
            const float sv = rintf(y[row * c + ch] - mean);
            if (!(sv >= -32768.0f && sv <= 32767.0f)) atomicOr(overflow, 1);
            sym[o] = (int16_t)fminf(fmaxf(sv, -32768.0f), 32767.0f);
        }
    }
}

__global__ __launch_bounds__(256) void gc_dequantize_i16_kernel(const int16_t* __restrict__ sym,
                                                                const float* __restrict__ params, int64_t n, int c,
                                                                float* __restrict__ yhat) {
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / c;
        const int ch = (int)(e - row * c);
        yhat[e] = (float)sym[(int64_t)ch * n + row] + params[row * 2 * c + c + ch];
    }
}

__device__ __forceinline__ float std_cdf(float x) { return 0.5f * erfcf(-0.70710678118654752440f * x); }

__global__ __launch_bounds__(256) void gc_forward_kernel(const float* __restrict__ y, const float* __restrict__ params,
                                                         int64_t n, int c, float* __restrict__ yhat,
                                                         float* __restrict__ lik) {
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / c;
        const int ch = (int)(e - row * c);
        const float scale = fmaxf(params[row * 2 * c + ch], SCALE_BOUND);
        const float mean = params[row * 2 * c + c + ch];
        const float v = rintf(y[e] - mean) + mean;
        if (yhat) yhat[e] = v;
        if (lik) {
            const float a = fabsf(v - mean);
            const float L = std_cdf((0.5f - a) / scale) - std_cdf((-0.5f - a) / scale);
            lik[(int64_t)ch * n + row] = fmaxf(L, LIKELIHOOD_BOUND);
        }
    }
}

}  // namespace pcc

using namespace pcc;

#define ELEMWISE_GRID(total) dim3(blocks_for((total), 256, 8192)), dim3(256), 0, as_stream(stream)

extern "C" {

int pcc_eb_quantize(const float* z, int64_t n, int32_t c, const float* medians, int32_t* symbols, float* z_hat,
                    void* stream) {
    if (n <= 0) return PCC_OK;
    hipLaunchKernelGGL(eb_quantize_kernel, ELEMWISE_GRID(n * c), z, n, c, medians, symbols, z_hat);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_eb_dequantize(const int32_t* symbols, int64_t n, int32_t c, const float* medians, float* z_hat, void* stream) {
    if (n <= 0) return PCC_OK;
    hipLaunchKernelGGL(eb_dequantize_kernel, ELEMWISE_GRID(n * c), symbols, n, c, medians, z_hat);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_eb_likelihood(const float* z_hat, int64_t n, int32_t c, const float* eb_params, float* lik, void* stream) {
    if (n <= 0) return PCC_OK;
    hipLaunchKernelGGL(eb_likelihood_kernel, ELEMWISE_GRID(n * c), z_hat, n, c, eb_params, lik);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_gc_encode_prep(const float* y, const float* params, int64_t n, int32_t c, const float* scale_table,
                       int32_t levels, int32_t* symbols, int32_t* indexes, void* stream) {
    PCC_REQUIRE(levels >= 2 && levels <= 256, "pcc_gc_encode_prep: levels %d out of range", levels);
    if (n <= 0) return PCC_OK;
    hipLaunchKernelGGL(gc_encode_prep_kernel, ELEMWISE_GRID(n * c), y, params, n, c, scale_table, levels, symbols,
                       indexes);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_gc_encode_prep_packed(const float* y, const float* params, int64_t n, int32_t c, const float* scale_table,
                              int32_t levels, const int32_t* perm, int16_t* symbols, uint8_t* indexes, int32_t* overflow,
                              void* stream) {
    PCC_REQUIRE(levels >= 2 && levels <= 256, "pcc_gc_encode_prep_packed: levels %d out of range", levels);
    PCC_REQUIRE(indexes != nullptr, "pcc_gc_encode_prep_packed: index plane required");
    PCC_REQUIRE(symbols == nullptr || (y != nullptr && overflow != nullptr), "pcc_gc_encode_prep_packed: symbols need y and the overflow word");
    if (overflow) PCC_CHECK_HIP(hipMemsetAsync(overflow, 0, sizeof(int32_t), as_stream(stream)));
    if (n <= 0) return PCC_OK;
    hipLaunchKernelGGL(gc_encode_prep_packed_kernel, ELEMWISE_GRID(n * c), y, params, n, c, scale_table, levels, perm, symbols,
                       indexes, overflow);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_gc_dequantize_i16(const int16_t* symbols, const float* params, int64_t n, int32_t c, float* y_hat, void* stream) {
    if (n <= 0) return PCC_OK;
    hipLaunchKernelGGL(gc_dequantize_i16_kernel, ELEMWISE_GRID(n * c), symbols, params, n, c, y_hat);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_gc_dequantize(const int32_t* symbols, const float* params, int64_t n, int32_t c, float* y_hat, void* stream) {
    if (n <= 0) return PCC_OK;
    hipLaunchKernelGGL(gc_dequantize_kernel, ELEMWISE_GRID(n * c), symbols, params, n, c, y_hat);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_gc_forward(const float* y, const float* params, int64_t n, int32_t c, float* y_hat, float* lik, void* stream) {
    if (n <= 0) return PCC_OK;
    hipLaunchKernelGGL(gc_forward_kernel, ELEMWISE_GRID(n * c), y, params, n, c, y_hat, lik);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

}  // extern "C"
