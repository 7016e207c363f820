// Development aid: MFMA stream fed from LDS the way conv_mfma_kernel<64,128,2,2> feeds it
// (per sub-block: 1 A + 2 B ds_read_b128, 8 MFMAs, fragments double-buffered), no global traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 12288; i += 256) smem[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    const float* Ab = smem + ((wid >> 1) * 32 + r) * 32 + h * 4;
    const float* Wb = smem + 2048 + (h * 128 + (wid & 1) * 64 + r) * 4;
    for (int it = 0; it < iters; ++it) {
        const float* A = Ab + (it & 1) * 6144;
        const float* W = Wb + (it & 1) * 6144;
        f32x4 av[2], bv[2][2];
        if (MODE != 1) __builtin_amdgcn_s_setprio(1);
        av[0] = *reinterpret_cast<const f32x4*>(A);
        bv[0][0] = *reinterpret_cast<const f32x4*>(W);
        bv[0][1] = *reinterpret_cast<const f32x4*>(W + 128);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cb = kk & 1, nb = cb ^ 1;
            if (kk + 1 < 4) {
                av[nb] = *reinterpret_cast<const f32x4*>(A + 8 * (kk + 1));
                bv[nb][0] = *reinterpret_cast<const f32x4*>(W + 2 * (kk + 1) * 512);
                bv[nb][1] = *reinterpret_cast<const f32x4*>(W + 2 * (kk + 1) * 512 + 128);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][s], bv[cb][n][s], acc[n], 0, 0, 0);
        }
        if (MODE != 1) __builtin_amdgcn_s_setprio(0);
        if (MODE == 2) __syncthreads();
    }
    float s = 0.f;
    for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(int lds) {
    float* out; (void)hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 2000, blocks = 256 * 12;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("MODE=%d lds=%d  %.3f ms  %.1f TFLOP/s\n", MODE, lds, ms, (double)blocks * 4 * iters * 32 * 4096.0 / ms / 1e9);
    }
    (void)hipFree(out);
}
int main() { run<0>(49152); run<1>(49152); run<2>(49152); run<0>(65536); run<0>(98304); run<2>(32768); return 0; }
