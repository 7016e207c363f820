// Gather side of the 64 -> 64 convolution in isolation: uniformly random 256-B rows of a 1.3 GB fp32 matrix
// into LDS by buffer_load ... lds, 128 rows per workgroup and step, double-buffered, no MFMA.
//   MODE 0: a row's two 128-B halves are requested in consecutive steps (what conv_mfma_buf_kernel does at
//           cin = 64: one 32-channel chunk per step)
//   MODE 1: both halves in the same step (a 64-channel step)
// Prints TB/s of gathered bytes.  hipcc -O3 -std=c++17 --offload-arch=gfx950 gather_rows.hip -o gather_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const float* x, const int* idx, int n_rows, int steps, int* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];       // 2 x 128 rows x (MODE ? 256 : 128) B
    constexpr int ROWB = MODE ? 256 : 128;
    constexpr int BUF = 128 * ROWB / 4;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((uint32_t)n_rows * 256u), 0x00020000);
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    // MODE 0: 8 lanes per row, 8 rows per instruction, 4 instructions per wave (32 rows of 128 B)
    // MODE 1: 16 lanes per row, 4 rows per instruction, 8 instructions per wave (32 rows of 256 B)
    constexpr int LPR = ROWB / 16, RPI = 64 / LPR, NI = 32 / RPI;
    const int* my = idx + (size_t)blockIdx.x * steps * 128;
    int acc = 0;
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        const int srow = MODE ? s : (s >> 1);                       // MODE 0: steps 2j and 2j+1 fetch the halves of row set j
        const int half = MODE ? 0 : (s & 1);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = w * 32 + i * RPI + lane / LPR;
            const int row = my[srow * 128 + r];
            const uint32_t off = (uint32_t)row * 256u + (uint32_t)(half * 128 + (lane % LPR) * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf * BUF + (w * NI + i) * 256), 16, off, 0, 0, 0);
        }
        __syncthreads();
        acc += __builtin_bit_cast(int, smem[buf * BUF + t]);
    }
    if (acc == 0x7fffffff) *sink = acc;
#endif
}

int main() {
    const int n_rows = 5156101, steps = 64, blocks = 256 * 3 * 8;
    float* x; int* idx; int* sink;
    hipMalloc(&x, (size_t)n_rows * 256);
    hipMemset(x, 0, (size_t)n_rows * 256);
    std::vector<int> h((size_t)blocks * steps * 128);
    std::mt19937 g(1);
    for (auto& v : h) v = (int)(g() % n_rows);
    hipMalloc(&idx, h.size() * 4);
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(256), 2 * 128 * 128, 0, x, idx, n_rows, steps, sink);
            else hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 2 * 128 * 256, 0, x, idx, n_rows, steps / 2, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)blocks * (steps / 2) * 128 * 256;      // both modes gather the same rows once
            printf("mode %d (%s): %.3f ms  %.2f TB/s\n", mode, mode ? "256 B per step" : "2 x 128 B in consecutive steps", ms, bytes / ms / 1e9);
        }
    }
    return 0;
}
