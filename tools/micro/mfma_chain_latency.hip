// How long does a DEPENDENT chain of fp32 MFMAs take per instruction?  One workgroup of 4 waves (one per SIMD), each wave
// running N x CH MFMAs as CH independent accumulator chains, interleaved — the time of a small convolution launch is one
// wave's chain (csrc/conv.hip, conv_small_kernel), so this is its floor.  Timed with HIP events over a long chain.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 mfma_chain_latency.hip -o mfma_chain_latency && ./mfma_chain_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(256) void chain16(float* out, int n, float a, float b) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// one chain, NV lane-selects (v_cndmask_b32) in front of every MFMA: does the vector ALU work hide in the MFMA's shadow?
template <int NV>
__global__ __launch_bounds__(256) void chain16_valu(float* out, int n, float a, float b) {
    f32x4 acc = {0, 0, 0, 0};
    const bool x = (threadIdx.x & 2) != 0, y = (threadIdx.x & 16) != 0;
    float p0 = a, p1 = a * 1.5f, q0 = b, q1 = b * 0.5f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float aa = p0, bb = q0;
            if (NV >= 1) aa = x ? p1 : p0;
            if (NV >= 2) bb = y ? q1 : q0;
            if (NV >= 3) aa = y ? aa : q1;
            if (NV >= 4) bb = x ? bb : p1;
            asm volatile("" : "+v"(aa), "+v"(bb));          // keep the selects in front of this MFMA
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, bb, acc, 0, 0, 0);
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(q0), "+v"(q1));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[3];
}

// the same selects, but (MODE 0) feeding nothing the MFMAs read, or (MODE 1) feeding the MFMA after the next one
template <int MODE>
__global__ __launch_bounds__(256) void chain16_valu_apart(float* out, int n, float a, float b) {
    f32x4 acc = {0, 0, 0, 0};
    const bool x = (threadIdx.x & 2) != 0, y = (threadIdx.x & 16) != 0;
    float p0 = a, p1 = a * 1.5f, q0 = b, q1 = b * 0.5f, junk = 0.0f;
    float a1 = a, b1 = b, a2 = a, b2 = b;          // operands of the next MFMA / of the one after
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float aa = x ? p1 : p0, bb = y ? q1 : q0;
            asm volatile("" : "+v"(aa), "+v"(bb));
            if (MODE == 0) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p0, q0, acc, 0, 0, 0);
                junk += aa * bb;
            } else {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
                a1 = a2; b1 = b2; a2 = aa; b2 = bb;
            }
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(q0), "+v"(q1));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[3] + junk;
}

// ND LDS reads (ds_read_b32, results unused by the chain) or NS scalar instructions between two dependent MFMAs
template <int ND, int NSALU>
__global__ __launch_bounds__(256) void chain16_lds(float* out, int n, float a, float b) {
    __shared__ float sm[1024];
    sm[threadIdx.x] = a; sm[threadIdx.x + 256] = b; sm[threadIdx.x + 512] = a; sm[threadIdx.x + 768] = b;
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
    float junk[4] = {0, 0, 0, 0};
    const float* p = sm + threadIdx.x;
    int sacc = n;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int d = 0; d < ND; ++d) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(junk[d]) : "v"((uint32_t)(uintptr_t)p), "n"(256 * 0));
#pragma unroll
            for (int d = 0; d < NSALU; ++d) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[3] + junk[0] + junk[1] + junk[2] + junk[3] + (float)sacc;
}

template <int CH>
__global__ __launch_bounds__(256) void chain32(float* out, int n, float a, float b) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int j = 0; j < 16; ++j) acc[c][j] = 0;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K>
static void run(const char* name, K kern, int ch, int blocks, float* out) {
    const int n = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, n, 1.0f, 1e-9f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / ((double)n * 8 * ch);
    printf("%-28s %4d workgroup(s): %7.2f ns per MFMA per wave (%6.2f ns per step of one chain)\n", name, blocks, per, per * ch);
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    for (int blocks : {1, 256}) {
        run("16x16x4, 1 chain", chain16<1>, 1, blocks, out);
        run("16x16x4, 2 chains", chain16<2>, 2, blocks, out);
        run("16x16x4, 4 chains", chain16<4>, 4, blocks, out);
        run("16x16x4, 1 chain + 1 select", chain16_valu<1>, 1, blocks, out);
        run("16x16x4, 1 chain + 2 selects", chain16_valu<2>, 1, blocks, out);
        run("16x16x4, 1 chain + 4 selects", chain16_valu<4>, 1, blocks, out);
        run("16x16x4 + 2 unrelated selects", chain16_valu_apart<0>, 1, blocks, out);
        run("16x16x4 + 2 selects 2 ahead", chain16_valu_apart<1>, 1, blocks, out);
        run("16x16x4 + 2 ds_read_b32", chain16_lds<2, 0>, 1, blocks, out);
        run("16x16x4 + 4 ds_read_b32", chain16_lds<4, 0>, 1, blocks, out);
        run("16x16x4 + 4 s_add", chain16_lds<0, 4>, 1, blocks, out);
        run("32x32x2, 1 chain", chain32<1>, 1, blocks, out);
        run("32x32x2, 2 chains", chain32<2>, 2, blocks, out);
    }
    return 0;
}
