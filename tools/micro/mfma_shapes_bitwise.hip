// Are two fp32 MFMA shapes interchangeable BIT FOR BIT?  One v_mfma_f32_16x16x4_f32 (contraction over k = 0..3) against the
// chain of two v_mfma_f32_32x32x2_f32 (k = 0,1 then k = 2,3) on the same operands and the same accumulator — what a
// 16-row tile variant of csrc/conv.hip would have to reproduce to stay bit-identical with the 32-row kernels.  Also the
// operand swap (D^T = B^T A^T) of the 32x32x2 shape, which csrc/conv_co.hip relies on.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 mfma_shapes_bitwise.hip -o mfma_shapes_bitwise && ./mfma_shapes_bitwise
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// One wave.  A [32][4], B [4][32], C [32][32] row-major in global memory.
//   out32: C + A B by two 32x32x2 MFMAs (k = 0,1 ; k = 2,3), full 32 x 32
//   out32t: the same through swapped operands (computes the transpose), stored back un-transposed
//   out16: the top-left 16 x 16 block by ONE 16x16x4 MFMA
__global__ void probe(const float* A, const float* B, const float* C, float* out32, float* out32t, float* out16, float* outv) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc, acct;
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        acc[i] = C[row * 32 + r];          // D[row][col = r]
        acct[i] = C[r * 32 + row];         // D^T[row' = col index of D][col' = r = row of D]
    }
    for (int kk = 0; kk < 2; ++kk) {
        const float a = A[r * 4 + 2 * kk + h];        // A[i = r][k = h]
        const float b = B[(2 * kk + h) * 32 + r];     // B[k = h][j = r]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        acct = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acct, 0, 0, 0);   // (B^T)[i = r][k = h] = B[h][r], (A^T)[k = h][j = r] = A[r][h]
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        out32[row * 32 + r] = acc[i];
        out32t[r * 32 + row] = acct[i];
    }
    // 16x16x4: lane l: i / j = l & 15, k = l >> 4; D: lane holds column j = l & 15, rows 4 (l >> 4) + reg
    const int j = lane & 15, kq = lane >> 4;
    f32x4 c16;
    for (int reg = 0; reg < 4; ++reg) c16[reg] = C[(4 * kq + reg) * 32 + j];
    const float a16 = A[j * 4 + kq];                  // A[i = j][k = kq]
    const float b16 = B[kq * 32 + j];                 // B[k = kq][j]
    c16 = __builtin_amdgcn_mfma_f32_16x16x4f32(a16, b16, c16, 0, 0, 0);
    for (int reg = 0; reg < 4; ++reg) out16[(4 * kq + reg) * 16 + j] = c16[reg];
    // the same contraction as a chain of scalar fused multiply-adds (v_fma_f32), k ascending: what csrc/conv.hip's thin kernels run
    for (int e = lane; e < 1024; e += 64) {
        const int i = e >> 5, jj = e & 31;
        float acc1 = C[e];
        for (int k = 0; k < 4; ++k) acc1 = fmaf(A[i * 4 + k], B[k * 32 + jj], acc1);
        outv[e] = acc1;
    }
}

int main() {
    const int trials = 20000;
    float *A, *B, *C, *o32, *o32t, *o16, *ov;
    hipMallocManaged(&ov, 1024 * 4);
    hipMallocManaged(&A, 128 * 4); hipMallocManaged(&B, 128 * 4); hipMallocManaged(&C, 1024 * 4);
    hipMallocManaged(&o32, 1024 * 4); hipMallocManaged(&o32t, 1024 * 4); hipMallocManaged(&o16, 256 * 4);
    srand(1);
    long diff16 = 0, difft = 0, total16 = 0, totalt = 0, diffv = 0;
    for (int t = 0; t < trials; ++t) {
        // magnitudes spread over many binades so that every product / partial sum rounds
        auto rnd = [&](int spread) { return (float)((rand() / (double)RAND_MAX - 0.5) * ldexp(1.0, rand() % spread - spread / 2)); };
        for (int i = 0; i < 128; ++i) { A[i] = rnd(12); B[i] = rnd(12); }
        for (int i = 0; i < 1024; ++i) C[i] = (t & 1) ? rnd(20) : 0.0f;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, A, B, C, o32, o32t, o16, ov);
        hipDeviceSynchronize();
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                ++total16;
                uint32_t x, y;
                __builtin_memcpy(&x, &o32[i * 32 + j], 4); __builtin_memcpy(&y, &o16[i * 16 + j], 4);
                diff16 += x != y;
            }
        for (int i = 0; i < 1024; ++i) {
            ++totalt;
            uint32_t x, y;
            __builtin_memcpy(&x, &o32[i], 4); __builtin_memcpy(&y, &o32t[i], 4);
            difft += x != y;
            __builtin_memcpy(&y, &ov[i], 4);
            diffv += x != y;
        }
    }
    printf("16x16x4 vs two 32x32x2 (k = 0..3): %ld of %ld elements differ bitwise\n", diff16, total16);
    printf("32x32x2 with swapped operands (transposed product): %ld of %ld elements differ bitwise\n", difft, totalt);
    printf("two 32x32x2 MFMAs vs a chain of four v_fma_f32 (k ascending): %ld of %ld elements differ bitwise\n", diffv, totalt);
    return 0;
}
