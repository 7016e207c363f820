// Development aid: the leanest version of the conv main loop: buffer_load ... lds with scalar per-step
// offsets, compile-time LDS buffer parity (no per-step VALU address math).  MODE 1 adds one neighbour
// index load per offset (buffer_load_dword, OOB-safe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int CIN = 128, COUT = 128, K = 27, CCH = CIN / 32;
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ fin, const float* __restrict__ wp, const int* __restrict__ nbr,
                                          float* __restrict__ fout, int n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // 2 x 64 x 32
    float* Ws = smem + 2 * 2048;      // 2 x 8 x 128 x 4
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wrow = (wid >> 1) * 32, wcol = (wid & 1) * 64;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    f32x16 acc[2];
    for (int nn = 0; nn < 2; ++nn) for (int i = 0; i < 16; ++i) acc[nn][i] = 0.f;
    const int gchunk = t & 7;
    int grow[2];
    for (int i = 0; i < 2; ++i) grow[i] = wid * 16 + 8 * i + (lane >> 3);
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    unsigned a_voff[2], w_voff[4], q16[2];
    __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fin), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, K * CIN * COUT * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(nbr), 0, 0x7fffffff, 0x00020000);
    for (int j = 0; j < 4; ++j) { const int f = t + 256 * j; const int g = f / 128, col = f - g * 128; w_voff[j] = (unsigned)((g * COUT + col) * 16); }
    unsigned n_voff[2];
    int idx_nxt[2];
    for (int i = 0; i < 2; ++i) { q16[i] = (gchunk ^ ((grow[i] >> 1) & 7)) * 16; n_voff[i] = (unsigned)((row0 + grow[i]) * K * 4); }
    auto set_src = [&](int kk) {
        for (int i = 0; i < 2; ++i) {
            unsigned src;
            if (MODE & 1) src = (unsigned)idx_nxt[i];
            else { src = (unsigned)(row0 + grow[i] + 3 * kk); if (src >= (unsigned)n) src -= n; }
            a_voff[i] = src * (CIN * 4) + q16[i];
        }
    };
    auto load_idx = [&](int kk) {
        if (MODE & 1)
            for (int i = 0; i < 2; ++i) idx_nxt[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc_n, n_voff[i], kk * 4, 0);
    };
    const int sw = (r >> 1) & 7;
    // lane-constant LDS byte addresses of my fragments, for both buffers
    unsigned a_addr[2][4], w_addr[2];
    for (int b = 0; b < 2; ++b) {
        for (int kk = 0; kk < 4; ++kk)
            a_addr[b][kk] = (unsigned)((b * 2048 + (wrow + r) * 32 + (((kk ^ (sw >> 1)) << 1) | (h ^ (sw & 1))) * 4) * 4);
        w_addr[b] = (unsigned)((2 * 2048 + b * 4096 + (h * 128 + wcol + r) * 4) * 4);
    }
    auto dma = [&](int kk, int c, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(As + buf * 2048 + (wave_u * 2 + i) * 256), 16, a_voff[i], c * 128, 0, 0);
        const int wso = (kk * (CIN / 4) + c * 8) * COUT * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(Ws + buf * 4096 + (wave_u * 64 + 256 * j) * 4), 16, w_voff[j], wso, 0, 0);
    };
    auto lds4 = [&](unsigned addr) { return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(smem) + addr); };
    auto compute = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        f32x4 av[2], bv[2][2];
        __builtin_amdgcn_s_setprio(1);
        av[0] = lds4(a_addr[buf][0]);
        bv[0][0] = lds4(w_addr[buf]);
        bv[0][1] = lds4(w_addr[buf] + 512);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cb = kk & 1, nbb = cb ^ 1;
            if (kk + 1 < 4) {
                av[nbb] = lds4(a_addr[buf][kk + 1]);
                bv[nbb][0] = lds4(w_addr[buf] + 2 * (kk + 1) * 2048);
                bv[nbb][1] = lds4(w_addr[buf] + 2 * (kk + 1) * 2048 + 512);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) acc[nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][s], bv[cb][nn][s], acc[nn], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    load_idx(0);
    set_src(0);
    load_idx(1);
    dma(0, 0, B0{});
    __syncthreads();
    for (int kk = 0; kk < K; ++kk) {
        // chunks 0..2 prefetch the next chunk of the same offset; chunk 3 switches to the next offset
        dma(kk, 1, B1{}); compute(B0{}); __syncthreads();
        dma(kk, 2, B0{}); compute(B1{}); __syncthreads();
        dma(kk, 3, B1{}); compute(B0{}); __syncthreads();
        if (kk + 1 < K) { set_src(kk + 1); dma(kk + 1, 0, B0{}); load_idx(kk + 2 < K ? kk + 2 : kk); }
        compute(B1{}); __syncthreads();
    }
    for (int nn = 0; nn < 2; ++nn)
        for (int reg = 0; reg < 16; ++reg) {
            const int64_t pos = row0 + wrow + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (pos < n) fout[pos * COUT + wcol + 32 * nn + r] = acc[nn][reg];
        }
}
template <int MODE>
void run(int n) {
    float *fin, *wp, *fout; int* nbr;
    (void)hipMalloc(&fin, (size_t)n * CIN * 4); (void)hipMalloc(&wp, (size_t)K * CIN * COUT * 4); (void)hipMalloc(&fout, (size_t)n * COUT * 4);
    (void)hipMalloc(&nbr, (size_t)n * K * 4);
    (void)hipMemset(fin, 0, (size_t)n * CIN * 4); (void)hipMemset(wp, 0, (size_t)K * CIN * COUT * 4);
    int* hn = (int*)malloc((size_t)n * K * 4);
    for (int64_t i = 0; i < n; ++i) for (int kk = 0; kk < K; ++kk) hn[i * K + kk] = (int)((i + 3 * kk) % n);
    (void)hipMemcpy(nbr, hn, (size_t)n * K * 4, hipMemcpyHostToDevice); free(hn);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int lds = 2 * (2048 + 4096) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(n / 64), dim3(256), lds, 0, fin, wp, nbr, fout, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("MODE=%d n=%d  %.3f ms  %.1f TFLOP/s\n", MODE, n, ms, 2.0 * n * K * CIN * COUT / ms / 1e9);
    }
    (void)hipFree(fin); (void)hipFree(wp); (void)hipFree(fout); (void)hipFree(nbr);
}
int main() { run<0>(1 << 20); run<1>(1 << 20); return 0; }
