// Development aid: as mfma_dma.hip but with NBUF LDS buffers and the DMA issued NBUF-1 steps ahead
// (partial s_waitcnt vmcnt before a bare s_barrier), to see whether DMA latency is what idles the MFMA pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
constexpr int CIN = 128, COUT = 128, K = 27;
template <int NBUF, bool RANDOM>
__global__ __launch_bounds__(256) void k(const float* __restrict__ fin, const float* __restrict__ wp, const int* __restrict__ nbr,
                                          float* __restrict__ fout, int n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // NBUF x 64 x 32
    float* Ws = smem + NBUF * 2048;      // NBUF x 8 x 128 x 4
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
    int64_t w_lane_off[4];
    for (int j = 0; j < 4; ++j) { const int f = t + 256 * j; const int g = f / 128, col = f - g * 128; w_lane_off[j] = ((int64_t)g * COUT + col) * 4; }
    const float* a_src[2];
    auto set_src = [&](int kk) {
        for (int i = 0; i < 2; ++i) {
            int64_t src;
            if (RANDOM) { uint64_t hsh = (uint64_t)(row0 + grow[i]) * 0x9E3779B97F4A7C15ull + (uint64_t)kk * 0xBF58476D1CE4E5B9ull; hsh ^= hsh >> 29; src = (int64_t)(hsh % (uint64_t)n); }
            else { src = row0 + grow[i] + 3 * kk; if (src >= n) src -= n; }
            const int q = gchunk ^ ((grow[i] >> 1) & 7);
            a_src[i] = fin + src * CIN + q * 4;
        }
    };
    auto dma = [&](int kk, int c, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[i] + c * 32), (lds_ptr_t)(As + buf * 2048 + (wave_u * 2 + i) * 256), 16, 0, 0);
        const float* wbase = wp + ((int64_t)kk * (CIN / 4) + c * 8) * COUT * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wbase + w_lane_off[j]), (lds_ptr_t)(Ws + buf * 4096 + (wave_u * 64 + 256 * j) * 4), 16, 0, 0);
    };
    const int sw = (r >> 1) & 7;
    int a_off[4];
    for (int kk = 0; kk < 4; ++kk) a_off[kk] = (((kk ^ (sw >> 1)) << 1) | (h ^ (sw & 1))) * 4;
    constexpr int AHEAD = NBUF - 1;
    for (int s0 = 0; s0 < AHEAD; ++s0) { if ((s0 & 3) == 0) set_src(s0 >> 2); dma(s0 >> 2, s0 & 3, s0); }
    if (AHEAD == 1) __builtin_amdgcn_s_waitcnt(0x0070); else __builtin_amdgcn_s_waitcnt(0x0076);   // vmcnt(0) / vmcnt(6)
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int step = 0; step < K * 4; ++step) {
        const int nstep = step + AHEAD;
        int nb = cur + AHEAD; if (nb >= NBUF) nb -= NBUF;
        if (nstep < K * 4) {
            if ((nstep & 3) == 0) set_src(nstep >> 2);
            dma(nstep >> 2, nstep & 3, nb);
        }
        const float* A = As + cur * 2048 + (wrow + r) * 32;
        const float* W = Ws + cur * 4096 + (h * 128 + wcol + r) * 4;
        f32x4 av[2], bv[2][2];
        __builtin_amdgcn_s_setprio(1);
        av[0] = *reinterpret_cast<const f32x4*>(A + a_off[0]);
        bv[0][0] = *reinterpret_cast<const f32x4*>(W);
        bv[0][1] = *reinterpret_cast<const f32x4*>(W + 128);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cb = kk & 1, nbb = cb ^ 1;
            if (kk + 1 < 4) {
                av[nbb] = *reinterpret_cast<const f32x4*>(A + a_off[kk + 1]);
                bv[nbb][0] = *reinterpret_cast<const f32x4*>(W + 2 * (kk + 1) * 512);
                bv[nbb][1] = *reinterpret_cast<const f32x4*>(W + 2 * (kk + 1) * 512 + 128);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) acc[nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][s], bv[cb][nn][s], acc[nn], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        // the image of step+1 must be complete; the DMAs of step+AHEAD (6 per wave, issued above) may stay in flight
        if (AHEAD == 1 || nstep >= K * 4) __builtin_amdgcn_s_waitcnt(0x0070); else __builtin_amdgcn_s_waitcnt(0x0076);
        __builtin_amdgcn_s_barrier();
        cur = (cur + 1 == NBUF) ? 0 : cur + 1;
    }
    for (int nn = 0; nn < 2; ++nn)
        for (int reg = 0; reg < 16; ++reg) {
            const int64_t pos = row0 + wrow + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (pos < n) fout[pos * COUT + wcol + 32 * nn + r] = acc[nn][reg];
        }
}
template <int NBUF, bool RANDOM>
void run(int n, int extra) {
    float *fin, *wp, *fout;
    (void)hipMalloc(&fin, (size_t)n * CIN * 4); (void)hipMalloc(&wp, (size_t)K * CIN * COUT * 4); (void)hipMalloc(&fout, (size_t)n * COUT * 4);
    (void)hipMemset(fin, 0, (size_t)n * CIN * 4); (void)hipMemset(wp, 0, (size_t)K * CIN * COUT * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int lds = NBUF * (2048 + 4096) * 4 + extra;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<NBUF, RANDOM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NBUF, RANDOM>), dim3(n / 64), dim3(256), lds, 0, fin, wp, nullptr, fout, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("%s NBUF=%d lds=%d n=%d  %.3f ms  %.1f TFLOP/s\n", RANDOM ? "random" : "local ", NBUF, lds, n, ms, 2.0 * n * K * CIN * COUT / ms / 1e9);
    }
    (void)hipFree(fin); (void)hipFree(wp); (void)hipFree(fout);
}
int main() { const int n = 5 << 20; run<2, false>(n, 0); run<2, true>(n, 0); run<2, true>(n, 24576); run<3, true>(n, 0); run<4, true>(n, 0); return 0; }
