// Development aid: minimal double-buffered LDS-DMA + MFMA loop with the tile shape of
// conv_mfma_kernel<64,128,2,2> (A 64x32, W 32x128 per step; 27 offsets x 4 chunks), to find what
// the bare structure reaches before neighbour tables, masks and epilogues are added.
//   MODE 0: A rows = tile rows shifted by 3k (no index table)      MODE 1: + nbr table read per offset
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
constexpr int CIN = 128, COUT = 128, K = 27;
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
    int64_t w_lane_off[4];
    for (int j = 0; j < 4; ++j) { const int f = t + 256 * j; const int g = f / 128, col = f - g * 128; w_lane_off[j] = ((int64_t)g * COUT + col) * 4; }
    const float* a_src[2];
    unsigned a_voff[2], w_voff[4];
    __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fin), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, K * CIN * COUT * 4, 0x00020000);
    for (int j = 0; j < 4; ++j) w_voff[j] = (unsigned)(w_lane_off[j] * 4);
    auto set_src = [&](int kk) {
        for (int i = 0; i < 2; ++i) {
            int64_t src;
            if (MODE & 1) src = nbr[(row0 + grow[i]) * K + kk];
            else src = (row0 + grow[i] + 3 * kk) % n;
            const int q = gchunk ^ ((grow[i] >> 1) & 7);
            a_src[i] = fin + src * CIN + q * 4;
            a_voff[i] = (unsigned)((src * CIN + q * 4) * 4);
        }
    };
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
    auto dma = [&](int kk, int c, int buf) {
        if (MODE & 16) {
#pragma unroll
            for (int i = 0; i < 2; ++i) sink += *reinterpret_cast<const f32x4*>(a_src[i] + c * 32);
            const float* wb = wp + ((int64_t)kk * (CIN / 4) + c * 8) * COUT * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) sink += *reinterpret_cast<const f32x4*>(wb + w_lane_off[j]);
            return;
        }
        if (MODE & 64) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(As + buf * 2048 + (wave_u * 2 + i) * 256), 16, a_voff[i], c * 128, 0, 0);
            const int wso = (kk * (CIN / 4) + c * 8) * COUT * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(Ws + buf * 4096 + (wave_u * 64 + 256 * j) * 4), 16, w_voff[j], wso, 0, 0);
            return;
        }
        if (MODE & 32) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(fin + (t & 63) * 4), (lds_ptr_t)(As + buf * 2048 + (wave_u * 2 + i) * 256), 16, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wp + (t & 63) * 4), (lds_ptr_t)(Ws + buf * 4096 + (wave_u * 64 + 256 * j) * 4), 16, 0, 0);
            return;
        }
        if (!(MODE & 2))
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[i] + c * 32), (lds_ptr_t)(As + buf * 2048 + (wave_u * 2 + i) * 256), 16, 0, 0);
        const float* wbase = wp + ((int64_t)kk * (CIN / 4) + c * 8) * COUT * 4;
        if (!(MODE & 4))
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wbase + w_lane_off[j]), (lds_ptr_t)(Ws + buf * 4096 + (wave_u * 64 + 256 * j) * 4), 16, 0, 0);
    };
    const int sw = (r >> 1) & 7;
    int a_off[4];
    for (int kk = 0; kk < 4; ++kk) a_off[kk] = (((kk ^ (sw >> 1)) << 1) | (h ^ (sw & 1))) * 4;
    set_src(0);
    dma(0, 0, 0);
    __syncthreads();
    int cur = 0;
    for (int step = 0; step < K * 4; ++step) {
        const int nstep = step + 1;
        if (!(MODE & 8) && nstep < K * 4) {
            if ((nstep & 3) == 0) set_src(nstep >> 2);
            dma(nstep >> 2, nstep & 3, cur ^ 1);
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
            const int cb = kk & 1, nb = cb ^ 1;
            if (kk + 1 < 4) {
                av[nb] = *reinterpret_cast<const f32x4*>(A + a_off[kk + 1]);
                bv[nb][0] = *reinterpret_cast<const f32x4*>(W + 2 * (kk + 1) * 512);
                bv[nb][1] = *reinterpret_cast<const f32x4*>(W + 2 * (kk + 1) * 512 + 128);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) acc[nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][s], bv[cb][nn][s], acc[nn], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if ((MODE & 8) && nstep < K * 4) {
            if ((nstep & 3) == 0) set_src(nstep >> 2);
            dma(nstep >> 2, nstep & 3, cur ^ 1);
        }
        __syncthreads();
        cur ^= 1;
    }
    for (int nn = 0; nn < 2; ++nn)
        for (int reg = 0; reg < 16; ++reg) {
            const int64_t pos = row0 + wrow + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (pos < n) fout[pos * COUT + wcol + 32 * nn + r] = acc[nn][reg] + sink[0] + sink[1] + sink[2] + sink[3];
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
int main() { run<0>(1 << 20); run<1>(1 << 20); run<2>(1 << 20); run<4>(1 << 20); run<6>(1 << 20); run<64>(1 << 20); run<65>(1 << 20); return 0; }
