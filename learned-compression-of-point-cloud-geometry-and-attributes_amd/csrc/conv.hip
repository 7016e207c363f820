// Sparse convolution forward for gfx950.
//
// Output-stationary implicit GEMM: a workgroup owns BM (64) output rows x BN output channels and
// walks the kernel offsets k (in ascending order) and the input channels in chunks of 32.  For each
// (k, chunk) it gathers the neighbour rows' 128-byte slices straight into LDS by LDS-DMA (a line
// of zeros for absent neighbours), stages the matching 32 x BN weight slab, and feeds fp32 MFMA
// (v_mfma_f32_32x32x2_f32; exact fp32, SURVEY.md §7 "fp32 parity budget").  Accumulation order per
// output element is fixed (k ascending, channel ascending inside the MFMA chain), so a result
// depends only on the element's own neighbourhood — never on row order, tile placement or
// arrival order.  That is what lets encoder and decoder reproduce h_s bit-exactly
// (the job of the reference's Sorted* shims, model/entropy_models.py:12-102).
//
// Rows are executed in neighbour-mask order (pcc_order_rows_by_mask); offsets that no row of a 32-row
// MFMA tile has are skipped: adding an all-zero product is exact, so skipping does not change results.
//
// Roofline: MFMA fp32 (157 TFLOP/s); algorithmic FLOPs per launch = 2 * pairs * cin * cout.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace pcc {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ConvArgs {
    const float* fin;
    const float* w;       // raw [K, cin, cout] (thin path)
    const float* wp;      // packed [K, cinp/4, coutp, 4] (MFMA path)
    const float* bias;    // [cout] or null
    const int32_t* nbr;   // [n_out, K] by OUTPUT ROW, or null (identity); the MFMA kernels read row order[pos] of it
    const int32_t* order;   // [n_out] execution position -> output row, or null (natural order)
    const uint32_t* gmask;  // [ceil(n_out/32)] offsets live per 32 positions, or null (all live)
    float* fout;
    const float* film;      // [n_out, 2*cout] or null
    const float* residual;  // [n_out, cout] or null
    int64_t n_in, n_out;
    int cin, cout, coutp, K, act;
    int bf16;    // buffer kernel: fin / wp hold bf16 (pcc_conv_fwd_bf16); accumulation and output stay fp32
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == PCC_ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == PCC_ACT_LEAKY_RELU) return v > 0.0f ? v : 0.01f * v;
    return v;
}

// ---------------------------------------------------------------------------------------------
// weight packing: Wp[k][g][col][s] = W[k][4g+s][col], zero padded to cinp (x32) and coutp (x32)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int K, int cin, int cout,
                                                           int cinp, int coutp, float* __restrict__ wp) {
    const int64_t total = (int64_t)K * cinp * coutp;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(e & 3);
        int64_t t = e >> 2;
        const int col = (int)(t % coutp);
        t /= coutp;
        const int g = (int)(t % (cinp / 4));
        const int k = (int)(t / (cinp / 4));
        const int ci = 4 * g + s;
        wp[e] = (ci < cin && col < cout) ? w[((int64_t)k * cin + ci) * cout + col] : 0.0f;
    }
}

// bf16 packing: Wp[k][g][col][j] = bf16(W[k][8g+j][col]), zero padded to cinp (x64) and coutp (x32)
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const float* __restrict__ w, int K, int cin, int cout,
                                                                int cinp, int coutp, __bf16* __restrict__ wp) {
    const int64_t total = (int64_t)K * cinp * coutp;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e & 7);
        int64_t t = e >> 3;
        const int col = (int)(t % coutp);
        t /= coutp;
        const int g = (int)(t % (cinp / 8));
        const int k = (int)(t / (cinp / 8));
        const int ci = 8 * g + j;
        wp[e] = (__bf16)((ci < cin && col < cout) ? w[((int64_t)k * cin + ci) * cout + col] : 0.0f);
    }
}

// Split-bf16 ("x3") packing: every fp32 weight is the EXACT sum of three bf16 numbers (24 significand bits = 8 + 8 + 8,
// by truncation: hi = top 16 bits of w, mid = top 16 bits of w - hi, lo = w - hi - mid).  Layout per (k, 32-channel chunk):
// [plane hi|mid|lo][group of 8 channels][col][8 bf16] — one linear LDS-DMA per step, each 16-B fragment one MFMA operand.
__device__ __forceinline__ void split_bf16x3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    const uint32_t xb = __float_as_uint(x);
    const uint32_t hb = xb & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(hb);
    const uint32_t mb = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mb);
    hi = (unsigned short)(hb >> 16);
    mid = (unsigned short)(mb >> 16);
    lo = (unsigned short)(__float_as_uint(r2) >> 16);
}

__global__ __launch_bounds__(256) void pack_weights_x3_kernel(const float* __restrict__ w, int K, int cin, int cout,
                                                              int cinp, int coutp, unsigned short* __restrict__ wp) {
    const int64_t total = (int64_t)K * cinp * coutp;              // one thread per weight, three outputs
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e & 7);
        int64_t t = e >> 3;
        const int col = (int)(t % coutp);
        t /= coutp;
        const int g = (int)(t & 3);
        t >>= 2;
        const int c = (int)(t % (cinp / 32));
        const int k = (int)(t / (cinp / 32));
        const int ci = 32 * c + 8 * g + j;
        const float v = (ci < cin && col < cout) ? w[((int64_t)k * cin + ci) * cout + col] : 0.0f;
        unsigned short hi, mid, lo;
        split_bf16x3(v, hi, mid, lo);
        const int64_t base = ((int64_t)k * (cinp / 32) + c) * 12;          // 12 (plane, group) blocks of coutp x 8
        wp[((base + 0 * 4 + g) * coutp + col) * 8 + j] = hi;
        wp[((base + 1 * 4 + g) * coutp + col) * 8 + j] = mid;
        wp[((base + 2 * 4 + g) * coutp + col) * 8 + j] = lo;
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA path
// ---------------------------------------------------------------------------------------------
constexpr int A_LD_DMA = 32;   // LDS-DMA image: unpadded 128-B rows, 16-B slots XOR-swizzled by (row >> 1) & 7

template <int BM, int BN, bool X3 = false>
constexpr int conv_lds_bytes() { return 2 * (BM * A_LD_DMA + (X3 ? 12 : 8) * BN * 4) * (int)sizeof(float); }

// 1 KB of zeros: the gather source of absent neighbours on the LDS-DMA path
__device__ float g_zero_line[256] = {0.0f};   // cin <= 256: the per-step channel offset stays inside

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// 64-bit-addressed variant (operands of 4 GiB and more; PCC_CONV_PATH=global): both operands go global ->
// LDS with global_load_lds_dwordx4 (no staging registers, no ds_write).  The A image is unpadded; each
// 16-B slot p of row R holds global chunk p ^ ((R >> 1) & 7), the permutation being applied on the
// per-lane SOURCE address (the LDS side of an LDS-DMA is lane-linear) and undone on the fragment reads.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool HAS_NBR>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int RPT = BM / 32;              // gather rows per thread (8 lanes per row)
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int A_LD = A_LD_DMA;
    constexpr int A_ELEMS = BM * A_LD;
    constexpr int W_ELEMS = 8 * BN * 4;
    constexpr int W_LOADS = (8 * BN) / 256;  // float4 per thread per chunk
    static_assert(WAVES_M * WAVES_N == 4 && MT >= 1 && NT >= 1 && W_LOADS >= 1, "bad tiling");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Ws = smem + 2 * A_ELEMS;

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wrow = (wid / WAVES_N) * WM, wcol = (wid % WAVES_N) * WN;

    const int ntiles_n = a.coutp / BN;
    const int64_t tile = blockIdx.x / ntiles_n;
    const int nt = blockIdx.x - (int)(tile * ntiles_n);
    const int64_t row0 = tile * BM;
    const int CCH = a.cin / 32;
    const int K = a.K;

    // which kernel offsets are live for this tile / for each of this wave's 32-row MFMA tiles
    uint32_t tmask, mmask[MT];
    {
        const uint32_t all = (K >= 32) ? 0xffffffffu : ((1u << K) - 1u);
        if (a.gmask) {
            const int64_t g0 = row0 >> 5;
            const int64_t ng = (a.n_out + 31) >> 5;
            tmask = 0u;
#pragma unroll
            for (int g = 0; g < BM / 32; ++g) tmask |= (g0 + g < ng) ? (a.gmask[g0 + g] & all) : 0u;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int64_t g = g0 + (wrow >> 5) + m;
                const uint32_t v = (g < ng) ? (a.gmask[g] & all) : 0u;
                mmask[m] = __builtin_amdgcn_readfirstlane(v);    // wave-uniform by construction
            }
        } else {
            tmask = all;
#pragma unroll
            for (int m = 0; m < MT; ++m) mmask[m] = all;
        }
        tmask = __builtin_amdgcn_readfirstlane(tmask);
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.0f;

    // gather roles: 8 lanes per row (16 B each).  Register path: rows grow + 32 i.  DMA path: one
    // wave-instruction fills 1 KB = 8 consecutive rows, wave w issues instructions 4w .. 4w+3.
    const int gchunk = t & 7;
    int grow[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) grow[i] = wid * (BM / 4) + 8 * i + (lane >> 3);
    int idx_cur[RPT], idx_nxt[RPT];
    // per-thread constants of the tile: which of my 4 gather rows exist, and where their nbr rows start.
    // Index loads are kept raw (no select on the loaded value) so that hipcc does not wait for them
    // at the point of issue; validity is folded in when the index is consumed.
    bool rvalid[RPT];
    const int32_t* nbr_row[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int64_t pos = row0 + grow[i];
        rvalid[i] = pos < a.n_out;
        const int64_t ps = rvalid[i] ? pos : a.n_out - 1;
        nbr_row[i] = HAS_NBR ? a.nbr + (a.order ? (int64_t)a.order[ps] : ps) * K : nullptr;
        idx_cur[i] = idx_nxt[i] = (int)ps;          // identity map when nbr == NULL (kernel_size 1)
    }

    auto load_idx = [&](int k, int (&dst)[RPT]) {     // unconditional loads: see the note above
        if (HAS_NBR) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) dst[i] = nbr_row[i][k];
        }
    };
    // ---- LDS-DMA path ----------------------------------------------------------------------------
    // Per-lane source row pointers (already at my swizzled 16-B chunk) are recomputed only when the
    // offset k changes, and the weight-slab lane offsets once per tile: the per-step address work is one
    // 64-bit add per DMA.  VALU issue is shared with the MFMAs of the co-resident waves, so the ~10
    // VALU instructions of a full address computation per DMA measurably slow the MFMA stream.
    const float* a_src[RPT];
    auto set_a_src = [&](const int (&idx)[RPT]) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const bool ok = rvalid[i] && idx[i] >= 0;
            const int q = gchunk ^ ((grow[i] >> 1) & 7);                  // global chunk for my LDS slot
            a_src[i] = (ok ? (a.fin + (int64_t)idx[i] * a.cin) : g_zero_line) + q * 4;
        }
    };
    int64_t w_lane_off[W_LOADS];
#pragma unroll
    for (int j = 0; j < W_LOADS; ++j) {
        const int f = t + 256 * j;
        const int g = f / BN, col = f - g * BN;
        w_lane_off[j] = ((int64_t)g * a.coutp + nt * BN + col) * 4;
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);               // provably wave-uniform LDS bases
    auto dma_step = [&](int k, int c, int buf) {
        {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            float* dst = As + buf * A_ELEMS + (wave_u * RPT + i) * 256;   // + lane * 16 B added by the hardware
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[i] + c * 32), (lds_ptr_t)dst, 16, 0, 0);
        }
        }
        const float* wbase = a.wp + ((int64_t)k * (a.cin / 4) + c * 8) * a.coutp * 4;     // scalar
#pragma unroll
        for (int j = 0; j < W_LOADS; ++j) {
            float* dst = Ws + buf * W_ELEMS + (wave_u * 64 + 256 * j) * 4;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wbase + w_lane_off[j]), (lds_ptr_t)dst, 16, 0, 0);
        }
    };

    // A-fragment addressing.  Lane (r, h) of a wave reads, for sub-block kk, the 16-B chunk 2 kk + h of
    // row wrow + 32 m + r.  DMA image: physical slot = chunk ^ ((r >> 1) & 7), i.e. bit 0 -> h ^ (sw & 1)
    // and bits 1-2 -> kk ^ (sw >> 1): four lane-constant offsets.
    const int sw = (r >> 1) & 7;
    int a_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        a_off[kk] = (((kk ^ (sw >> 1)) << 1) | (h ^ (sw & 1))) * 4;

    // MFMA block of one step, specialised at compile time on WHICH of the wave's 32-row tiles have
    // offset k (LIVE bit m = tile m).  Every variant is straight-line code: fragments of sub-block
    // kk+1 are fetched from LDS while the MFMAs of sub-block kk run, so the MFMA stream never waits
    // on an LDS round trip inside a step, also when only part of the wave's rows take the offset
    // (tiles on a boundary between two neighbour-mask groups).
    auto compute_live = [&](int buf, auto live_tag) {
        constexpr unsigned LIVE = decltype(live_tag)::value;
        const float* Ab = As + buf * A_ELEMS + (wrow + r) * A_LD;
        const float* Wb = Ws + buf * W_ELEMS + (h * BN + wcol + r) * 4;
        f32x4 av[2][MT], bv[2][NT];
        // MFMA-issuing waves outrank the waves busy with DMA issue / address math on the same SIMD (+3-4 %)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if ((LIVE >> m) & 1u) av[0][m] = *reinterpret_cast<const f32x4*>(Ab + 32 * m * A_LD + a_off[0]);
#pragma unroll
        for (int n = 0; n < NT; ++n) bv[0][n] = *reinterpret_cast<const f32x4*>(Wb + 32 * n * 4);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cb = kk & 1, nb = cb ^ 1;
            if (kk + 1 < 4) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if ((LIVE >> m) & 1u) av[nb][m] = *reinterpret_cast<const f32x4*>(Ab + 32 * m * A_LD + a_off[kk + 1]);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    bv[nb][n] = *reinterpret_cast<const f32x4*>(Wb + (2 * (kk + 1) * BN + 32 * n) * 4);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        if ((LIVE >> m) & 1u)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][m][s], bv[cb][n][s], acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto compute = [&](int buf, int k) {
        unsigned live = 0;
#pragma unroll
        for (int m = 0; m < MT; ++m) live |= ((mmask[m] >> k) & 1u) << m;      // wave-uniform (SGPR)
        if constexpr (MT == 1) {
            if (live) compute_live(buf, std::integral_constant<unsigned, 1u>{});
        } else {
            static_assert(MT == 2, "live-pattern dispatch written for MT <= 2");
            if (live == 3u) compute_live(buf, std::integral_constant<unsigned, 3u>{});
            else if (live == 1u) compute_live(buf, std::integral_constant<unsigned, 1u>{});
            else if (live == 2u) compute_live(buf, std::integral_constant<unsigned, 2u>{});
        }
    };

    if (tmask != 0u) {
        // Step sequence: live offsets k ascending, CCH channel chunks each.  Every global load in the loop
        // is unconditional and lands directly in loop-carried registers: a load under a branch (or a
        // select on a freshly loaded value) makes hipcc wait for it at the join, which would put the
        // L2 / HBM round trip of the neighbour indices in front of the MFMA stream.
        uint32_t rem = tmask;
        int k = __builtin_ctz(rem);
        rem &= rem - 1u;
        int knext = rem ? __builtin_ctz(rem) : -1;
        load_idx(k, idx_cur);
        set_a_src(idx_cur);
        dma_step(k, 0, 0);
        load_idx(knext >= 0 ? knext : k, idx_nxt);
        __syncthreads();        // (emits vmcnt(0): the first DMA / loads have landed)
        int c = 0, cur = 0;
        while (true) {
            int nk = k, nc = c + 1;
            if (nc == CCH) { nc = 0; nk = knext; }
            const bool has_next = nk >= 0;
            const bool advance = (nc == 0);
            if (advance) {
#pragma unroll
                for (int i = 0; i < RPT; ++i) idx_cur[i] = idx_nxt[i];     // values that arrived >= one step ago
                set_a_src(idx_cur);
            }
            if (has_next) {
                // buffer cur^1 was last read in the previous step, which every wave has left (barrier)
                dma_step(nk, nc, cur ^ 1);
            }
            // indices of the offset after next: issued a full step (or more) before their first use
            uint32_t rem2 = rem;
            int kn2 = knext;
            if (advance) { rem2 &= rem2 - 1u; kn2 = rem2 ? __builtin_ctz(rem2) : -1; }
            load_idx(kn2 >= 0 ? kn2 : k, idx_nxt);
            compute(cur, k);
            __syncthreads();     // vmcnt(0) + lgkmcnt(0) + barrier: next image complete
            if (!has_next) break;
            if (advance) { k = nk; rem = rem2; knext = kn2; }
            c = nc;
            cur ^= 1;
        }
    }

    // epilogue: D[row = (reg&3) + 8*(reg>>2) + 4*h][col = r] per 32x32 tile
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int col = nt * BN + wcol + 32 * n + r;
            if (col >= a.cout) continue;
            const float bcol = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int64_t pos = row0 + wrow + 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (pos >= a.n_out) continue;
                const int64_t row = a.order ? a.order[pos] : pos;
                float v = acc[m][n][reg] + bcol;
                if (a.film) {
                    const float* fr = a.film + row * (2 * (int64_t)a.cout);
                    v = v * fr[col] + fr[a.cout + col];
                }
                v = apply_act(v, a.act);
                if (a.residual) v += a.residual[row * a.cout + col];
                a.fout[row * a.cout + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA path, buffer-addressed (the default).  Same tiling, LDS image, accumulation order and
// epilogue as conv_mfma_kernel above — results are bit-identical — but the main loop is stripped of
// everything that competes with the MFMA stream for issue slots:
//   * gathers and weight slabs are `buffer_load_dwordx4 ... lds` with a lane-constant 32-bit VGPR
//     offset and a scalar per-step offset: no 64-bit VALU address per DMA.  Absent neighbours and
//     rows past n_out use an out-of-range offset; the buffer unit returns zeros for those lanes,
//     so no zero line and no select are needed;
//   * cin / 32 (CCH) is a template parameter: the steps of one kernel offset are straight-line code,
//     the LDS buffer parity of every step is a compile-time constant and all fragment addresses are
//     lane constants computed once per tile (zero VALU address work per step);
//   * neighbour indices are fetched once per offset (in the offset's last step, one offset ahead),
//     unconditionally: a load under a branch makes hipcc wait for it at the join.
// Measured on MI355X (tools/micro/mfma_buf.hip): the bare loop reaches 146-148 TFLOP/s of the
// 157 TFLOP/s fp32 MFMA peak; with 64-bit global LDS-DMA addressing it stops at 136-139.
// Limits: fin, nbr and the packed weights must each be < 4 GiB (32-bit buffer offsets); larger
// operands take conv_mfma_kernel.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t BUF_OOB = 0xFFFFF000u;       // voffset of lanes that must read zeros (>= num_records)
[[maybe_unused]] constexpr uint32_t BUF_FLAGS = 0x00020000u;     // raw buffer, 32-bit elements (gfx9 family word 3)

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// X3 (split-bf16 arithmetic on fp32 data, pcc_conv_fwd_x3): the A image is the fp32 one; after a lane's 8 consecutive
// channels of a row land in registers they are split exactly into three bf16 vectors (hi, mid, lo: and / subtract / and /
// subtract / pack), the weights come pre-split as three planes, and the six products whose weight is at least 2^-16 of the
// leading one — lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi, in that fixed order — go through v_mfma_f32_32x32x16_bf16
// with fp32 accumulation: 12 MFMAs of 8 passes per 32 x 32 x 32 block instead of 16 of 16 (3/8 of the matrix-pipe time),
// every bf16 product exact in fp32, the dropped terms below 3 x 2^-24 of |x||w| — the size of an fp32 rounding.
template <int BM, int BN, int WAVES_M, int WAVES_N, int CCH, bool HAS_NBR, bool BF16 = false, bool X3 = false>
__global__ __launch_bounds__(256) void conv_mfma_buf_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)           // the buffer builtins exist in the device pass only; the host pass needs just the stub
    constexpr int RPT = BM / 32;              // gather DMAs per thread and step (8 lanes per row)
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int A_ELEMS = BM * 32;          // unpadded 128-B rows, 16-B slots XOR-swizzled by (row >> 1) & 7
    constexpr int W_ELEMS = (X3 ? 12 : 8) * BN * 4;
    constexpr int W_LOADS = ((X3 ? 12 : 8) * BN) / 256;
    static_assert(WAVES_M * WAVES_N == 4 && MT >= 1 && MT <= 2 && NT >= 1 && W_LOADS >= 1, "bad tiling");
    static_assert(!(BF16 && X3), "one operand mode at a time");
    // BF16: the same byte images — a step is 64 bf16 channels (128 B per gathered row, 8 groups of 8 channels per
    // weight column) instead of 32 floats, and each 16-B fragment feeds ONE v_mfma_f32_32x32x16_bf16 (8 k-values
    // per lane) instead of four v_mfma_f32_32x32x2_f32.  CCH then counts 64-channel chunks.
    constexpr int ESZ = BF16 ? 2 : 4;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Ws = smem + 2 * A_ELEMS;

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wrow = (wid / WAVES_N) * WM, wcol = (wid % WAVES_N) * WN;

    const int ntiles_n = a.coutp / BN;
    const int64_t tile = blockIdx.x / ntiles_n;
    const int nt = blockIdx.x - (int)(tile * ntiles_n);
    const int64_t row0 = tile * BM;
    const int K = a.K;

    uint32_t tmask, mmask[MT];
    {
        const uint32_t all = (K >= 32) ? 0xffffffffu : ((1u << K) - 1u);
        if (a.gmask) {
            const int64_t g0 = row0 >> 5;
            const int64_t ng = (a.n_out + 31) >> 5;
            tmask = 0u;
#pragma unroll
            for (int g = 0; g < BM / 32; ++g) tmask |= (g0 + g < ng) ? (a.gmask[g0 + g] & all) : 0u;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int64_t g = g0 + (wrow >> 5) + m;
                const uint32_t v = (g < ng) ? (a.gmask[g] & all) : 0u;
                mmask[m] = __builtin_amdgcn_readfirstlane(v);
            }
        } else {
            tmask = all;
#pragma unroll
            for (int m = 0; m < MT; ++m) mmask[m] = all;
        }
        tmask = __builtin_amdgcn_readfirstlane(tmask);
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.0f;
        }

    if (tmask != 0u) {
        __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.fin), 0, (int)(uint32_t)(a.n_in * a.cin * ESZ), BUF_FLAGS);
        __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.wp), 0, (int)((uint32_t)K * a.cin * a.coutp * (X3 ? 6 : ESZ)), BUF_FLAGS);
        __amdgpu_buffer_rsrc_t rsrc_n = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int32_t*>(a.nbr), 0, HAS_NBR ? (int)(uint32_t)(a.n_out * K * 4) : 0, BUF_FLAGS);

        // gather roles: one wave-instruction fills 1 KB = 8 consecutive rows; wave w owns rows
        // [w BM/4, (w+1) BM/4) of the image; lane -> (row, 16-B slot)
        const int gchunk = t & 7;
        uint32_t q16[RPT], n_voff[RPT], a_voff[RPT];
        int idx_nxt[RPT];
        bool rvalid[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int grow = wid * (BM / 4) + 8 * i + (lane >> 3);
            const int64_t pos = row0 + grow;
            rvalid[i] = pos < a.n_out;
            const int64_t ps = rvalid[i] ? pos : a.n_out - 1;
            q16[i] = (uint32_t)((gchunk ^ ((grow >> 1) & 7)) * 16);     // global chunk held by my LDS slot
            n_voff[i] = (uint32_t)(((HAS_NBR && a.order) ? (int64_t)a.order[ps] : ps) * K * 4);      // the table is by output row
            idx_nxt[i] = (int)ps;                                        // identity map when nbr == NULL
        }
        uint32_t w_voff[W_LOADS];
#pragma unroll
        for (int j = 0; j < W_LOADS; ++j) {
            const int f = t + 256 * j;
            const int g = f / BN, col = f - g * BN;
            w_voff[j] = (uint32_t)((g * a.coutp + nt * BN + col) * 16);
        }
        const int wave_u = __builtin_amdgcn_readfirstlane(wid);
        const uint32_t w_kstride = X3 ? (uint32_t)(a.cin / 32) * 12u * a.coutp * 16u
                                      : (uint32_t)(a.cin * ESZ / 16) * a.coutp * 16;   // bytes per kernel offset (16-B channel groups)
        const uint32_t w_cstride = (uint32_t)(X3 ? 12 : 8) * a.coutp * 16;             // bytes per chunk
        const uint32_t a_row_bytes = (uint32_t)a.cin * ESZ;

        auto load_idx = [&](int k) {
            if constexpr (HAS_NBR) {
#pragma unroll
                for (int i = 0; i < RPT; ++i)
                    idx_nxt[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc_n, n_voff[i], k * 4, 0);
            }
        };
        auto set_src = [&](bool real) {        // idx_nxt -> byte offsets of my gather rows (or out of range -> zeros)
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const bool ok = real && rvalid[i] && idx_nxt[i] >= 0;
                a_voff[i] = ok ? (uint32_t)idx_nxt[i] * a_row_bytes + q16[i] : BUF_OOB;
            }
        };
        auto dma = [&](int k, auto cc, auto bufc) {
            constexpr int c = decltype(cc)::value;
            constexpr int buf = decltype(bufc)::value;
#pragma unroll
            for (int i = 0; i < RPT; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(As + buf * A_ELEMS + (wave_u * RPT + i) * 256), 16,
                                                         a_voff[i], c * 128, 0, 0);
            const uint32_t wso = (uint32_t)k * w_kstride + (uint32_t)c * w_cstride;
#pragma unroll
            for (int j = 0; j < W_LOADS; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(Ws + buf * W_ELEMS + (wave_u * 64 + 256 * j) * 4), 16,
                                                         w_voff[j], wso, 0, 0);
        };

        // lane-constant LDS byte addresses of my fragments in both buffers.  Lane (r, h) reads, for
        // sub-block kk, the 16-B chunk 2 kk + h of row wrow + 32 m + r; its slot is chunk ^ ((r >> 1) & 7).
        const int sw = (r >> 1) & 7;
        uint32_t a_addr[2][4], w_addr[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                a_addr[b][kk] = (uint32_t)((b * A_ELEMS + (wrow + r) * 32 + (((kk ^ (sw >> 1)) << 1) | (h ^ (sw & 1))) * 4) * 4);
            w_addr[b] = (uint32_t)((2 * A_ELEMS + b * W_ELEMS + (h * BN + wcol + r) * 4) * 4);
        }
        auto lds4 = [&](uint32_t addr) { return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(smem) + addr); };

        // X3: lane (r, h) needs, for k16-step q, channels 16 q + 8 h .. + 7 of its rows = 16-B slots 4 q + 2 h and + 1
        uint32_t a3_addr[2][2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    a3_addr[b][q][e] = (uint32_t)((b * A_ELEMS + (wrow + r) * 32 + ((4 * q + 2 * h + e) ^ sw) * 4) * 4);
        auto split8 = [&](const f32x4& lo4, const f32x4& hi4, bf16x8& ph, bf16x8& pm, bf16x8& pl) {
            uint32_t hb[8], mb[8], lb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = j < 4 ? lo4[j] : hi4[j - 4];
                hb[j] = __float_as_uint(x) & 0xFFFF0000u;
                const float r1 = x - __uint_as_float(hb[j]);
                mb[j] = __float_as_uint(r1) & 0xFFFF0000u;
                lb[j] = __float_as_uint(r1 - __uint_as_float(mb[j]));
            }
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 vh, vm, vl;
#pragma unroll
            for (int d = 0; d < 4; ++d) {          // element 2 d in the low half, 2 d + 1 in the high half
                vh[d] = __builtin_amdgcn_perm(hb[2 * d + 1], hb[2 * d], 0x07060302u);
                vm[d] = __builtin_amdgcn_perm(mb[2 * d + 1], mb[2 * d], 0x07060302u);
                vl[d] = __builtin_amdgcn_perm(lb[2 * d + 1], lb[2 * d], 0x07060302u);
            }
            ph = __builtin_bit_cast(bf16x8, vh);
            pm = __builtin_bit_cast(bf16x8, vm);
            pl = __builtin_bit_cast(bf16x8, vl);
        };
        auto compute_live_x3 = [&](auto bufc, auto live_tag) {
            constexpr int buf = decltype(bufc)::value;
            constexpr unsigned LIVE = decltype(live_tag)::value;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bf16x8 ah[MT], am[MT], al[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if ((LIVE >> m) & 1u)
                        split8(lds4(a3_addr[buf][q][0] + 32 * m * 32 * 4), lds4(a3_addr[buf][q][1] + 32 * m * 32 * 4), ah[m], am[m], al[m]);
                bf16x8 bh[NT], bm[NT], bl[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const uint32_t wb = w_addr[buf] + (uint32_t)((2 * q * BN + 32 * n) * 16);
                    bh[n] = __builtin_bit_cast(bf16x8, lds4(wb));
                    bm[n] = __builtin_bit_cast(bf16x8, lds4(wb + 4 * BN * 16));
                    bl[n] = __builtin_bit_cast(bf16x8, lds4(wb + 8 * BN * 16));
                }
#define PCC_X3_TERM(A, B)                                                                                                   \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int n = 0; n < NT; ++n) if ((LIVE >> m) & 1u)     \
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m], B[n], acc[m][n], 0, 0, 0)
                PCC_X3_TERM(al, bh);
                PCC_X3_TERM(ah, bl);
                PCC_X3_TERM(am, bm);
                PCC_X3_TERM(am, bh);
                PCC_X3_TERM(ah, bm);
                PCC_X3_TERM(ah, bh);
#undef PCC_X3_TERM
            }
            __builtin_amdgcn_s_setprio(0);
        };
        auto compute_live = [&](auto bufc, auto live_tag) {
            if constexpr (X3) {
                compute_live_x3(bufc, live_tag);
                return;
            }
            constexpr int buf = decltype(bufc)::value;
            constexpr unsigned LIVE = decltype(live_tag)::value;
            f32x4 av[2][MT], bv[2][NT];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if ((LIVE >> m) & 1u) av[0][m] = lds4(a_addr[buf][0] + 32 * m * 32 * 4);
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[0][n] = lds4(w_addr[buf] + 32 * n * 16);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cb = kk & 1, nb = cb ^ 1;
                if (kk + 1 < 4) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        if ((LIVE >> m) & 1u) av[nb][m] = lds4(a_addr[buf][kk + 1] + 32 * m * 32 * 4);
#pragma unroll
                    for (int n = 0; n < NT; ++n) bv[nb][n] = lds4(w_addr[buf] + (2 * (kk + 1) * BN + 32 * n) * 16);
                }
                if constexpr (BF16) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            if ((LIVE >> m) & 1u)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[cb][m]),
                                                                                    __builtin_bit_cast(bf16x8, bv[cb][n]), acc[m][n], 0, 0, 0);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n)
                                if ((LIVE >> m) & 1u)
                                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][m][s], bv[cb][n][s], acc[m][n], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        };
        auto compute = [&](auto bufc, unsigned live) {        // live: wave-uniform, bit m = 32-row tile m has this offset
            if constexpr (MT == 1) {
                if (live) compute_live(bufc, std::integral_constant<unsigned, 1u>{});
            } else {
                if (live == 3u) compute_live(bufc, std::integral_constant<unsigned, 3u>{});
                else if (live == 1u) compute_live(bufc, std::integral_constant<unsigned, 1u>{});
                else if (live == 2u) compute_live(bufc, std::integral_constant<unsigned, 2u>{});
            }
        };

        // One kernel offset = CCH steps of straight-line code.  Step c computes chunk c from buffer
        // (P + c) & 1 while the DMAs of the following step fill the other buffer; the offset's last
        // step starts the next live offset (its indices arrived one offset ago) and fetches the indices
        // of the one after.  When no offset follows, that step's gather is pointed out of range (no
        // traffic) so that the code stays branch-free; its image is never read.
        uint32_t rem = tmask;
        int k = __builtin_ctz(rem);
        rem &= rem - 1u;
        auto offset_body = [&](auto pc) {
            constexpr int P = decltype(pc)::value;
            const int knext = rem ? __builtin_ctz(rem) : -1;
            const uint32_t rem2 = rem & (rem - 1u);
            const int kn2 = rem2 ? __builtin_ctz(rem2) : (knext >= 0 ? knext : k);
            unsigned live = 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) live |= ((mmask[m] >> k) & 1u) << m;
            static_for<0, CCH>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int buf = (P + c) & 1;
                if constexpr (c + 1 < CCH) {
                    dma(k, std::integral_constant<int, c + 1>{}, std::integral_constant<int, buf ^ 1>{});
                } else {
                    set_src(knext >= 0);
                    dma(knext >= 0 ? knext : k, std::integral_constant<int, 0>{}, std::integral_constant<int, buf ^ 1>{});
                    load_idx(kn2);
                }
                compute(std::integral_constant<int, buf>{}, live);
                __syncthreads();       // vmcnt(0) + barrier: the next image is complete, this one is free
            });
            const bool more = knext >= 0;
            k = more ? knext : k;
            rem = rem2;
            return more;
        };

        load_idx(k);
        set_src(true);
        dma(k, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        load_idx(rem ? __builtin_ctz(rem) : k);
        __syncthreads();
        while (true) {
            if (!offset_body(std::integral_constant<int, 0>{})) break;
            if constexpr (CCH & 1) {
                if (!offset_body(std::integral_constant<int, 1>{})) break;
            }
        }
    }

    // epilogue: D[row = (reg&3) + 8*(reg>>2) + 4*h][col = r] per 32x32 tile
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int col = nt * BN + wcol + 32 * n + r;
                if (col >= a.cout) continue;
                const float bcol = a.bias ? a.bias[col] : 0.0f;
                const int64_t pos = row0 + wrow + 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (pos >= a.n_out) continue;
                const int64_t row = a.order ? a.order[pos] : pos;
                float v = acc[m][n][reg] + bcol;
                if (a.film) {
                    const float* fr = a.film + row * (2 * (int64_t)a.cout);
                    v = v * fr[col] + fr[a.cout + col];
                }
                v = apply_act(v, a.act);
                if (a.residual) v += a.residual[row * a.cout + col];
                a.fout[row * a.cout + col] = v;
            }
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Small launches (a few thousand rows and fewer: the coarse levels of every frame, every level of a small frame).
// With fewer workgroups than CUs the time of a launch is the time of ONE workgroup, and in the kernels above that is the
// serial chain of one wave's MFMAs: a 32 x 32 tile over K x cin contraction steps of v_mfma_f32_32x32x2_f32 (64 cycles per
// two channels) = 46 us for 27 x 128 channels at the peak clock, whatever the row count (measured: 66 us).  Here
//   * a wave owns ONE 16 x 16 block (v_mfma_f32_16x16x4_f32: 32 cycles per four channels, issued back to back on one
//     accumulator — tools/micro/mfma_chain_latency.hip — and the same fused multiply-add chain bit for bit as two chained
//     v_mfma_f32_32x32x2_f32: tools/micro/mfma_shapes_bitwise.hip, 0 of 5.1 M elements differ),
//     a workgroup 32 rows x 32 columns: four times the workgroups, every chain a quarter as long (12 us);
//   * nothing but the chain runs in the MFMA waves' instruction stream.  The fp32 MFMA shares its SIMD with the vector ALU
//     (any vector instruction between two MFMAs costs its issue time plus ~14 cycles, unhidden) and an LDS-DMA costs its
//     issuing wave 60-180 cycles: so four LOADER waves (one per SIMD, beside the MFMA wave) issue every DMA of the tile,
//     NS steps ahead in NS LDS stages, and keep the cursor over the tile's live offsets; an MFMA wave reads the next step's
//     operands from LDS between its MFMAs (ds_read_b128 with immediate offsets: no address arithmetic) and does its only
//     vector work — picking two floats of each 16-byte chunk, five LDS bases — in one burst per step;
//   * the neighbour indices of the tile's 32 rows are staged in LDS once, so every memory operation of a loader's loop is
//     an LDS-DMA and its counted s_waitcnt is exact; one bare s_barrier per step joins the two roles.
// A step = SC chunks of 32 channels of one live offset.  Same packed weights, same A-image swizzle, same accumulation
// order (offsets ascending, channels in MFMA order) and same epilogue as conv_mfma_buf_kernel: results are bit-identical
// (tests/test_conv_small.py).  In-kernel anatomy at 128 -> 128, 1,136 rows (s_memtime stamps): index staging 0.6 us,
// first operands 1.1, loop 23.5 (27 steps of 1,917 cycles: 1,024 of MFMA chain, ~290 picking, the rest LDS waits and the
// barrier), epilogue 1.0.
// ---------------------------------------------------------------------------------------------
template <int SC, int NS>
constexpr int conv_small_lds_bytes() { return (NS * SC * (32 * 32 + 8 * 32 * 4) + 32 * 32) * (int)sizeof(float); }

template <int SC, int NS>
__global__ __launch_bounds__(512) void conv_small_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int A_STAGE = SC * 32 * 32;          // floats: SC images of 32 rows x 32 channels (128-B rows, swizzled slots)
    constexpr int W_STAGE = SC * 8 * 32 * 4;       // floats: SC slabs of 8 channel groups x 32 columns x 4
    constexpr int STAGE = A_STAGE + W_STAGE;
    constexpr int DMAS = 2 * SC;                   // LDS-DMA instructions per loader thread and step
    static_assert(NS >= 2 && (NS - 1) * DMAS <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* idx_s = reinterpret_cast<int*>(smem + NS * STAGE);     // [32 rows][K] neighbour indices of my tile (-1 = absent)

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);  // 0-3: MFMA waves, 4-7: loader waves (one of each per SIMD)
    const int K = a.K;
    const int ntiles_n = a.coutp / 32;
    const int64_t tile = blockIdx.x / ntiles_n;
    const int nt = blockIdx.x - (int)(tile * ntiles_n);
    const int64_t row0 = tile * 32;
    uint32_t tmask = (K >= 32) ? 0xffffffffu : ((1u << K) - 1u);
    if (a.gmask) tmask &= a.gmask[row0 >> 5];
    tmask = __builtin_amdgcn_readfirstlane(tmask);
    const int CG = (a.cin / 32) / SC;                            // steps per offset
    const int total = __builtin_popcount(tmask) * CG;
    constexpr int WAIT_FIRST = (NS - 1) * DMAS, WAIT_LOOP = (NS - 2) * DMAS;

    if (tmask != 0u) {
        for (int e = t; e < 32 * K; e += 512) {
            const int lr = e / K;
            const int64_t pos = row0 + lr;
            idx_s[e] = pos < a.n_out ? a.nbr[(a.order ? (int64_t)a.order[pos] : pos) * K + (e - lr * K)] : -1;
        }
        __syncthreads();
    }
    // Barrier B_s, s = 0 .. total: step s has landed in LDS (every loader wave has waited for its share) and the MFMA waves
    // have their operands of step s - 1 out of LDS.  Both roles pass total + 1 of them.
    if (wave_u >= 4) {
        if (tmask == 0u) return;
        // ---- loader waves: issue every LDS-DMA of the tile (an LDS-DMA costs its wave 60-180 cycles of issue: in an MFMA
        // wave's stream that is time off the chain), NS steps ahead ----
        const int lw = wave_u - 4, tl = t - 256;
        __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fin), 0, (int)(uint32_t)(a.n_in * a.cin * 4), BUF_FLAGS);
        __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, (int)((uint32_t)K * a.cin * a.coutp * 4), BUF_FLAGS);
        // gather role: one wave-instruction fills 1 KB = 8 consecutive rows of an image; lane -> (row, 16-B slot)
        const int grow = 8 * lw + (lane >> 3);
        const uint32_t q16 = (uint32_t)(((tl & 7) ^ ((grow >> 1) & 7)) * 16);
        const uint32_t w_voff = (uint32_t)(((tl >> 5) * a.coutp + nt * 32 + (tl & 31)) * 16);
        const uint32_t w_kstride = (uint32_t)(a.cin / 4) * a.coutp * 16;     // bytes per kernel offset
        const uint32_t w_cstride = 8u * a.coutp * 16;                        // bytes per 32-channel chunk
        const uint32_t a_row_bytes = (uint32_t)a.cin * 4;
        // cursor (wave-uniform, in scalar registers: a vector-register offset would put every DMA in a readfirstlane loop).
        // Past the last step the DMAs are pointed out of range: zeros, no traffic, branch-free.
        uint32_t rem_i = tmask;
        int k_i = __builtin_ctz(rem_i), cg_i = 0, issued = 0, st_i = 0;
        rem_i &= rem_i - 1u;
        auto issue = [&]() {
            const bool real = issued < total;
            const int idxv = idx_s[grow * K + k_i];
            const uint32_t av = (real && idxv >= 0) ? (uint32_t)idxv * a_row_bytes + q16 : BUF_OOB;
            const uint32_t wv = real ? w_voff : BUF_OOB;
            const uint32_t aso = (uint32_t)__builtin_amdgcn_readfirstlane(cg_i * SC * 128);
            const uint32_t wso = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)k_i * w_kstride + (uint32_t)(cg_i * SC) * w_cstride));
            float* sb = smem + st_i * STAGE + lw * 256;
#pragma unroll
            for (int c = 0; c < SC; ++c)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sb + c * 1024), 16, av, aso + c * 128, 0, 0);
#pragma unroll
            for (int c = 0; c < SC; ++c)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + A_STAGE + c * 1024), 16, wv, wso + (uint32_t)c * w_cstride, 0, 0);
            st_i = (st_i + 1 == NS) ? 0 : st_i + 1;
            ++issued;
            const bool wrap = cg_i + 1 == CG;
            cg_i = wrap ? 0 : cg_i + 1;
            const bool adv = wrap && rem_i != 0u;
            k_i = __builtin_amdgcn_readfirstlane(adv ? __builtin_ctz(rem_i) : k_i);
            rem_i = __builtin_amdgcn_readfirstlane(adv ? (rem_i & (rem_i - 1u)) : rem_i);
        };
#pragma unroll
        for (int s = 0; s < NS; ++s) issue();
        __builtin_amdgcn_s_waitcnt((WAIT_FIRST & 15) | (7 << 4) | (15 << 8) | ((WAIT_FIRST >> 4) << 14));
        __builtin_amdgcn_s_barrier();                                         // B_0
        for (int s = 0; s < total; ++s) {
            // steps 0 .. s + NS - 1 are issued; all but the newest NS - 2 have landed = step s + 1 is complete
            __builtin_amdgcn_s_waitcnt((WAIT_LOOP & 15) | (7 << 4) | (15 << 8) | ((WAIT_LOOP >> 4) << 14));
            __builtin_amdgcn_s_barrier();                                     // B_(s+1): and step s's stage is free
            issue();                                                          // step s + NS into it
        }
        return;
    }

    // ---- MFMA waves ----
    const int wrow = 16 * (wave_u >> 1), wcol = 16 * (wave_u & 1);
    const int j16 = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (tmask != 0u) {
        // Fragments.  Lane (j16, kq) needs floats (kq >> 1) and (kq >> 1) + 2 of the 16-B chunk 2 kk + (kq & 1) of row
        // wrow + j16 (A) / of channel group 2 kk + (kq & 1), column wcol + j16 (W): the first feeds the sub-block's first MFMA,
        // the second its second.  Read as dwords these are 4-way bank conflicts (ds_read_b32 banks by address mod 128 B inside
        // 32-lane groups: 16 rows x 2 chunks land on 8 banks) and the LDS array, twice over, times the step.  So a lane reads
        // its whole 16-B chunk (ds_read_b128: groups of 16 lanes, 64 banks — conflict-free on this image, the lanes of the
        // upper half share the lower half's addresses) and picks its two floats afterwards.
        // Byte addresses: a per-stage base per pattern (four for A — the slot swizzle is an XOR — and one for W), everything
        // else in the instructions' immediate offsets.
        const int R = wrow + j16, sw16 = (R >> 1) & 7;
        uint32_t a_pat[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a_pat[kk] = (uint32_t)((R * 32 + (((2 * kk + (kq & 1)) ^ sw16) * 4)) * 4);
        const uint32_t w_pat = (uint32_t)((A_STAGE + ((kq & 1) * 32 + wcol + j16) * 4) * 4);
        uint32_t a_base[4], w_base;                                          // of the stage read next
        int st_c = 0;
        auto bases = [&]() {
            const uint32_t sb = (uint32_t)(st_c * STAGE * 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a_base[kk] = sb + a_pat[kk];
            w_base = sb + w_pat;
            st_c = (st_c + 1 == NS) ? 0 : st_c + 1;
        };
        f32x4 ra[SC * 4], rb[SC * 4];                                         // the next step's chunks, as read
        float fa0[SC * 4], fa1[SC * 4], fb0[SC * 4], fb1[SC * 4];             // this step's MFMA operands
        // (inline assembly: the immediate offset carries the chunk / sub-block displacement, so that no vector add sits
        // between the MFMAs.  The compiler does not count these reads in lgkmcnt: pick() waits for them explicitly.)
        auto load_sub = [&](auto qc) {                                        // sub-block q = 4 c + kk
            constexpr int q = decltype(qc)::value;
            constexpr int ao = (q >> 2) * 4096, wo = (q >> 2) * 4096 + (q & 3) * 1024;
            const uint32_t ab = a_base[q & 3], wb = w_base;
            f32x4 va, vb;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(va) : "v"(ab), "n"(ao));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vb) : "v"(wb), "n"(wo));
            ra[q] = va; rb[q] = vb;
        };
        // lanes 32-63 (kq >> 1 = 1) take floats 1 and 3 of a chunk, lanes 0-31 floats 0 and 2: one v_cndmask_b32 per operand
        // (written out: from "odd ? v[1] : v[0]" the compiler builds an indexed extract of three selects; volatile: behind the wait)
        const uint64_t upper = 0xFFFFFFFF00000000ull;
        auto sel = [&](float lo, float hi) {
            float r;
            asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(lo), "v"(hi), "s"(upper));
            return r;
        };
        auto pick = [&]() {
            __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0) (vmcnt, expcnt untouched)
            __builtin_amdgcn_sched_barrier(0);        // nothing that reads the chunks moves above the wait
#pragma unroll
            for (int q = 0; q < SC * 4; ++q) {
                fa0[q] = sel(ra[q][0], ra[q][1]); fa1[q] = sel(ra[q][2], ra[q][3]);
                fb0[q] = sel(rb[q][0], rb[q][1]); fb1[q] = sel(rb[q][2], rb[q][3]);
            }
        };
        auto mfma_sub = [&](int q) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[q], fb0[q], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[q], fb1[q], acc, 0, 0, 0);
        };
        // Step s: its operands are in registers.  The fp32 MFMA shares the SIMD with the vector ALU: a vector instruction
        // between two MFMAs of the chain costs its own issue time plus ~14 cycles, nothing of it hidden
        // (tools/micro/mfma_chain_latency.hip); LDS reads and scalar instructions are nearly free there.  So the MFMAs of a
        // step run as one chain with only those between them — the LDS reads of step s + 1's chunks, two sub-blocks per gap,
        // behind barrier B_(s+1) — and the vector work (LDS bases, picking the next operands) is one burst at its end.
        // sched_barrier pins that order.
        auto step = [&]() {
            mfma_sub(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            static_for<1, SC * 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (2 * (q - 1) < SC * 4) load_sub(std::integral_constant<int, 2 * (q - 1)>{});
                if constexpr (2 * (q - 1) + 1 < SC * 4) load_sub(std::integral_constant<int, 2 * (q - 1) + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                mfma_sub(q);
                __builtin_amdgcn_sched_barrier(0);
            });
            bases();
            pick();
            __builtin_amdgcn_sched_barrier(0);
        };

        __builtin_amdgcn_s_barrier();                                         // B_0
        bases();
        static_for<0, SC * 4>([&](auto qc) { load_sub(qc); });
        bases();
        pick();
        __builtin_amdgcn_sched_barrier(0);
        for (int s = 0; s < total; ++s) step();
    }

    // epilogue: register e = D[row 4 (lane >> 4) + e][column lane & 15] of my 16 x 16 block
    const int col = nt * 32 + wcol + j16;
    if (col < a.cout) {
        const float bcol = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t pos = row0 + wrow + 4 * kq + e;
            if (pos >= a.n_out) continue;
            const int64_t row = a.order ? a.order[pos] : pos;
            float v = acc[e] + bcol;
            if (a.film) {
                const float* fr = a.film + row * (2 * (int64_t)a.cout);
                v = v * fr[col] + fr[a.cout + col];
            }
            v = apply_act(v, a.act);
            if (a.residual) v += a.residual[row * a.cout + col];
            a.fout[row * a.cout + col] = v;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// thin path: cin <= 16 (q-map branches, input layers).  HBM/latency bound; W lives in LDS.
// One thread per (row, output channel); the gathered inputs are broadcast across the row's lanes.
// ---------------------------------------------------------------------------------------------
template <int CIN, int CPT>
__global__ __launch_bounds__(256) void conv_thin_kernel(const ConvArgs a) {
    // one thread = one output row x CPT consecutive output channels: the neighbour index and the CIN
    // input values are loaded once per offset and reused for CPT FMAs (W rows come from LDS)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = a.K, cout = a.cout;
    const int wtotal = K * CIN * cout;
    for (int e = threadIdx.x; e < wtotal; e += 256) smem[e] = a.w[e];
    __syncthreads();
    const int groups = cout / CPT;
    const int64_t total = a.n_out * groups;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / groups;
        const int co = (int)(e - row * groups) * CPT;
        float acc[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[j] = 0.0f;
        // Offsets in blocks of 9 (one z-slice): the 9 indices are loaded first, then the 9 gathers — independent
        // loads in flight instead of 27 dependent index -> value round trips per row.  The FMAs keep their order
        // (k ascending) and absent neighbours are still skipped, so results are unchanged bit for bit.
        for (int k0 = 0; k0 < K; k0 += 9) {
            int idx[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) idx[u] = (k0 + u < K) ? (a.nbr ? a.nbr[row * K + k0 + u] : (int)row) : -1;
            float in[9][CIN];
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const int64_t src = idx[u] >= 0 ? idx[u] : 0;          // row 0 always exists; its values are not used
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) in[u][ci] = a.fin[src * CIN + ci];
            }
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                if (idx[u] < 0) continue;
                const float* wk = smem + ((k0 + u) * CIN) * cout + co;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                    for (int j = 0; j < CPT; ++j) acc[j] = fmaf(in[u][ci], wk[ci * cout + j], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            float v = acc[j] + (a.bias ? a.bias[co + j] : 0.0f);
            if (a.film) {
                const float* fr = a.film + row * (2 * (int64_t)cout);
                v = v * fr[co + j] + fr[cout + co + j];
            }
            v = apply_act(v, a.act);
            if (a.residual) v += a.residual[row * cout + co + j];
            a.fout[row * cout + co + j] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// thin inputs, WIDE outputs (cin <= 8, cout a multiple of 32: 2 -> 128 and 4 -> 64, the first layers of the q-map heads and
// of g_a) as a dense product on the matrix cores.  conv_thin_kernel gives a thread 8 output channels of one row: the 16
// threads of a 128-wide row each load its 27 indices and 54 inputs and fetch every weight from LDS once per 4
// multiply-adds — 0.27-0.45 ms for layers that write 136-218 MB (0.5-1 TB/s; round-3 profiles).  But such a layer is a
// GEMM with a short inner dimension, [rows x K cin] x [K cin x cout], and the fp32 MFMA is the same fused multiply-add chain
// as v_fma_f32, bit for bit (tools/micro/mfma_shapes_bitwise.hip): im2col_thin_kernel gathers a row's K cin inputs (zeros
// for absent neighbours, zero padding up to a multiple of 32) into a dense matrix, and the kernel_size-1 instance of
// conv_mfma_buf_kernel multiplies it with the re-laid-out weights.  The columns are stored in the order the MFMA loop
// contracts them — within 8 channels it visits 0, 4, 1, 5, 2, 6, 3, 7 — so that per output element the products are added
// for (k, ci) ascending exactly as conv_thin_kernel adds them (an absent neighbour adds fma(0, w, acc) = acc; the
// accumulator starts at +0): same bits as the scalar kernel (tests/test_hip_parity.py).
// (Two scalar-pipe attempts of round 3 — a lane per row with the weights as scalar operands, then as broadcast LDS reads —
// were bound by scalar-load latency and LDS issue: 0.17-0.57 ms.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_thin_kernel(const float* __restrict__ fin, int cin, const int32_t* __restrict__ nbr,
                                                          int64_t n_out, int K, float* __restrict__ out, int k2) {
    const int q = k2 >> 2;                                   // float4 pieces per row
    const int64_t total = n_out * q;
    const int kc = K * cin;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / q;
        const int c0 = (int)(e - row * q) * 4;               // physical columns c0 .. c0 + 3
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = c0 + s;
            // physical column 8 g + s8 holds the logical column the MFMA loop visits at position inv[s8] of the group
            const int s8 = c & 7;
            const int logical = (c & ~7) + ((s8 & 3) << 1 | (s8 >> 2));
            float x = 0.0f;
            if (logical < kc) {
                const int k = logical / cin, ci = logical - k * cin;
                const int idx = nbr[row * K + k];
                if (idx >= 0) x = fin[(int64_t)idx * cin + ci];
            }
            v[s] = x;
        }
        *reinterpret_cast<f32x4*>(out + row * k2 + c0) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// narrow heads (cout <= 4 with wide inputs: the occupancy logit, q-map outputs).  Padding cout to a
// 32-wide MFMA tile would gather every neighbour row (cin * 4 bytes) to produce a handful of
// numbers.  Instead: scores[i, k*cout + c] = in[i] . W[k][:, c] is one dense GEMM over the INPUT rows
// (no gather; run through the MFMA kernel as a kernel_size-1 convolution), and this kernel adds
// the <= 27 scalars each output row needs:  out[j, c] = bias[c] + sum_k scores[nbr(j,k), k*cout + c].
// HBM-bound: one 4-byte read per (pair, c) instead of cin * 4 bytes per pair.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_sum_kernel(const float* __restrict__ scores, int ld,
                                                         const int32_t* __restrict__ nbr, int K, int cout,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int64_t n_out, int act) {
    const int64_t total = n_out * cout;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / cout;
        const int c = (int)(e - row * cout);
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) {
            const int idx = nbr[row * K + k];
            if (idx >= 0) acc += scores[(int64_t)idx * ld + k * cout + c];
        }
        out[e] = apply_act(acc + (bias ? bias[c] : 0.0f), act);
    }
}

// K = 27 form of the same sum with the memory operations batched.  A block owns 256 consecutive output rows: their
// 256 x 27 neighbour indices (27,648 contiguous bytes) are staged through LDS with 16-byte coalesced loads (a lane
// reading its own 108-byte row from global touches a new line almost every step), then each thread walks its row in
// z-slices of 9 offsets — 9 independent score gathers in flight instead of 27 index -> score round trips.  Absent
// neighbours are still skipped and k still ascends, so the sum is the one gather_sum_kernel computes, bit for bit.
template <int COUT>
__global__ __launch_bounds__(256) void gather_sum27_kernel(const float* __restrict__ scores, int ld,
                                                           const int32_t* __restrict__ nbr, const float* __restrict__ bias,
                                                           float* __restrict__ out, int64_t n_out, int act) {
    __shared__ __attribute__((aligned(16))) int32_t idx_s[256 * 27];
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    const int64_t rows = (n_out - row0 < 256) ? (n_out - row0) : 256;
    const int64_t words = rows * 27;
    const int32_t* src = nbr + row0 * 27;                 // 27,648 B per block: 16-byte aligned
    for (int64_t w = 4 * (int64_t)threadIdx.x; w < words; w += 1024) {
        if (w + 4 <= words) {
            *reinterpret_cast<int4*>(idx_s + w) = *reinterpret_cast<const int4*>(src + w);
        } else {
            for (int64_t u = w; u < words; ++u) idx_s[u] = src[u];
        }
    }
    __syncthreads();
    if ((int64_t)threadIdx.x >= rows) return;
    const int32_t* mine = idx_s + threadIdx.x * 27;       // stride 27 words: conflict-free
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int k0 = 0; k0 < 27; k0 += 9) {
        int idx[9];
        float v[9][COUT];
#pragma unroll
        for (int u = 0; u < 9; ++u) idx[u] = mine[k0 + u];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const float* sp = scores + (int64_t)(idx[u] >= 0 ? idx[u] : 0) * ld + (k0 + u) * COUT;   // row 0 exists; unused
#pragma unroll
            for (int c = 0; c < COUT; ++c) v[u][c] = sp[c];
        }
#pragma unroll
        for (int u = 0; u < 9; ++u)
            if (idx[u] >= 0) {
#pragma unroll
                for (int c = 0; c < COUT; ++c) acc[c] += v[u][c];
            }
    }
    float* o = out + (row0 + threadIdx.x) * COUT;
#pragma unroll
    for (int c = 0; c < COUT; ++c) o[c] = apply_act(acc[c] + (bias ? bias[c] : 0.0f), act);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool HAS_NBR>
static int launch_mfma_impl(const ConvArgs& a, hipStream_t st) {
    static bool attr_set = false;
    auto kern = conv_mfma_kernel<BM, BN, WAVES_M, WAVES_N, HAS_NBR>;
    const int lds = conv_lds_bytes<BM, BN>();
    if (!attr_set) {
        PCC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const int64_t tiles = (a.n_out + BM - 1) / BM;
    const int64_t blocks = tiles * (a.coutp / BN);
    PCC_REQUIRE(blocks < (1ll << 31), "conv: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, a);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int CCH, bool HAS_NBR, bool BF16 = false, bool X3 = false>
static int launch_mfma_buf_impl(const ConvArgs& a, hipStream_t st) {
    static bool attr_set = false;
    auto kern = conv_mfma_buf_kernel<BM, BN, WAVES_M, WAVES_N, CCH, HAS_NBR, BF16, X3>;
    const int lds = conv_lds_bytes<BM, BN, X3>();
    if (!attr_set) {
        PCC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const int64_t tiles = (a.n_out + BM - 1) / BM;
    const int64_t blocks = tiles * (a.coutp / BN);
    PCC_REQUIRE(blocks < (1ll << 31), "conv: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, a);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

// 32-bit buffer offsets: every operand the buffer path addresses must stay below BUF_OOB bytes
static bool fits_buffer_path(const ConvArgs& a) {
    const uint64_t lim = BUF_OOB;
    return (uint64_t)a.n_in * a.cin * 4 <= lim && (uint64_t)a.n_out * a.K * 4 <= lim && (uint64_t)a.K * a.cin * a.coutp * 4 <= lim;
}

// PCC_CONV_PATH=global forces the 64-bit-addressed kernel (testing; the >= 4 GiB fallback)
static int conv_path() {
    static int path = -1;
    if (path < 0) { const char* e = getenv("PCC_CONV_PATH"); path = (e && e[0] == 'g') ? 1 : 0; }
    return path;
}

template <int SC, int NS>
static int launch_small_impl(const ConvArgs& a, hipStream_t st) {
    static bool attr_set = false;
    auto kern = conv_small_kernel<SC, NS>;
    const int lds = conv_small_lds_bytes<SC, NS>();
    if (!attr_set) {
        PCC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const int64_t blocks = ((a.n_out + 31) / 32) * (a.coutp / 32);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, st, a);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

// The small-launch kernel takes a map convolution of at most PCC_CONV_SMALL_MAX (default 640; 0 = never) 32 x 32 output
// tiles — twice that for outputs narrower than 128 columns, whose ordinary kernel (128-row tiles) fills the chip later.
// Measured on MI355X (tools/conv_small_bench.py, one launch): 128 -> 128 on 56 .. 1,136 rows 27 us against 66, 64 -> 64
// 16.5 against 72, 64 -> 128 17 against 38; even at 616 tiles (4,904 rows x 128 columns: 59 against 66; 19,256 rows x 64
// columns = 1,204 tiles: 62 against 74), behind from 1,000 tiles on (8,000 rows x 128: 90 against 66).
static int64_t g_small_max = -1;
static int64_t small_max_value() {
    if (g_small_max < 0) { const char* e = getenv("PCC_CONV_SMALL_MAX"); g_small_max = e ? atoll(e) : 640; }
    return g_small_max;
}
static bool small_launch(const ConvArgs& a) {
    const int64_t small_max = small_max_value() * (a.coutp % 128 == 0 ? 1 : 2);
    return a.nbr && conv_path() == 0 && a.cin % 32 == 0 && a.coutp % 32 == 0 && ((a.n_out + 31) / 32) * (a.coutp / 32) <= small_max;
}

static int launch_small(const ConvArgs& a, hipStream_t st) {
    const int cch = a.cin / 32;
    // a step of four chunks (27 barriers for 128 channels instead of 54) while one workgroup per CU — 100 KB of LDS — holds
    // the launch; else two chunks per step at 52 KB (three workgroups per CU); odd chunk counts one chunk per step
    const int64_t wgs = ((a.n_out + 31) / 32) * (a.coutp / 32);
    if (cch % 4 == 0 && wgs <= 256) return launch_small_impl<4, 3>(a, st);
    return cch % 2 == 0 ? launch_small_impl<2, 3>(a, st) : launch_small_impl<1, 8>(a, st);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_mfma(const ConvArgs& a, hipStream_t st) {
    const int path = conv_path();
    if (path == 0 && fits_buffer_path(a)) {
#define PCC_BUF_CASE(C)                                                                                        \
    case C:                                                                                                    \
        return a.nbr ? launch_mfma_buf_impl<BM, BN, WAVES_M, WAVES_N, C, true>(a, st)                          \
                     : launch_mfma_buf_impl<BM, BN, WAVES_M, WAVES_N, C, false>(a, st);
        switch (a.cin / 32) {
            PCC_BUF_CASE(1) PCC_BUF_CASE(2) PCC_BUF_CASE(3) PCC_BUF_CASE(4)
            PCC_BUF_CASE(5) PCC_BUF_CASE(6) PCC_BUF_CASE(7) PCC_BUF_CASE(8)
            default: break;
        }
#undef PCC_BUF_CASE
    }
    return a.nbr ? launch_mfma_impl<BM, BN, WAVES_M, WAVES_N, true>(a, st) : launch_mfma_impl<BM, BN, WAVES_M, WAVES_N, false>(a, st);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_mfma_bf16(const ConvArgs& a, hipStream_t st) {
#define PCC_BF16_CASE(C)                                                                                       \
    case C:                                                                                                    \
        return a.nbr ? launch_mfma_buf_impl<BM, BN, WAVES_M, WAVES_N, C, true, true>(a, st)                    \
                     : launch_mfma_buf_impl<BM, BN, WAVES_M, WAVES_N, C, false, true>(a, st);
    switch (a.cin / 64) {
        PCC_BF16_CASE(1) PCC_BF16_CASE(2) PCC_BF16_CASE(3) PCC_BF16_CASE(4)
        default: break;
    }
#undef PCC_BF16_CASE
    pcc::set_error("pcc_conv_fwd_bf16: cin=%d not supported (multiples of 64 up to 256)", a.cin);
    return PCC_ERR_UNSUPPORTED;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_mfma_x3(const ConvArgs& a, hipStream_t st) {
#define PCC_X3_CASE(C)                                                                                         \
    case C:                                                                                                    \
        return a.nbr ? launch_mfma_buf_impl<BM, BN, WAVES_M, WAVES_N, C, true, false, true>(a, st)             \
                     : launch_mfma_buf_impl<BM, BN, WAVES_M, WAVES_N, C, false, false, true>(a, st);
    switch (a.cin / 32) {
        PCC_X3_CASE(1) PCC_X3_CASE(2) PCC_X3_CASE(3) PCC_X3_CASE(4)
        PCC_X3_CASE(5) PCC_X3_CASE(6) PCC_X3_CASE(7) PCC_X3_CASE(8)
        default: break;
    }
#undef PCC_X3_CASE
    pcc::set_error("pcc_conv_fwd_x3: cin=%d not supported (multiples of 32 up to 256)", a.cin);
    return PCC_ERR_UNSUPPORTED;
}

template <int CIN, int CPT>
static int launch_thin_cpt(const ConvArgs& a, unsigned nb, size_t lds, hipStream_t st) {
    auto kern = conv_thin_kernel<CIN, CPT>;
    if (lds > 64 * 1024) {                        // beyond the default dynamic-LDS limit (24 -> 32 with 27 offsets: 81 KB)
        static bool attr_set = false;
        if (!attr_set) {
            PCC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, st, a);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

template <int CIN>
static int launch_thin(const ConvArgs& a, hipStream_t st) {
    const size_t lds = (size_t)a.K * CIN * a.cout * sizeof(float);
    PCC_REQUIRE(lds <= 160 * 1024, "conv(thin): weights %zu B exceed the 160 KB of LDS (cin=%d cout=%d K=%d)", lds, CIN, a.cout, a.K);
    // channels per thread: the largest of 8, 4, 2, 1 dividing cout
    const int cpt = (a.cout % 8 == 0) ? 8 : (a.cout % 4 == 0) ? 4 : (a.cout % 2 == 0) ? 2 : 1;
    // every block stages the whole weight tensor into LDS first (27.6 KB for 2 -> 128): a few resident blocks per CU
    // that stride over the rows, not one block per 256 outputs
    const unsigned nb = blocks_for(a.n_out * (a.cout / cpt), 256, lds >= 4096 ? 2048u : (1u << 20));
    if (cpt == 8) return launch_thin_cpt<CIN, 8>(a, nb, lds, st);
    if (cpt == 4) return launch_thin_cpt<CIN, 4>(a, nb, lds, st);
    if (cpt == 2) return launch_thin_cpt<CIN, 2>(a, nb, lds, st);
    return launch_thin_cpt<CIN, 1>(a, nb, lds, st);
}

}  // namespace pcc

using namespace pcc;

static inline int round_up32(int v) { return (v + 31) / 32 * 32; }

extern "C" {

int64_t pcc_conv_packed_elems(int32_t K, int32_t cin, int32_t cout) {
    return (int64_t)K * round_up32(cin) * round_up32(cout);
}

int pcc_conv_pack_weights(const float* w, int32_t K, int32_t cin, int32_t cout, float* w_packed, void* stream) {
    PCC_REQUIRE(K >= 1 && cin >= 1 && cout >= 1, "pcc_conv_pack_weights: bad shape");
    const int cinp = round_up32(cin), coutp = round_up32(cout);
    const int64_t total = (int64_t)K * cinp * coutp;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks_for(total, 256, 4096)), dim3(256), 0, as_stream(stream), w, K, cin,
                       cout, cinp, coutp, w_packed);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int64_t pcc_conv_packed_elems_bf16(int32_t K, int32_t cin, int32_t cout) {
    return (int64_t)K * ((cin + 63) / 64 * 64) * round_up32(cout);
}

int pcc_conv_pack_weights_bf16(const float* w, int32_t K, int32_t cin, int32_t cout, uint16_t* w_packed, void* stream) {
    PCC_REQUIRE(K >= 1 && cin >= 1 && cout >= 1, "pcc_conv_pack_weights_bf16: bad shape");
    const int cinp = (cin + 63) / 64 * 64, coutp = round_up32(cout);
    const int64_t total = (int64_t)K * cinp * coutp;
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(blocks_for(total, 256, 4096)), dim3(256), 0, as_stream(stream), w, K, cin,
                       cout, cinp, coutp, reinterpret_cast<__bf16*>(w_packed));
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_conv_fwd_bf16(const uint16_t* fin, int64_t n_in, int32_t cin, const uint16_t* w_packed, const float* bias, const int32_t* nbr,
                      const int32_t* order, const uint32_t* group_mask32, int32_t K, float* fout, int64_t n_out, int32_t cout,
                      int32_t act, const float* film, const float* residual, void* stream) {
    PCC_REQUIRE(K >= 1 && K <= 27, "pcc_conv_fwd_bf16: K=%d out of range", K);
    PCC_REQUIRE(cin % 64 == 0 && cin <= 256, "pcc_conv_fwd_bf16: cin must be a multiple of 64 up to 256 (got %d)", cin);
    PCC_REQUIRE(nbr != nullptr || (K == 1 && n_in == n_out), "pcc_conv_fwd_bf16: nbr == NULL needs K == 1 and n_in == n_out");
    PCC_REQUIRE(act >= 0 && act <= 2, "pcc_conv_fwd_bf16: bad activation %d", act);
    PCC_REQUIRE(w_packed != nullptr, "pcc_conv_fwd_bf16: packed weights required");
    PCC_REQUIRE(((reinterpret_cast<uintptr_t>(fin) | reinterpret_cast<uintptr_t>(w_packed)) & 15) == 0,
                "pcc_conv_fwd_bf16: fin and w_packed must be 16-byte aligned (16-byte LDS-DMA loads)");
    if (n_out <= 0) return PCC_OK;
    ConvArgs a;
    a.fin = reinterpret_cast<const float*>(fin); a.w = nullptr; a.wp = reinterpret_cast<const float*>(w_packed); a.bias = bias;
    a.nbr = nbr; a.order = order; a.gmask = group_mask32; a.fout = fout; a.film = film; a.residual = residual;
    a.n_in = n_in; a.n_out = n_out; a.cin = cin; a.cout = cout; a.coutp = round_up32(cout); a.K = K; a.act = act; a.bf16 = 1;
    PCC_REQUIRE((uint64_t)n_in * cin * 2 <= BUF_OOB && (uint64_t)n_out * K * 4 <= BUF_OOB,
                "pcc_conv_fwd_bf16: operands of 4 GiB and more are not supported");
    hipStream_t st = as_stream(stream);
    const int64_t wgs128 = ((a.n_out + 63) / 64) * (a.coutp / 128);
    if (a.coutp % 128 == 0 && wgs128 < 768) return launch_mfma_bf16<64, 64, 2, 2>(a, st);
    if (a.coutp % 128 == 0) return launch_mfma_bf16<64, 128, 2, 2>(a, st);
    if (a.coutp % 64 == 0) return launch_mfma_bf16<128, 64, 2, 2>(a, st);
    return launch_mfma_bf16<128, 32, 4, 1>(a, st);
}

int64_t pcc_conv_packed_elems_x3(int32_t K, int32_t cin, int32_t cout) {
    return (int64_t)K * round_up32(cin) * round_up32(cout) * 3;
}

int pcc_conv_pack_weights_x3(const float* w, int32_t K, int32_t cin, int32_t cout, uint16_t* w_packed, void* stream) {
    PCC_REQUIRE(K >= 1 && cin >= 1 && cout >= 1, "pcc_conv_pack_weights_x3: bad shape");
    const int cinp = round_up32(cin), coutp = round_up32(cout);
    const int64_t total = (int64_t)K * cinp * coutp;
    hipLaunchKernelGGL(pack_weights_x3_kernel, dim3(blocks_for(total, 256, 4096)), dim3(256), 0, as_stream(stream), w, K, cin,
                       cout, cinp, coutp, w_packed);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_conv_fwd_x3(const float* fin, int64_t n_in, int32_t cin, const uint16_t* w_packed, const float* bias, const int32_t* nbr,
                    const int32_t* order, const uint32_t* group_mask32, int32_t K, float* fout, int64_t n_out, int32_t cout,
                    int32_t act, const float* film, const float* residual, void* stream) {
    PCC_REQUIRE(K >= 1 && K <= 27, "pcc_conv_fwd_x3: K=%d out of range", K);
    PCC_REQUIRE(cin % 32 == 0 && cin <= 256, "pcc_conv_fwd_x3: cin must be a multiple of 32 up to 256 (got %d)", cin);
    PCC_REQUIRE(nbr != nullptr || (K == 1 && n_in == n_out), "pcc_conv_fwd_x3: nbr == NULL needs K == 1 and n_in == n_out");
    PCC_REQUIRE(act >= 0 && act <= 2, "pcc_conv_fwd_x3: bad activation %d", act);
    PCC_REQUIRE(w_packed != nullptr, "pcc_conv_fwd_x3: packed weights required");
    PCC_REQUIRE(((reinterpret_cast<uintptr_t>(fin) | reinterpret_cast<uintptr_t>(w_packed)) & 15) == 0,
                "pcc_conv_fwd_x3: fin and w_packed must be 16-byte aligned (16-byte LDS-DMA loads)");
    if (n_out <= 0) return PCC_OK;
    ConvArgs a;
    a.fin = fin; a.w = nullptr; a.wp = reinterpret_cast<const float*>(w_packed); a.bias = bias;
    a.nbr = nbr; a.order = order; a.gmask = group_mask32; a.fout = fout; a.film = film; a.residual = residual;
    a.n_in = n_in; a.n_out = n_out; a.cin = cin; a.cout = cout; a.coutp = round_up32(cout); a.K = K; a.act = act; a.bf16 = 0;
    PCC_REQUIRE((uint64_t)n_in * cin * 4 <= BUF_OOB && (uint64_t)n_out * K * 4 <= BUF_OOB && (uint64_t)K * cin * a.coutp * 6 <= BUF_OOB,
                "pcc_conv_fwd_x3: operands of 4 GiB and more are not supported");
    hipStream_t st = as_stream(stream);
    // Tile shapes, measured on the config-2 frame: with the matrix-pipe time at 3/8 the loop waits for its gathers, so
    // occupancy decides — the three-plane weight slab makes a 64 x 128 tile 64 KB of LDS (two workgroups per CU) and a
    // 64 x 64 tile 40 KB (four).  64-wide outputs: 64 x 64 tiles 5.2 ms against 5.75 for 128 x 64 on the 5.16 M-row layers;
    // 128-wide outputs: 64 x 128 tiles 4.95-5.15 ms against 5.2-5.35 for 64 x 64 (which gathers every row twice).
    const int64_t wgs128 = ((a.n_out + 63) / 64) * (a.coutp / 128);
    if (a.coutp % 128 == 0 && wgs128 < 768) return launch_mfma_x3<64, 64, 2, 2>(a, st);
    if (a.coutp % 128 == 0) return launch_mfma_x3<64, 128, 2, 2>(a, st);
    if (a.coutp % 64 == 0) return launch_mfma_x3<64, 64, 2, 2>(a, st);
    pcc::set_error("pcc_conv_fwd_x3: cout=%d not supported (output width rounded up to 32 must be a multiple of 64)", cout);
    return PCC_ERR_UNSUPPORTED;
}

int pcc_im2col_thin(const float* fin, int32_t cin, const int32_t* nbr, int64_t n_out, int32_t K, float* out, int32_t k2,
                    void* stream) {
    PCC_REQUIRE(K >= 1 && K <= 27 && cin >= 1 && cin <= 16, "pcc_im2col_thin: bad shape (K=%d cin=%d)", K, cin);
    PCC_REQUIRE(k2 % 32 == 0 && k2 >= K * cin, "pcc_im2col_thin: row length %d must be a multiple of 32 and >= K * cin = %d", k2, K * cin);
    PCC_REQUIRE(nbr != nullptr && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "pcc_im2col_thin: nbr required, out 16-byte aligned");
    if (n_out <= 0) return PCC_OK;
    hipLaunchKernelGGL(im2col_thin_kernel, dim3(blocks_for(n_out * (k2 / 4), 256, 1 << 20)), dim3(256), 0, as_stream(stream), fin, cin, nbr,
                       n_out, K, out, k2);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_gather_sum_fwd(const float* scores, int32_t ld, const int32_t* nbr, int32_t K, int32_t cout, const float* bias,
                       float* out, int64_t n_out, int32_t act, void* stream) {
    PCC_REQUIRE(K >= 1 && K <= 27 && cout >= 1 && ld >= K * cout, "pcc_gather_sum_fwd: bad shape (K=%d cout=%d ld=%d)", K, cout, ld);
    PCC_REQUIRE(nbr != nullptr, "pcc_gather_sum_fwd: neighbour table required");
    PCC_REQUIRE(act >= 0 && act <= 2, "pcc_gather_sum_fwd: bad activation %d", act);
    if (n_out <= 0) return PCC_OK;
    const unsigned nb27 = blocks_for(n_out, 256);
    const bool al16 = (reinterpret_cast<uintptr_t>(nbr) & 15) == 0;      // the K = 27 kernel stages indices with 16-byte loads
#define PCC_GS27(C)                                                                                                    \
    hipLaunchKernelGGL(gather_sum27_kernel<C>, dim3(nb27), dim3(256), 0, as_stream(stream), scores, ld, nbr, bias, out, n_out, act)
    if (K == 27 && al16 && cout == 1) PCC_GS27(1);
    else if (K == 27 && al16 && cout == 2) PCC_GS27(2);
    else if (K == 27 && al16 && cout == 3) PCC_GS27(3);
    else if (K == 27 && al16 && cout == 4) PCC_GS27(4);
    else
#undef PCC_GS27
        hipLaunchKernelGGL(gather_sum_kernel, dim3(blocks_for(n_out * cout, 256, 1 << 20)), dim3(256), 0, as_stream(stream),
                           scores, ld, nbr, K, cout, bias, out, n_out, act);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int64_t pcc_conv_small_max(int64_t workgroups) {
    const int64_t before = small_max_value();
    if (workgroups >= 0) g_small_max = workgroups;
    return before;
}

int pcc_conv_fwd(const float* fin, int64_t n_in, int32_t cin, const float* w, const float* w_packed, const float* bias,
                 const int32_t* nbr, const int32_t* order, const uint32_t* group_mask32, int32_t K, float* fout,
                 int64_t n_out, int32_t cout, int32_t act, const float* film, const float* residual, void* stream) {
    PCC_REQUIRE(K >= 1 && K <= 27, "pcc_conv_fwd: K=%d out of range", K);
    PCC_REQUIRE(cin % 32 != 0 || cin <= 256, "pcc_conv_fwd: MFMA path supports cin <= 256 (got %d)", cin);
    PCC_REQUIRE(nbr != nullptr || (K == 1 && n_in == n_out), "pcc_conv_fwd: nbr == NULL needs K == 1 and n_in == n_out");
    PCC_REQUIRE(act >= 0 && act <= 2, "pcc_conv_fwd: bad activation %d", act);
    if (n_out <= 0) return PCC_OK;
    ConvArgs a;
    a.fin = fin; a.w = w; a.wp = w_packed; a.bias = bias; a.nbr = nbr; a.order = order; a.gmask = group_mask32; a.fout = fout;
    a.film = film; a.residual = residual; a.n_in = n_in; a.n_out = n_out; a.cin = cin; a.cout = cout;
    a.coutp = round_up32(cout); a.K = K; a.act = act; a.bf16 = 0;
    hipStream_t st = as_stream(stream);
    if (cin % 32 == 0) {
        PCC_REQUIRE(w_packed != nullptr, "pcc_conv_fwd: MFMA path (cin=%d) needs packed weights", cin);
        PCC_REQUIRE(((reinterpret_cast<uintptr_t>(fin) | reinterpret_cast<uintptr_t>(w_packed)) & 15) == 0,
                    "pcc_conv_fwd: fin and w_packed must be 16-byte aligned (16-byte LDS-DMA loads)");
        // Row-tile height, measured on MI355X: 128-wide outputs run best with 64-row tiles (48 KB of LDS per
        // workgroup -> 3 workgroups per CU, twice the tiles per launch -> less tail loss on mid-size layers);
        // 64-wide outputs with 128-row tiles (half the weight-slab traffic per MFMA).  PCC_CONV_BM=64|128
        // forces one height for A/B testing.
        static int bm = -1;
        if (bm < 0) { const char* e = getenv("PCC_CONV_BM"); bm = e ? atoi(e) : 0; }
        // Launches of a few hundred 32 x 32 tiles: one 16 x 16 MFMA block per wave, deep gather pipeline (conv_small_kernel)
        if (bm == 0 && small_launch(a) && fits_buffer_path(a)) return launch_small(a, st);
        // Launches that would not fill the chip (256 CUs x 3 workgroups) with 128-wide tiles are split
        // into 64-wide column tiles: twice the workgroups, half the MFMAs per step and workgroup.  The
        // accumulation order of an output element does not depend on the tile shape, so results are
        // bit-identical across configurations.
        const int64_t wgs128 = ((a.n_out + 63) / 64) * (a.coutp / 128);
        if (a.coutp % 128 == 0 && bm == 0 && wgs128 < 768) return launch_mfma<64, 64, 2, 2>(a, st);
        // Mid-size launches (one to three rounds of 64 x 128 workgroups on 256 CUs x 3) take 32-row tiles — all four waves on
        // one 32-row MFMA tile, each a 32-column slice: twice the workgroups at 40 KB of LDS (four per CU), so the launch's
        // last round is fuller, and a workgroup executes exactly the offsets its own 32 rows need.  Measured on the config-2
        // frame: the 72 k-row layers 0.34-0.51 -> 0.27-0.47 ms (12 launches), no gain from 233 k rows on, a loss on the
        // 1.26 M-row layers (more weight-slab traffic per MFMA).  PCC_CONV_BM32_MAX=<workgroups> moves the switch (0 = off).
        static int64_t bm32_max = -1;
        if (bm32_max < 0) { const char* e = getenv("PCC_CONV_BM32_MAX"); bm32_max = e ? atoll(e) : 3000; }
        if (a.coutp % 128 == 0 && (bm == 32 || (bm == 0 && wgs128 >= 768 && wgs128 < bm32_max))) return launch_mfma<32, 128, 1, 4>(a, st);
        if (a.coutp % 128 == 0) return bm == 128 ? launch_mfma<128, 128, 2, 2>(a, st) : launch_mfma<64, 128, 2, 2>(a, st);
        if (a.coutp % 64 == 0) return bm == 64 ? launch_mfma<64, 64, 2, 2>(a, st) : launch_mfma<128, 64, 2, 2>(a, st);
        return launch_mfma<128, 32, 4, 1>(a, st);
    }
    PCC_REQUIRE(w != nullptr, "pcc_conv_fwd: thin path (cin=%d) needs raw weights", cin);
    switch (cin) {
        case 1: return launch_thin<1>(a, st);
        case 2: return launch_thin<2>(a, st);
        case 3: return launch_thin<3>(a, st);        // 3, 6, 12, 24: C_bottleneck * 3 / 2 of a hyperprior over a narrow tensor
        case 4: return launch_thin<4>(a, st);        // (the two-hyperprior variant codes the 2-channel q-map: model/entropy_models.py:140-147)
        case 6: return launch_thin<6>(a, st);
        case 8: return launch_thin<8>(a, st);
        case 12: return launch_thin<12>(a, st);
        case 16: return launch_thin<16>(a, st);
        case 24: return launch_thin<24>(a, st);
        default:
            pcc::set_error("pcc_conv_fwd: unsupported cin=%d (need a multiple of 32 or one of 1,2,3,4,6,8,12,16,24)", cin);
            return PCC_ERR_UNSUPPORTED;
    }
}

}  // extern "C"
