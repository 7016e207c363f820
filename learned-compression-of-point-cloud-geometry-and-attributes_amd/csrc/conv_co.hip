// Sparse convolution forward, "compacted offsets" form (gfx950): full MFMA tiles in the map's own row order.
//
// conv.hip runs output rows sorted by neighbour mask so that a 32-row MFMA tile can skip the kernel offsets none of its
// rows has.  That order has two costs, both measured on the config-2 frame (round 3):
//   * gathered rows come from all over the feature tensor — every gather is an HBM access, and under that load the chip
//     delivers 2.0-2.1 GHz instead of 2.35 (the same kernel on the same launches with rows in the map's own order ISSUES
//     125-139 TFLOP/s instead of 106-125, but then multiplies zeros in 40-75 % of its tiles);
//   * where masks are too diverse to sort into full tiles (the sparse sets an untrained decoder keeps: 5.4 of 27
//     neighbours per row, tens of thousands of distinct masks) a tile executes 1.6-1.8 x the offsets its rows need.
// Here a workgroup owns a GROUP of 128 consecutive output rows (a "team": the map's own order, i.e. spatially compact) and,
// per kernel offset k, the COMPACTED list of the group's rows that have a neighbour at k (pcc_compact_map builds the
// lists once per map).  The MFMA tiles are 32 consecutive LIST entries — full whatever the masks look like (the last
// tile of an offset is padded) — and the accumulators of the group live in LDS between offsets:
//     for k ascending:  C <- ACC[rows of the list]   (LDS -> registers, one row per lane)
//                       C += X[gathered neighbours] . W[k]   (CCH steps of 32 channels, as in conv.hip)
//                       ACC[rows of the list] <- C
// The matrix product is issued transposed (D^T = W^T X^T: the MFMA's A operand is the weight fragment, B the gathered
// rows), which puts ONE output row in each lane (lane r of the wave = list entry r, its 16 registers = 16 output
// channels), so a lane reads and writes its own row of ACC with 16-byte LDS accesses wherever that row sits.
//
// Arithmetic: per output element the same fp32 MFMA chain as conv.hip — offsets ascending, channel chunks ascending,
// v_mfma_f32_32x32x2_f32 k-steps in the same order, accumulator starting at +0; moving the accumulator through LDS
// between offsets does not change its value, and conv.hip's extra steps (offsets a row lacks but its tile has) add
// exact zeros.  Results are bit-identical to conv.hip's (tests/test_conv_co.py), hence independent of row order, group
// placement and batch composition like there.
//
// Roofline: MFMA fp32 (157.3 TFLOP/s); algorithmic FLOPs per launch = 2 * pairs * cin * cout.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace pcc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int CO_RT = PCC_COMPACT_GROUP;      // rows per group (team)
constexpr int CO_DUMMY = CO_RT;               // ACC row that padded list entries read and write
constexpr uint32_t CO_BUF_OOB = 0xFFFFF000u;  // voffset of lanes that must read zeros (>= num_records)
constexpr uint32_t CO_BUF_FLAGS = 0x00020000u;

struct ConvCoArgs {
    const float* fin;
    const float* wp;            // packed [K, cin/4, coutp, 4] (pcc_conv_pack_weights)
    const float* bias;          // [cout] or null
    const int32_t* ent_in;      // [n_groups][K][CO_RT] input row of list entry p, -1 = padding
    const uint32_t* ent_row4;   // [n_groups][K][32]: byte s of word r = group-local output row of entry 32 s + r
    const uint8_t* cnt;         // [n_groups][32] list lengths
    float* fout;
    const float* film;          // [n_out, 2*cout] or null
    const float* residual;      // [n_out, cout] or null
    int64_t n_in, n_out, n_groups;
    int cin, cout, K, act;
};

template <int I, int N, class F>
__device__ __forceinline__ void co_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        co_static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ float co_act(float v, int act) {
    if (act == PCC_ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == PCC_ACT_LEAKY_RELU) return v > 0.0f ? v : 0.01f * v;
    return v;
}

// ---------------------------------------------------------------------------------------------
// per-group compacted offset lists of a kernel map
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void compact_map_kernel(const int32_t* __restrict__ nbr, int64_t n_out, int K,
                                                          int32_t* __restrict__ ent_in, uint32_t* __restrict__ ent_row4,
                                                          uint8_t* __restrict__ cnt) {
    __shared__ int32_t tile[CO_RT * 27];
    __shared__ int32_t lin[27 * CO_RT];
    __shared__ uint8_t lrow[27 * CO_RT];
    __shared__ uint8_t c_s[32];
    const int64_t g = blockIdx.x;
    const int64_t row0 = g * CO_RT;
    const int rows = (int)((n_out - row0 < CO_RT) ? (n_out - row0) : CO_RT);
    const int total = rows * K;
    const int32_t* src = nbr + row0 * K;
    for (int e = threadIdx.x; e < CO_RT * K; e += 256) {
        tile[e] = e < total ? src[e] : -1;
        lin[e] = -1;
        lrow[e] = (uint8_t)CO_DUMMY;
    }
    if (threadIdx.x < 32) c_s[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int k = wid; k < K; k += 4) {
        int base = 0;
#pragma unroll
        for (int half = 0; half < CO_RT / 64; ++half) {
            const int r = half * 64 + lane;
            const int32_t v = tile[r * K + k];
            const bool valid = v >= 0;
            const unsigned long long m = __ballot(valid);
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (valid) {
                lin[k * CO_RT + pos] = v;
                lrow[k * CO_RT + pos] = (uint8_t)r;
            }
            base += __popcll(m);
        }
        if (lane == 0) c_s[k] = (uint8_t)base;
    }
    __syncthreads();
    int32_t* dst = ent_in + g * K * CO_RT;
    for (int e = threadIdx.x; e < K * CO_RT; e += 256) dst[e] = lin[e];
    uint32_t* dr = ent_row4 + g * K * 32;
    for (int e = threadIdx.x; e < K * 32; e += 256) {
        const int k = e >> 5, r = e & 31;
        const uint8_t* l = lrow + k * CO_RT + r;
        uint32_t w = 0;
#pragma unroll
        for (int s = 0; s < CO_RT / 32; ++s) w |= (uint32_t)l[32 * s] << (8 * s);
        dr[e] = w;
    }
    if (threadIdx.x < 32) cnt[g * 32 + threadIdx.x] = c_s[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// the convolution.  Workgroup = 4 waves = TEAMS teams of BN / 32 waves; a team owns one group of CO_RT rows and BN
// output columns, wave cb of the team the 32-column block cb of every MFMA tile of its team.  ACC rows are touched by
// exactly one wave per column block, so the accumulators need no synchronisation; the staging buffers (gathered rows per
// team, one weight slab shared by the teams) are double-buffered with one barrier per step like conv.hip's.
// ---------------------------------------------------------------------------------------------
template <int BN>
constexpr int co_lds_floats() {
    constexpr int WPT = BN / 32, TEAMS = 4 / WPT;
    return TEAMS * (CO_RT + 1) * (BN + 4) + 2 * TEAMS * CO_RT * 32 + 2 * 8 * BN * 4;
}

template <int BN, int CCH>
__global__ __launch_bounds__(256) void conv_co_kernel(const ConvCoArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WPT = BN / 32;                   // waves per team = 32-column blocks
    constexpr int TEAMS = 4 / WPT;
    constexpr int RT = CO_RT;
    constexpr int SMAX = RT / 32;                  // MFMA tiles per list
    constexpr int ACC_LD = BN + 4;                 // floats; + 4: rows 16 bytes apart in the bank pattern
    constexpr int ACC_ELEMS = (RT + 1) * ACC_LD;   // per team (+ the dummy row)
    constexpr int A_ELEMS = RT * 32;               // per team and buffer: 128-B rows, 16-B slots XOR-swizzled by (row >> 1) & 7
    constexpr int W_ELEMS = 8 * BN * 4;
    constexpr int W_LOADS = (8 * BN) / 256;
    constexpr int RPT = RT / 8 / WPT;              // gather DMAs per wave and step at a full list
    static_assert(BN == 64 || BN == 128, "column tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ACC = smem;
    float* As = smem + TEAMS * ACC_ELEMS;
    float* Ws = As + 2 * TEAMS * A_ELEMS;

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int team = __builtin_amdgcn_readfirstlane(wid / WPT), cb = __builtin_amdgcn_readfirstlane(wid % WPT);
    const int ntiles_n = a.cout / BN;
    const int64_t wg = blockIdx.x / ntiles_n;
    const int nt = blockIdx.x - (int)(wg * ntiles_n);
    const int64_t gt = wg * TEAMS + team;          // my team's group
    const bool team_live = gt < a.n_groups;
    const int K = a.K;

    // list lengths of my team (lane k holds offset k) and the offsets any team of the workgroup has
    int c_v = 0;
    uint32_t tmask = 0u;
#pragma unroll
    for (int tt = 0; tt < TEAMS; ++tt) {
        const int64_t g2 = wg * TEAMS + tt;
        const int v = (g2 < a.n_groups && lane < K) ? (int)a.cnt[g2 * 32 + lane] : 0;
        tmask |= (uint32_t)__ballot(v > 0);
        if (tt == team) c_v = v;
    }
    tmask = __builtin_amdgcn_readfirstlane(tmask);

    // my column block of ACC starts at zero (rows no list ever names keep it)
    float* ACCt = ACC + team * ACC_ELEMS;
    for (int e = lane; e < (RT + 1) * 8; e += 64) {
        const int row = e >> 3, ch = e & 7;
        *reinterpret_cast<f32x4*>(ACCt + row * ACC_LD + cb * 32 + 4 * ch) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }

    if (tmask != 0u) {
        __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.fin), 0, (int)(uint32_t)(a.n_in * a.cin * 4), CO_BUF_FLAGS);
        __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.wp), 0, (int)((uint32_t)K * a.cin * a.cout * 4), CO_BUF_FLAGS);
        __amdgpu_buffer_rsrc_t rsrc_e = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int32_t*>(a.ent_in), 0, (int)(uint32_t)(a.n_groups * K * RT * 4), CO_BUF_FLAGS);
        __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t*>(a.ent_row4), 0, (int)(uint32_t)(a.n_groups * K * 32 * 4), CO_BUF_FLAGS);

        // gather roles: one wave-instruction fills 1 KB = 8 consecutive list entries; instruction i of wave cb covers
        // entries (i WPT + cb) 8 .. + 7 of my team's image; lane -> (entry, 16-B slot)
        uint32_t q16[RPT], e_voff[RPT], a_voff[RPT];
        int idx_nxt[RPT];
        int p_of[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int p = (i * WPT + cb) * 8 + (lane >> 3);
            p_of[i] = p;
            q16[i] = (uint32_t)(((lane & 7) ^ ((p >> 1) & 7)) * 16);
            e_voff[i] = team_live ? (uint32_t)((gt * K * RT + p) * 4) : CO_BUF_OOB;
            idx_nxt[i] = -1;
            a_voff[i] = CO_BUF_OOB;
        }
        const uint32_t r4_voff = team_live ? (uint32_t)((gt * K * 32 + r) * 4) : CO_BUF_OOB;
        uint32_t lrow4_nxt = 0u, lrow4_cur = 0u;
        uint32_t w_voff[W_LOADS];
#pragma unroll
        for (int j = 0; j < W_LOADS; ++j) {
            const int f = t + 256 * j;
            const int g = f / BN, col = f - g * BN;
            w_voff[j] = (uint32_t)((g * a.cout + nt * BN + col) * 16);
        }
        const uint32_t w_kstride = (uint32_t)(a.cin / 4) * a.cout * 16;       // bytes per kernel offset
        const uint32_t w_cstride = 8u * a.cout * 16;                           // bytes per 32-channel chunk
        const uint32_t a_row_bytes = (uint32_t)a.cin * 4;
        const int wave_u = __builtin_amdgcn_readfirstlane(wid);

        auto len_of = [&](int k) { return __builtin_amdgcn_readlane(c_v, k); };   // my team's list length at offset k
        auto load_idx = [&](int k) {
#pragma unroll
            for (int i = 0; i < RPT; ++i)
                idx_nxt[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc_e, e_voff[i], k * RT * 4, 0);
            lrow4_nxt = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc_r, r4_voff, k * 32 * 4, 0);
        };
        auto set_src = [&](int len) {            // idx_nxt -> byte offsets of my gather rows (or out of range -> zeros)
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const bool ok = p_of[i] < len && idx_nxt[i] >= 0;
                a_voff[i] = ok ? (uint32_t)idx_nxt[i] * a_row_bytes + q16[i] : CO_BUF_OOB;
            }
            lrow4_cur = lrow4_nxt;
        };
        auto dma = [&](int k, int len, auto cc, auto bufc) {
            constexpr int c = decltype(cc)::value;
            constexpr int buf = decltype(bufc)::value;
            float* Ab = As + (buf * TEAMS + team) * A_ELEMS;
#pragma unroll
            for (int i = 0; i < RPT; ++i)
                if ((i * WPT + cb) * 8 < len)      // wave-uniform: instructions past the list's end are not issued
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(Ab + (i * WPT + cb) * 256), 16, a_voff[i], c * 128, 0, 0);
            const uint32_t wso = (uint32_t)k * w_kstride + (uint32_t)c * w_cstride;
#pragma unroll
            for (int j = 0; j < W_LOADS; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(Ws + buf * W_ELEMS + (wave_u * 64 + 256 * j) * 4), 16,
                                                         w_voff[j], wso, 0, 0);
        };

        // lane-constant LDS byte addresses of my fragments.  Lane (r, h) reads, for sub-block kk, the 16-B chunk 2 kk + h
        // of list entry 32 s + r (slot = chunk ^ ((r >> 1) & 7)) and of weight column 32 cb + r.
        const int sw = (r >> 1) & 7;
        uint32_t x_addr[2][4], w_addr[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                x_addr[b][kk] = (uint32_t)((TEAMS * ACC_ELEMS + (b * TEAMS + team) * A_ELEMS + r * 32 +
                                            (((kk ^ (sw >> 1)) << 1) | (h ^ (sw & 1))) * 4) * 4);
            w_addr[b] = (uint32_t)((TEAMS * ACC_ELEMS + 2 * TEAMS * A_ELEMS + b * W_ELEMS + (h * BN + cb * 32 + r) * 4) * 4);
        }
        auto lds4 = [&](uint32_t addr) { return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(smem) + addr); };
        const uint32_t acc_base = (uint32_t)((team * ACC_ELEMS + cb * 32 + 4 * h) * 4);      // + row * ACC_LD * 4 + 32 j

        f32x16 acc[SMAX];

        auto compute = [&](auto bufc, auto sc) {
            constexpr int buf = decltype(bufc)::value;
            constexpr int S = decltype(sc)::value;
            f32x4 xv[2][S], wv[2];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < S; ++s) xv[0][s] = lds4(x_addr[buf][0] + s * 32 * 32 * 4);
            wv[0] = lds4(w_addr[buf]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cbuf = kk & 1, nb = cbuf ^ 1;
                if (kk + 1 < 4) {
#pragma unroll
                    for (int s = 0; s < S; ++s) xv[nb][s] = lds4(x_addr[buf][kk + 1] + s * 32 * 32 * 4);
                    wv[nb] = lds4(w_addr[buf] + (2 * (kk + 1) * BN) * 16);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[cbuf][q], xv[cbuf][s][q], acc[s], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        };

        // One kernel offset: C <- ACC, CCH steps, ACC <- C.  Step c computes chunk c from buffer (P + c) & 1 while the
        // DMAs of the following step fill the other one; the offset's last step starts the next live offset (its list
        // entries arrived one offset ago) and fetches the entries of the one after.
        uint32_t rem = tmask;
        int k = __builtin_ctz(rem);
        rem &= rem - 1u;
        auto offset_body = [&](auto pc, auto sc) {
            constexpr int P = decltype(pc)::value;
            constexpr int S = decltype(sc)::value;
            const int knext = rem ? __builtin_ctz(rem) : -1;
            const uint32_t rem2 = rem & (rem - 1u);
            const int kn2 = rem2 ? __builtin_ctz(rem2) : (knext >= 0 ? knext : k);
            const int len = len_of(k);
            const int len_next = knext >= 0 ? len_of(knext) : 0;
            uint32_t row_off[S > 0 ? S : 1];
            if constexpr (S > 0) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    row_off[s] = acc_base + ((lrow4_cur >> (8 * s)) & 0xFFu) * (uint32_t)(ACC_LD * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 v = lds4(row_off[s] + 32 * j);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[s][4 * j + e] = v[e];
                    }
                }
            }
            co_static_for<0, CCH>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int buf = (P + c) & 1;
                if constexpr (c + 1 < CCH) {
                    dma(k, len, std::integral_constant<int, c + 1>{}, std::integral_constant<int, buf ^ 1>{});
                } else {
                    // the row_off of this offset were taken above: lrow4_cur may move on
                    set_src(len_next);
                    dma(knext >= 0 ? knext : k, len_next, std::integral_constant<int, 0>{}, std::integral_constant<int, buf ^ 1>{});
                    load_idx(kn2);
                }
                if constexpr (S > 0) compute(std::integral_constant<int, buf>{}, sc);
                __syncthreads();       // vmcnt(0) + barrier: the next image is complete, this one is free
            });
            if constexpr (S > 0) {
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + row_off[s] + 32 * j) =
                            f32x4{acc[s][4 * j], acc[s][4 * j + 1], acc[s][4 * j + 2], acc[s][4 * j + 3]};
            }
            const bool more = knext >= 0;
            k = more ? knext : k;
            rem = rem2;
            return more;
        };
        auto offset_any = [&](auto pc) {
            const int S = (len_of(k) + 31) >> 5;
            switch (S) {
                case 0: return offset_body(pc, std::integral_constant<int, 0>{});
                case 1: return offset_body(pc, std::integral_constant<int, 1>{});
                case 2: return offset_body(pc, std::integral_constant<int, 2>{});
                case 3: return offset_body(pc, std::integral_constant<int, 3>{});
                default: return offset_body(pc, std::integral_constant<int, 4>{});
            }
        };

        load_idx(k);                        // the first offset's entries (every later offset's arrive an offset ahead)
        set_src(len_of(k));
        dma(k, len_of(k), std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        load_idx(rem ? __builtin_ctz(rem) : k);
        __syncthreads();
        while (true) {
            if (!offset_any(std::integral_constant<int, 0>{})) break;
            if constexpr (CCH & 1) {
                if (!offset_any(std::integral_constant<int, 1>{})) break;
            }
        }
    }

    // epilogue: my column block of my team's rows, 8 lanes x 16 B per row: bias, FiLM, activation, residual, store
    const int ch = lane & 7;
    const int col = nt * BN + cb * 32 + 4 * ch;
    f32x4 b4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + col);
    if (team_live) {
#pragma unroll 4
        for (int it = 0; it < RT / 8; ++it) {
            const int row = it * 8 + (lane >> 3);
            const int64_t grow = gt * RT + row;
            if (grow >= a.n_out) break;
            f32x4 v = *reinterpret_cast<const f32x4*>(ACCt + row * ACC_LD + cb * 32 + 4 * ch);
            v += b4;
            if (a.film) {
                const float* fr = a.film + grow * (2 * (int64_t)a.cout);
                const f32x4 be = *reinterpret_cast<const f32x4*>(fr + col);
                const f32x4 ga = *reinterpret_cast<const f32x4*>(fr + a.cout + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * be[e] + ga[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = co_act(v[e], a.act);
            if (a.residual) v += *reinterpret_cast<const f32x4*>(a.residual + grow * a.cout + col);
            *reinterpret_cast<f32x4*>(a.fout + grow * a.cout + col) = v;
        }
    }
#endif
}

template <int BN, int CCH>
static int launch_co(const ConvCoArgs& a, hipStream_t st) {
    static bool attr_set = false;
    auto kern = conv_co_kernel<BN, CCH>;
    const int lds = co_lds_floats<BN>() * (int)sizeof(float);
    if (!attr_set) {
        PCC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    constexpr int TEAMS = 4 / (BN / 32);
    const int64_t wgs = (a.n_groups + TEAMS - 1) / TEAMS;
    const int64_t blocks = wgs * (a.cout / BN);
    PCC_REQUIRE(blocks < (1ll << 31), "conv(co): grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, a);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

template <int BN>
static int launch_co_cch(const ConvCoArgs& a, hipStream_t st) {
    switch (a.cin / 32) {
        case 1: return launch_co<BN, 1>(a, st);
        case 2: return launch_co<BN, 2>(a, st);
        case 3: return launch_co<BN, 3>(a, st);
        case 4: return launch_co<BN, 4>(a, st);
        case 6: return launch_co<BN, 6>(a, st);
        case 8: return launch_co<BN, 8>(a, st);
        default: break;
    }
    pcc::set_error("pcc_conv_fwd_co: cin=%d not supported (32, 64, 96, 128, 192, 256)", a.cin);
    return PCC_ERR_UNSUPPORTED;
}

}  // namespace pcc

using namespace pcc;

extern "C" {

int64_t pcc_compact_map_groups(int64_t n_out) { return n_out <= 0 ? 0 : (n_out + CO_RT - 1) / CO_RT; }

int pcc_compact_map(const int32_t* nbr, int64_t n_out, int32_t K, int32_t* ent_in, uint32_t* ent_row4, uint8_t* cnt,
                    void* stream) {
    PCC_REQUIRE(K >= 1 && K <= 27, "pcc_compact_map: K=%d out of range", K);
    PCC_REQUIRE(nbr != nullptr && ent_in != nullptr && ent_row4 != nullptr && cnt != nullptr, "pcc_compact_map: null argument");
    if (n_out <= 0) return PCC_OK;
    const int64_t groups = pcc_compact_map_groups(n_out);
    PCC_REQUIRE(groups < (1ll << 31), "pcc_compact_map: too many rows");
    hipLaunchKernelGGL(compact_map_kernel, dim3((unsigned)groups), dim3(256), 0, as_stream(stream), nbr, n_out, K, ent_in,
                       ent_row4, cnt);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_conv_fwd_co(const float* fin, int64_t n_in, int32_t cin, const float* w_packed, const float* bias,
                    const int32_t* ent_in, const uint32_t* ent_row4, const uint8_t* cnt, int32_t K, float* fout,
                    int64_t n_out, int32_t cout, int32_t act, const float* film, const float* residual, void* stream) {
    PCC_REQUIRE(K >= 1 && K <= 27, "pcc_conv_fwd_co: K=%d out of range", K);
    PCC_REQUIRE(cin % 32 == 0 && cin <= 256, "pcc_conv_fwd_co: cin must be a multiple of 32 up to 256 (got %d)", cin);
    PCC_REQUIRE(cout % 64 == 0, "pcc_conv_fwd_co: cout must be a multiple of 64 (got %d)", cout);
    PCC_REQUIRE(act >= 0 && act <= 2, "pcc_conv_fwd_co: bad activation %d", act);
    PCC_REQUIRE(w_packed != nullptr && ent_in != nullptr && ent_row4 != nullptr && cnt != nullptr, "pcc_conv_fwd_co: null argument");
    PCC_REQUIRE(((reinterpret_cast<uintptr_t>(fin) | reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(fout) |
                  reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(film) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0,
                "pcc_conv_fwd_co: fin, w_packed, fout, bias, film and residual must be 16-byte aligned");
    if (n_out <= 0) return PCC_OK;
    ConvCoArgs a;
    a.fin = fin; a.wp = w_packed; a.bias = bias; a.ent_in = ent_in; a.ent_row4 = ent_row4; a.cnt = cnt; a.fout = fout;
    a.film = film; a.residual = residual; a.n_in = n_in; a.n_out = n_out; a.n_groups = pcc_compact_map_groups(n_out);
    a.cin = cin; a.cout = cout; a.K = K; a.act = act;
    const uint64_t lim = CO_BUF_OOB;
    PCC_REQUIRE((uint64_t)n_in * cin * 4 <= lim && (uint64_t)a.n_groups * K * CO_RT * 4 <= lim && (uint64_t)K * cin * cout * 4 <= lim,
                "pcc_conv_fwd_co: operands of 4 GiB and more are not supported (use pcc_conv_fwd)");
    hipStream_t st = as_stream(stream);
    if (cout % 128 == 0) return launch_co_cch<128>(a, st);
    return launch_co_cch<64>(a, st);
}

}  // extern "C"
