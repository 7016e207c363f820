// Sparse convolution forward, "compacted offsets" form (gfx950): full MFMA tiles in the map's own row order.
//
// conv.hip runs output rows sorted by neighbour mask so that a 32-row MFMA tile can skip the kernel offsets none of its
// rows has.  That order has two costs, both measured on the config-2 frame (round 3):
//   * gathered rows come from all over the feature tensor — every gather is an HBM access, and under that load the chip
//     delivers 2.0-2.1 GHz instead of 2.35 (the same kernel on the same launches with rows in the map's own order ISSUES
//     125-139 TFLOP/s instead of 106-125, but then multiplies zeros in 40-75 % of its tiles);
//   * where masks are too diverse to sort into full tiles (the sparse sets an untrained decoder keeps: 5.4 of 27
//     neighbours per row, tens of thousands of distinct masks) a tile executes 1.6-1.8 x the offsets its rows need.
// Here a workgroup owns a GROUP of 256 consecutive output rows (the map's own order, i.e. spatially compact) x 64 output
// columns and, per kernel offset k, the COMPACTED list of the group's rows that have a neighbour at k (pcc_compact_map
// builds the lists once per map).  The MFMA tiles are 32 consecutive LIST entries — full whatever the masks look like (the last
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
// per-group compacted offset lists of a kernel map.  Padding entries (a list is padded to a multiple of 32) gather
// nothing (input row -1) and name an output row that is NOT in the list — the first such row of the group; one exists
// whenever the list is shorter than the group — so the lanes that carry them read that row's accumulator, add exact
// zeros and write the same bits back while no other lane touches the row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void compact_map_kernel(const int32_t* __restrict__ nbr, int64_t n_out, int K,
                                                          int32_t* __restrict__ ent_in, uint32_t* __restrict__ ent_row4,
                                                          uint8_t* __restrict__ cnt) {
    __shared__ int32_t tile[CO_RT * 27];
    __shared__ int32_t lin[27 * CO_RT];
    __shared__ uint8_t lrow[27 * CO_RT];
    __shared__ int32_t c_s[32];
    const int64_t g = blockIdx.x;
    const int64_t row0 = g * CO_RT;
    const int rows = (int)((n_out - row0 < CO_RT) ? (n_out - row0) : CO_RT);
    const int total = rows * K;
    const int32_t* src = nbr + row0 * K;
    for (int e = threadIdx.x; e < CO_RT * K; e += 256) {
        tile[e] = e < total ? src[e] : -1;
        lin[e] = -1;
    }
    if (threadIdx.x < 32) c_s[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int k = wid; k < K; k += 4) {
        int base = 0, first_absent = CO_RT;
#pragma unroll
        for (int q = 0; q < CO_RT / 64; ++q) {
            const int r = q * 64 + lane;
            const int32_t v = tile[r * K + k];
            const bool valid = v >= 0;
            const unsigned long long m = __ballot(valid);
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (valid) {
                lin[k * CO_RT + pos] = v;
                lrow[k * CO_RT + pos] = (uint8_t)r;
            }
            base += __popcll(m);
            if (first_absent == CO_RT && ~m != 0ull) first_absent = q * 64 + (__ffsll((long long)~m) - 1);
        }
        for (int p = base + lane; p < CO_RT; p += 64) lrow[k * CO_RT + p] = (uint8_t)first_absent;      // base < CO_RT => first_absent < CO_RT
        if (lane == 0) c_s[k] = base;
    }
    __syncthreads();
    int32_t* dst = ent_in + g * K * CO_RT;
    for (int e = threadIdx.x; e < K * CO_RT; e += 256) dst[e] = lin[e];
    // ent_row4[g][k][sp][r]: byte i = output row of entry 32 (sp + 2 i) + r  (what wave-half sp's lane r needs)
    uint32_t* dr = ent_row4 + g * K * 64;
    for (int e = threadIdx.x; e < K * 64; e += 256) {
        const int k = e >> 6, sp = (e >> 5) & 1, r = e & 31;
        const uint8_t* l = lrow + k * CO_RT + r;
        uint32_t w = 0;
#pragma unroll
        for (int i = 0; i < CO_RT / 64; ++i) w |= (uint32_t)l[32 * (sp + 2 * i)] << (8 * i);
        dr[e] = w;
    }
    // lengths 0 .. 256: two bytes per offset
    if (threadIdx.x < 32) reinterpret_cast<uint16_t*>(cnt)[g * 32 + threadIdx.x] = (uint16_t)c_s[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// the convolution.  Workgroup = 4 waves on one group of CO_RT = 256 rows x 64 output columns: wave (cb, sp) owns the
// 32-column block cb of the MFMA tiles sp, sp + 2, sp + 4, sp + 6 of every list (a tile = 32 consecutive list entries).
// Within one offset a row is in one tile, so the waves' accumulator traffic is disjoint; between offsets a row may change
// tiles (= waves), so a wave writes its tiles back BEFORE the barrier that ends the offset's last step and reads the next
// offset's tiles after it.  The staging buffers (gathered rows, weight slab) are double-buffered with one barrier per
// step like conv.hip's.  160 KB of LDS hold one such workgroup per CU (accumulators 68 KB, row images 64 KB, weight
// slabs 16 KB): one wave per SIMD, so everything that is not an MFMA is issued in the shadow of one (see dma_piece).
// ---------------------------------------------------------------------------------------------
constexpr int CO_BN = 64;
constexpr int co_lds_floats() { return CO_RT * (CO_BN + 4) + 2 * CO_RT * 32 + 2 * 8 * CO_BN * 4; }

template <int CCH>
__global__ __launch_bounds__(256) void conv_co_kernel(const ConvCoArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BN = CO_BN;
    constexpr int RT = CO_RT;
    constexpr int SW = RT / 64;                    // MFMA tiles per wave and list (at most)
    constexpr int ACC_LD = BN + 4;                 // floats; + 4: rows 16 bytes apart in the bank pattern
    constexpr int ACC_ELEMS = RT * ACC_LD;
    constexpr int A_ELEMS = RT * 32;               // per buffer: 128-B rows, 16-B slots XOR-swizzled by (row >> 1) & 7
    constexpr int W_ELEMS = 8 * BN * 4;
    constexpr int W_LOADS = (8 * BN) / 256;
    constexpr int RPT = RT / 8 / 4;                // gather DMAs per wave and step at a full list

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ACC = smem;
    float* As = smem + ACC_ELEMS;
    float* Ws = As + 2 * A_ELEMS;

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int cb = __builtin_amdgcn_readfirstlane(wid & 1), sp = __builtin_amdgcn_readfirstlane(wid >> 1);
    const int ntiles_n = a.cout / BN;
    const int64_t gt = blockIdx.x / ntiles_n;      // my group
    const int nt = blockIdx.x - (int)(gt * ntiles_n);
    const int K = a.K;

    // list lengths (lane k holds offset k) and the offsets the group has
    const int c_v = lane < K ? (int)reinterpret_cast<const uint16_t*>(a.cnt)[gt * 32 + lane] : 0;
    const uint32_t tmask = __builtin_amdgcn_readfirstlane((uint32_t)__ballot(c_v > 0));

    // accumulators start at zero (rows no list ever names keep it): wave (cb, sp) clears rows sp, sp + 2, .. of block cb
    for (int e = lane; e < (RT / 2) * 8; e += 64) {
        const int row = 2 * (e >> 3) + sp, ch = e & 7;
        *reinterpret_cast<f32x4*>(ACC + row * ACC_LD + cb * 32 + 4 * ch) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }

    if (tmask != 0u) {
        __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.fin), 0, (int)(uint32_t)(a.n_in * a.cin * 4), CO_BUF_FLAGS);
        __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.wp), 0, (int)((uint32_t)K * a.cin * a.cout * 4), CO_BUF_FLAGS);
        __amdgpu_buffer_rsrc_t rsrc_e = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int32_t*>(a.ent_in), 0, (int)(uint32_t)(a.n_groups * K * RT * 4), CO_BUF_FLAGS);
        __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t*>(a.ent_row4), 0, (int)(uint32_t)(a.n_groups * K * 64 * 4), CO_BUF_FLAGS);

        // gather roles: one wave-instruction fills 1 KB = 8 consecutive list entries; instruction i of wave w covers
        // entries (4 i + w) 8 .. + 7 of the image; lane -> (entry, 16-B slot)
        const int wave_u = __builtin_amdgcn_readfirstlane(wid);
        uint32_t q16[RPT], e_voff[RPT], a_voff[RPT];
        int idx_nxt[RPT];
        int p_of[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int p = (i * 4 + wave_u) * 8 + (lane >> 3);
            p_of[i] = p;
            q16[i] = (uint32_t)(((lane & 7) ^ ((p >> 1) & 7)) * 16);
            e_voff[i] = (uint32_t)((gt * K * RT + p) * 4);
            idx_nxt[i] = -1;
            a_voff[i] = CO_BUF_OOB;
        }
        const uint32_t r4_voff = (uint32_t)((gt * K * 64 + sp * 32 + r) * 4);
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

        auto len_of = [&](int k) { return __builtin_amdgcn_readlane(c_v, k); };   // list length at offset k
        auto load_idx = [&](int k) {
#pragma unroll
            for (int i = 0; i < RPT; ++i)
                idx_nxt[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc_e, e_voff[i], k * RT * 4, 0);
            lrow4_nxt = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc_r, r4_voff, k * 64 * 4, 0);
        };
        auto set_src = [&](int len) {            // idx_nxt -> byte offsets of my gather rows (or out of range -> zeros)
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const bool ok = p_of[i] < len && idx_nxt[i] >= 0;
                a_voff[i] = ok ? (uint32_t)idx_nxt[i] * a_row_bytes + q16[i] : CO_BUF_OOB;
            }
            lrow4_cur = lrow4_nxt;
        };
        // The DMAs of one step as numbered pieces, so that compute() can issue them one at a time between its MFMAs: this
        // workgroup is alone on its CU (one wave per SIMD), so an instruction that is not issued in the shadow of a running
        // MFMA idles the matrix pipe (an LDS-DMA costs ~60 issue cycles; ten of them in front of a step were 15 % of it).
        // Pieces 0 .. RPT-1: my gather instructions (skipped past the list's last tile, wave-uniform); RPT .. RPT+W_LOADS-1:
        // my share of the weight slab.
        constexpr int NDMA = RPT + W_LOADS;
        auto dma_piece = [&](int k, int len, auto cc, auto bufc, auto pc) {
            constexpr int c = decltype(cc)::value;
            constexpr int buf = decltype(bufc)::value;
            constexpr int piece = decltype(pc)::value;
            if constexpr (piece < RPT) {
                // up to the end of the list's last TILE: its padding entries must read zeros (out-of-range lanes of an LDS-DMA
                // write zeros), not what an earlier step left in the image — they add into a real row's accumulator
                if ((piece * 4 + wave_u) * 8 < ((len + 31) & ~31))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(As + buf * A_ELEMS + (piece * 4 + wave_u) * 256), 16,
                                                             a_voff[piece], c * 128, 0, 0);
            } else {
                constexpr int j = piece - RPT;
                const uint32_t wso = (uint32_t)k * w_kstride + (uint32_t)c * w_cstride;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(Ws + buf * W_ELEMS + (wave_u * 64 + 256 * j) * 4), 16,
                                                         w_voff[j], wso, 0, 0);
            }
        };
        auto dma = [&](int k, int len, auto cc, auto bufc) {
            co_static_for<0, NDMA>([&](auto pc) { dma_piece(k, len, cc, bufc, pc); });
        };

        // lane-constant LDS byte addresses of my fragments.  Lane (r, h) reads, for sub-block kk, the 16-B chunk 2 kk + h
        // of list entry 32 (sp + 2 i) + r (slot = chunk ^ ((r >> 1) & 7)) and of weight column 32 cb + r.
        const int sw = (r >> 1) & 7;
        uint32_t x_addr[2][4], w_addr[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                x_addr[b][kk] = (uint32_t)((ACC_ELEMS + b * A_ELEMS + (sp * 32 + r) * 32 +
                                            (((kk ^ (sw >> 1)) << 1) | (h ^ (sw & 1))) * 4) * 4);
            w_addr[b] = (uint32_t)((ACC_ELEMS + 2 * A_ELEMS + b * W_ELEMS + (h * BN + cb * 32 + r) * 4) * 4);
        }
        auto lds4 = [&](uint32_t addr) { return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(smem) + addr); };
        const uint32_t acc_base = (uint32_t)((cb * 32 + 4 * h) * 4);      // + row * ACC_LD * 4 + 32 j

        f32x16 acc[SW];

        // MFMAs of one step; `issue(piece)` is called once per piece of the NEXT step's DMAs, each after a group of MFMAs
        // has been issued (the pieces ride in the MFMAs' shadow); pieces the MFMA slots do not cover are issued at the end.
        auto compute = [&](auto bufc, auto sc, auto&& issue) {
            constexpr int buf = decltype(bufc)::value;
            constexpr int S = decltype(sc)::value;
            f32x4 xv[2][S], wv[2];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < S; ++s) xv[0][s] = lds4(x_addr[buf][0] + s * 64 * 32 * 4);
            wv[0] = lds4(w_addr[buf]);
            co_static_for<0, 4>([&](auto kkc) {
                constexpr int kk = decltype(kkc)::value;
                constexpr int cbuf = kk & 1, nb = cbuf ^ 1;
                if constexpr (kk + 1 < 4) {
#pragma unroll
                    for (int s = 0; s < S; ++s) xv[nb][s] = lds4(x_addr[buf][kk + 1] + s * 64 * 32 * 4);
                    wv[nb] = lds4(w_addr[buf] + (2 * (kk + 1) * BN) * 16);
                }
                co_static_for<0, 4>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[cbuf][q], xv[cbuf][s][q], acc[s], 0, 0, 0);
                    constexpr int slot = kk * 4 + q;
                    if constexpr (slot < NDMA) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue(std::integral_constant<int, slot>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            });
            co_static_for<16, NDMA>([&](auto pc) { issue(pc); });
            __builtin_amdgcn_s_setprio(0);
        };

        // One kernel offset: C <- ACC, CCH steps, ACC <- C (before the last step's barrier).  Step c computes chunk c from
        // buffer (P + c) & 1 while the DMAs of the following step fill the other one; the offset's last step starts the
        // next live offset (its list entries arrived one offset ago) and fetches the entries of the one after.
        uint32_t rem = tmask;
        int k = __builtin_ctz(rem);
        rem &= rem - 1u;
        auto offset_body = [&](auto pc, auto sc) {
            constexpr int P = decltype(pc)::value;
            constexpr int S = decltype(sc)::value;          // my tiles of this list
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
                constexpr bool last = (c + 1 == CCH);
                if constexpr (last) {
                    // the row_off of this offset were taken above: lrow4_cur may move on
                    set_src(len_next);
                }
                const int dk = last ? (knext >= 0 ? knext : k) : k;
                const int dlen = last ? len_next : len;
                auto issue = [&](auto pcc) {
                    dma_piece(dk, dlen, std::integral_constant<int, last ? 0 : c + 1>{}, std::integral_constant<int, buf ^ 1>{}, pcc);
                };
                if constexpr (S > 0) compute(std::integral_constant<int, buf>{}, sc, issue);
                else co_static_for<0, NDMA>([&](auto pcc) { issue(pcc); });
                if constexpr (last) {
                    load_idx(kn2);
                    if constexpr (S > 0) {
#pragma unroll
                        for (int s = 0; s < S; ++s)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + row_off[s] + 32 * j) =
                                    f32x4{acc[s][4 * j], acc[s][4 * j + 1], acc[s][4 * j + 2], acc[s][4 * j + 3]};
                    }
                }
                __syncthreads();       // vmcnt(0) + lgkmcnt(0) + barrier: next image complete, this one free, accumulators visible
            });
            const bool more = knext >= 0;
            k = more ? knext : k;
            rem = rem2;
            return more;
        };
        auto offset_any = [&](auto pc) {
            const int S_total = (len_of(k) + 31) >> 5;
            const int S = (S_total - sp + 1) >> 1;          // tiles sp, sp + 2, .. below S_total
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
    } else {
        __syncthreads();                    // the zeroed accumulators of the other wave-half
    }

    // epilogue: column block cb of rows sp, sp + 2, ..: 8 lanes x 16 B per row: bias, FiLM, activation, residual, store
    const int ch = lane & 7;
    const int col = nt * BN + cb * 32 + 4 * ch;
    f32x4 b4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + col);
#pragma unroll 4
    for (int it = 0; it < RT / 16; ++it) {
        const int row = sp * (RT / 2) + it * 8 + (lane >> 3);
        const int64_t grow = gt * RT + row;
        if (grow >= a.n_out) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(ACC + row * ACC_LD + cb * 32 + 4 * ch);
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
#endif
}

template <int CCH>
static int launch_co(const ConvCoArgs& a, hipStream_t st) {
    static bool attr_set = false;
    auto kern = conv_co_kernel<CCH>;
    const int lds = co_lds_floats() * (int)sizeof(float);
    if (!attr_set) {
        PCC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const int64_t blocks = a.n_groups * (a.cout / CO_BN);
    PCC_REQUIRE(blocks < (1ll << 31), "conv(co): grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, a);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

static int launch_co_cch(const ConvCoArgs& a, hipStream_t st) {
    switch (a.cin / 32) {
        case 1: return launch_co<1>(a, st);
        case 2: return launch_co<2>(a, st);
        case 3: return launch_co<3>(a, st);
        case 4: return launch_co<4>(a, st);
        case 6: return launch_co<6>(a, st);
        case 8: return launch_co<8>(a, st);
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
    return launch_co_cch(a, as_stream(stream));
}

}  // extern "C"
