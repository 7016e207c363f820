// Backward of the sparse convolution (training path; the reference gets these
// from MinkowskiEngine's autograd functions behind every ME.Minkowski*Convolution*, train.py:194-206).
//
//   dX[i] = sum_k dY[j(i,k)] . W[k]^T   -> pcc_kernel_map_transpose + the forward kernel (pcc_conv_fwd) on
//                                          the transposed map with transposed weights: same MFMA loop.
//   dW[k] = sum_{(i,j) in map_k} X[i]^T dY[j]   -> conv_wgrad_kernel below (fp32 MFMA, rows are the GEMM's
//                                          K dimension), deterministic two-stage reduction.
//   db    = column sums of dY (caller).
//
// wgrad tiling: one workgroup owns kernel offset k, a 128 x 128 block of W[k] and every SPLIT-th group
// of 32 output rows (execution order of pcc_order_rows_by_mask; groups whose mask lacks k are
// skipped).  Per group the gathered X rows and the dY rows land in LDS as [chunk][row][32] images by
// buffer_load ... lds (absent neighbours: out-of-range offset -> zeros) and feed
// v_mfma_f32_32x32x2_f32 with A = X^T, B = dY: lane (r, h) reads X[row 2s+h][ch r] and dY[row 2s+h][co r] —
// two 128-B rows per ds_read_b32, all 64 banks once.  Single-buffered: 32 KB of LDS and 64
// accumulator registers per wave leave room for 4 workgroups per CU, which is what hides the
// index -> gather -> LDS round trips (no software pipeline).  Partials go to scratch
// [SPLIT][K][cin][cout]; a second kernel adds them in fixed order (bitwise reproducible).
// Roofline: MFMA fp32; algorithmic FLOPs = 2 * pairs * cin * cout, like the forward.
#include <stdlib.h>

#include "common.h"

namespace pcc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int WG_SPLIT_MAX = 128;     // row-group splits per (offset, weight block): the grid must be many times the chip
                                      // (256 CUs x 5 workgroups) because offsets differ 3x in pair count (the centre has them all)
constexpr int WG_SPLIT_THIN = 256;     // the scalar kernel walks its rows serially: many short walks
constexpr uint32_t WG_OOB = 0xFFFFF000u;
[[maybe_unused]] constexpr uint32_t WG_FLAGS = 0x00020000u;

__global__ __launch_bounds__(256) void map_transpose_kernel(const int32_t* __restrict__ nbr, int64_t n_out, int K,
                                                            int32_t* __restrict__ nbr_t, uint32_t* __restrict__ mask_t) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_out * K) return;
    const int i = nbr[e];
    if (i < 0) return;
    const int k = (int)(e % K);
    nbr_t[(int64_t)i * K + k] = (int32_t)(e / K);
    atomicOr(&mask_t[i], 1u << k);
}

// Workgroups are dispatched in blockIdx order and an offset's pair count falls with its distance from the centre
// (the centre offset pairs every row, a corner offset a third of them): 3x3x3 kernels hand out the centre first,
// then faces, edges, corners, so that the launch ends on its cheapest workgroups.
__device__ __forceinline__ int wgrad_offset_of(int slot, int K) {
    constexpr unsigned char heavy_first[27] = {13, 4, 10, 12, 14, 16, 22, 1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25,
                                               0, 2, 6, 8, 18, 20, 24, 26};
    return K == 27 ? heavy_first[slot] : slot;
}

struct WgradArgs {
    const float* fin;
    const float* dy;
    const int32_t* nbr;      // [n_out, K] in execution order
    const int32_t* order;    // [n_out] execution position -> row of dy, or null
    const uint32_t* gmask;   // [ceil(n_out / 32)] or null
    float* partial;          // [splits, K, cin, cout]
    int64_t n_in, n_out;
    int cin, cout, K;
    int split;               // row-group splits (MFMA kernel)
};

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                // [4 chunks][32 rows][32 ch]
    float* Bs = smem + 4 * 1024;     // [4 chunks][32 rows][32 co]
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    const int SPLIT = a.split;
    const int k = wgrad_offset_of(blockIdx.x / SPLIT, a.K), s = blockIdx.x % SPLIT;
    const int cin0 = blockIdx.y * 128, cout0 = blockIdx.z * 128;
    const int cbi = min(4, (a.cin - cin0) / 32), cbo = min(4, (a.cout - cout0) / 32);     // chunks present in this block
    // Blocks of at most 64 x 64 (64-channel layers) would leave three of the four waves without a tile and
    // half of the LDS images empty: there an iteration takes TWO row groups (waves 0-1 stage and consume the
    // first in image slots 0-1, waves 2-3 the second in slots 2-3), each wave multiplies 16 of its group's 32
    // rows into the whole 64 x 64 block, and every wave writes its own partial.
    const bool rowsplit = a.cin <= 64 && a.cout <= 64;       // a property of the layer, not of the block (host agrees)
    const int wm = rowsplit ? 0 : wave_u >> 1, wn = rowsplit ? 0 : wave_u & 1;           // 64 x 64 sub-block of the wave
    const int kp0 = rowsplit ? 8 * (wave_u & 1) : 0, kp1 = rowsplit ? kp0 + 8 : 16;
    const int cw = rowsplit ? (wave_u & 1) : wave_u;                                      // chunk this wave stages
    const int slotA = rowsplit ? 2 * (wave_u >> 1) : 2 * wm, slotB = rowsplit ? 2 * (wave_u >> 1) : 2 * wn;

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.0f;

    __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fin), 0, (int)(uint32_t)(a.n_in * a.cin * 4), WG_FLAGS);
    __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)(uint32_t)(a.n_out * a.cout * 4), WG_FLAGS);
    const int64_t ng = (a.n_out + 31) >> 5;
    const int slot = lane & 7, rsub = lane >> 3;          // DMA role: 8 lanes per row, 8 rows per instruction
    const bool live_m[2] = {2 * wm < cbi, 2 * wm + 1 < cbi};
    const bool live_n[2] = {2 * wn < cbo, 2 * wn + 1 < cbo};

    // my groups are s, s + SPLIT, ...; 64 of their masks are inspected per load (one per lane) and only the
    // groups that hold offset k are visited: a dependent scalar load per skipped group would cost ~1 us each
    int64_t gb = (int64_t)s - 64 * (int64_t)SPLIT;
    unsigned long long live = 0ull;
    auto pop = [&]() -> int64_t {                          // next of my groups that holds offset k, or -1
        while (!live) {
            gb += 64 * (int64_t)SPLIT;
            if (gb >= ng) return -1;
            const int64_t gmine = gb + (int64_t)lane * SPLIT;
            const uint32_t gml = (gmine < ng) ? (a.gmask ? a.gmask[gmine] : 0xffffffffu) : 0u;
            live = __ballot((gml >> k) & 1u);
        }
        const int bit = __ffsll(live) - 1;
        live &= live - 1;
        return gb + (int64_t)bit * SPLIT;
    };
    auto take = [&](bool& any) -> int64_t {                // this wave's group of the next iteration
        int64_t g = pop();
        any = g >= 0;
        if (rowsplit) {                                    // second group of the iteration (or none: zero images)
            const int64_t g2 = any ? pop() : -1;
            if (wave_u >= 2) g = g2;
        }
        return g;
    };
    // The neighbour indices and dY rows of a group are fetched one iteration ahead, beside the previous group's
    // gathers: an iteration then waits for one memory round trip (the gathers), not for two in sequence.
    int idx[4];
    int64_t row[4];
    bool okr[4];
    auto load_rows = [&](int64_t g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t pos = g * 32 + 8 * i + rsub;
            okr[i] = g >= 0 && pos < a.n_out;
            idx[i] = okr[i] ? a.nbr[pos * a.K + k] : -1;
            row[i] = okr[i] ? (a.order ? a.order[pos] : pos) : 0;
        }
    };
    bool any;
    int64_t g = take(any);
    load_rows(g);
    while (any) {
        // wave w stages one chunk of both operands: 4 instructions x 8 rows each
        uint32_t voa[4], vob[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            voa[i] = (idx[i] >= 0 && cw < cbi) ? (uint32_t)idx[i] * (uint32_t)(a.cin * 4) + (uint32_t)((cin0 + cw * 32) * 4 + slot * 16) : WG_OOB;
            vob[i] = (okr[i] && cw < cbo) ? (uint32_t)row[i] * (uint32_t)(a.cout * 4) + (uint32_t)((cout0 + cw * 32) * 4 + slot * 16) : WG_OOB;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(As + wave_u * 1024 + i * 256), 16, voa[i], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_ptr_t)(Bs + wave_u * 1024 + i * 256), 16, vob[i], 0, 0, 0);
        }
        g = take(any);
        load_rows(g);
        __syncthreads();
        const float* Ab = As + slotA * 1024 + h * 32 + r;
        const float* Bb = Bs + slotB * 1024 + h * 32 + r;
        for (int kp = kp0; kp < kp1; ++kp) {
            float av[2], bv[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) av[m] = Ab[m * 1024 + kp * 64];
#pragma unroll
            for (int n = 0; n < 2; ++n) bv[n] = Bb[n * 1024 + kp * 64];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
        __syncthreads();
    }

    // D[row = (reg & 3) + 8 (reg >> 2) + 4 h][col = r] of each 32 x 32 tile: row = input channel, col = output channel
    float* P = a.partial + ((int64_t)(rowsplit ? s * 4 + wave_u : s) * a.K + k) * a.cin * a.cout;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (!live_m[m]) continue;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (!live_n[n]) continue;
            const int co = cout0 + (2 * wn + n) * 32 + r;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int ci = cin0 + (2 * wm + m) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                P[(int64_t)ci * a.cout + co] = acc[m][n][reg];
            }
        }
    }
#endif
}

// Slice kernel for the 64-channel class (the default for K = 27): what bounds these shapes is the byte rate of their
// gathers — every (X row, dY row) pair is 512 B for 8 k FLOP, twice the forward convolution's bytes per FLOP, and the
// measured rates (47-57 TFLOP/s) are the forward kernel's gather rate divided by two — so the tiling is chosen to move
// fewer bytes WITHOUT idling waves (round 2's four-offsets-per-workgroup kernel, one offset per wave, moved fewer bytes
// and idled the waves whose offset a group lacks: same time; removed in round 4):
//   * a workgroup owns the nine offsets of one dz plane and every SPLIT-th group of 32 output rows; the group's dY rows are
//     staged ONCE and reused by every offset of the plane that occurs in the group (about four of nine on a surface):
//     8 + 8 L KB per L offset-steps instead of 16 L;
//   * every wave works in every step: wave (m, n) owns the 32 x 32 tile (m, n) of the 64 x 64 block and keeps one
//     accumulator per offset of the plane (9 x 16 registers); a step = one (group, offset) = 16 MFMAs per wave;
//   * double-buffered images, the forward kernel's loop shape: the gathers of step s + 1 are issued before the MFMAs of
//     step s, the indices of step s + 2 are fetched beside them, one barrier per step.
// AHEAD = how many steps the gathers run in front of the MFMAs: 1 = two images per operand and __syncthreads() per step;
// 2 = three images, the step's wait is a counted s_waitcnt vmcnt(4) (everything but the newest four memory operations — the
// X gathers of the step after next and the index loads of the one after that, two each at least — has landed) + a bare
// s_barrier, so two steps' gathers are in flight per workgroup.
template <int O, int AHEAD>
__global__ __launch_bounds__(256) void conv_wgrad_slice_kernel(const WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NB = AHEAD + 1;     // images per operand
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                 // [NB][2 chunks][32 rows][32 ch]
    float* Ys = smem + NB * 2048;     // [NB][2 chunks][32 rows][32 co]
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    const int SPLIT = a.split;
    const int set = blockIdx.x / SPLIT, s = blockIdx.x % SPLIT;
    const int k0 = set * O;
    const uint32_t set_mask = (((1u << O) - 1u) << k0) & 0x7FFFFFFu;
    const int cbi = a.cin / 32, cbo = a.cout / 32;
    const int tm = wave_u >> 1, tn = wave_u & 1;                  // my 32 x 32 tile of W[k]
    const bool tile_live = tm < cbi && tn < cbo;
    const int cw = wave_u & 1, rh = wave_u >> 1;                  // my DMA role: chunk cw, rows 16 rh .. 16 rh + 15

    f32x16 acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[o][i] = 0.0f;

    __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fin), 0, (int)(uint32_t)(a.n_in * a.cin * 4), WG_FLAGS);
    __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)(uint32_t)(a.n_out * a.cout * 4), WG_FLAGS);
    const int64_t ng = (a.n_out + 31) >> 5;
    const int slot = lane & 7, rsub = lane >> 3;

    // ---- step iterator: (group, offset) pairs of my groups, groups ascending, offsets ascending inside a group --------
    int64_t gb = (int64_t)s - 64 * (int64_t)SPLIT;
    unsigned long long glive = 0ull;
    uint32_t gml = 0u;
    int64_t it_g = -1;            // group of the pending offsets
    uint32_t it_rem = 0u;         // offsets of it_g not handed out yet
    int it_ysel = NB - 1;         // dY image of it_g (advances per group)
    struct Step { int64_t g; int o; int ysel; bool first; bool valid; };
    auto next_step = [&]() -> Step {
        Step st;
        st.first = false;
        if (it_rem == 0u) {
            while (!glive) {
                gb += 64 * (int64_t)SPLIT;
                if (gb >= ng) { st.g = -1; st.o = k0; st.ysel = 0; st.valid = false; return st; }
                const int64_t gmine = gb + (int64_t)lane * SPLIT;
                gml = (gmine < ng) ? (a.gmask ? a.gmask[gmine] : 0xffffffffu) : 0u;
                glive = __ballot((gml & set_mask) != 0u);
            }
            const int bit = __ffsll(glive) - 1;
            glive &= glive - 1;
            it_g = gb + (int64_t)bit * SPLIT;
            it_rem = (uint32_t)__shfl((int)gml, bit, 64) & set_mask;
            it_ysel = (it_ysel + 1 == NB) ? 0 : it_ysel + 1;
            st.first = true;
        }
        st.g = it_g;
        st.o = __builtin_ctz(it_rem);
        it_rem &= it_rem - 1u;
        st.ysel = it_ysel;
        st.valid = true;
        return st;
    };
    // indices of a step's rows (my two rows of the X image, and of the dY image when the step opens a group).  A valid
    // step always issues its two index loads (clamped addresses), whatever the lanes' rows: the counted wait relies on it.
    auto load_idx = [&](const Step& st, int (&idx)[2], int (&yrow)[2]) {
        if (!st.valid) {
            idx[0] = idx[1] = yrow[0] = yrow[1] = -1;
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t pos = st.g * 32 + 16 * rh + 8 * i + rsub;
            const bool ok = pos < a.n_out;
            const int64_t pc = ok ? pos : a.n_out - 1;
            const int v = a.nbr[pc * a.K + st.o];
            idx[i] = ok ? v : -1;
            yrow[i] = -1;
            if (st.first) {
                const int yr = a.order ? a.order[pc] : (int)pc;
                yrow[i] = ok ? yr : -1;
            }
        }
    };
    auto issue = [&](const Step& st, int xbuf, const int (&idx)[2], const int (&yrow)[2]) {
        if (!st.valid) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t vo = (idx[i] >= 0 && cw < cbi) ? (uint32_t)idx[i] * (uint32_t)(a.cin * 4) + (uint32_t)(cw * 128 + slot * 16) : WG_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(Xs + xbuf * 2048 + cw * 1024 + (16 * rh + 8 * i) * 32), 16, vo, 0, 0, 0);
        }
        if (st.first) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t vo = (yrow[i] >= 0 && cw < cbo) ? (uint32_t)yrow[i] * (uint32_t)(a.cout * 4) + (uint32_t)(cw * 128 + slot * 16) : WG_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_ptr_t)(Ys + st.ysel * 2048 + cw * 1024 + (16 * rh + 8 * i) * 32), 16, vo, 0, 0, 0);
            }
        }
    };
    auto compute = [&](const Step& st, int xbuf) {
        if (!tile_live) return;
        const float* Ab = Xs + xbuf * 2048 + tm * 1024 + h * 32 + r;
        const float* Bb = Ys + st.ysel * 2048 + tn * 1024 + h * 32 + r;
        const int oi = st.o - k0;
#define PCC_WG_CASE(I)                                                                                              \
    case I: {                                                                                                       \
        _Pragma("unroll") for (int kp = 0; kp < 16; ++kp)                                                           \
            acc[I < O ? I : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ab[kp * 64], Bb[kp * 64], acc[I < O ? I : 0], 0, 0, 0); \
    } break;
        switch (oi) {
            PCC_WG_CASE(0) PCC_WG_CASE(1) PCC_WG_CASE(2) PCC_WG_CASE(3) PCC_WG_CASE(4)
            PCC_WG_CASE(5) PCC_WG_CASE(6) PCC_WG_CASE(7) PCC_WG_CASE(8)
            default: break;
        }
#undef PCC_WG_CASE
    };

    int idxA[2], yrowA[2], idxB[2], yrowB[2];
    if constexpr (AHEAD == 1) {
        Step s0 = next_step();
        load_idx(s0, idxA, yrowA);
        issue(s0, 0, idxA, yrowA);                 // (waits for its own index loads)
        Step s1 = next_step();
        load_idx(s1, idxA, yrowA);
        __syncthreads();
        int xbuf = 0;
        while (s0.valid) {
            issue(s1, xbuf ^ 1, idxA, yrowA);
            Step s2 = next_step();
            load_idx(s2, idxB, yrowB);
            compute(s0, xbuf);
            __syncthreads();                       // vmcnt(0): step s1's images and step s2's indices have landed; s0's images are free
            s0 = s1;
            s1 = s2;
#pragma unroll
            for (int i = 0; i < 2; ++i) { idxA[i] = idxB[i]; yrowA[i] = yrowB[i]; }
            xbuf ^= 1;
        }
    } else {
        Step s0 = next_step();
        load_idx(s0, idxA, yrowA);
        issue(s0, 0, idxA, yrowA);
        Step s1 = next_step();
        load_idx(s1, idxA, yrowA);
        issue(s1, 1, idxA, yrowA);
        Step s2 = next_step();
        load_idx(s2, idxA, yrowA);                 // stays in registers until its gathers are issued in the loop
        int xb0 = 0;                               // image of s0; s1 = xb0 + 1, s2 = xb0 + 2 (mod 3)
        while (s0.valid) {
            Step s3 = next_step();
            load_idx(s3, idxB, yrowB);
            // all but the newest four memory operations have landed: newer than s0's gathers are s2's index loads (older
            // than the four), s1's gathers (>= 2) and s3's index loads (2)
            if (s1.valid && s3.valid) __builtin_amdgcn_s_waitcnt(0x0074);
            else __builtin_amdgcn_s_waitcnt(0x0070);
            __builtin_amdgcn_s_barrier();
            issue(s2, xb0 == 0 ? 2 : xb0 - 1, idxA, yrowA);
            compute(s0, xb0);
            s0 = s1;
            s1 = s2;
            s2 = s3;
#pragma unroll
            for (int i = 0; i < 2; ++i) { idxA[i] = idxB[i]; yrowA[i] = yrowB[i]; }
            xb0 = (xb0 == 2) ? 0 : xb0 + 1;
        }
    }

    if (!tile_live) return;
#pragma unroll
    for (int o = 0; o < O; ++o) {
        if (k0 + o >= a.K) break;
        float* P = a.partial + ((int64_t)s * a.K + (k0 + o)) * a.cin * a.cout;
        const int co = tn * 32 + r;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int ci = tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            P[(int64_t)ci * a.cout + co] = acc[o][reg];
        }
    }
#endif
}

// bf16 operands (pcc_conv_wgrad_bf16; cin, cout multiples of 64): X and dY are bf16, products accumulate in fp32 on
// v_mfma_f32_32x32x16_bf16.  The GEMM's K dimension is the ROWS, but the LDS images are [row][64 channels]
// (128-B rows, what the 16-B-per-lane LDS-DMA produces), so a lane's 8 consecutive k-values of one channel sit
// 128 B apart.  gfx950's transposing LDS read does the operand transpose: ds_read_b64_tr_b16 takes, per 16-lane
// group, a block of 4 rows x 16 columns (lane 4q+p addresses row q, columns 4p..4p+3) and hands lane i the 4 rows
// of column i — two of them make one MFMA operand (8 two-byte reads plus packing without it: 2.2x slower).
// Rows 128 B apart put rows q and q+2 on the same banks: the reads are 2-way conflicted, which hides under the
// gathers the kernel is bound by.
// A workgroup owns offset k, a block of up to 128 x 128 of W[k] = TM x TN tiles of 32 x 32 dealt round-robin to the
// four waves (128 x 128: a wave holds one column of 4 tiles and reads its dY operand once), and every SPLIT-th
// group of 32 rows (two MFMA k-steps).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(const WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* As = reinterpret_cast<unsigned short*>(smem);             // [2 chunks][32 rows][64 ch]
    unsigned short* Bs = As + 2 * 2048;                                       // [2 chunks][32 rows][64 co]
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    const int SPLIT = a.split;
    const int k = wgrad_offset_of(blockIdx.x / SPLIT, a.K), s = blockIdx.x % SPLIT;
    const int cin0 = blockIdx.y * 128, cout0 = blockIdx.z * 128;
    const int cbi = min(2, (a.cin - cin0) / 64), cbo = min(2, (a.cout - cout0) / 64);     // 64-channel chunks present
    const int TM = 2 * cbi, TN = 2 * cbo, NTILES = TM * TN;
    // 64 x 64 blocks fill one image slot per operand: the other slot takes a SECOND row group per iteration (waves
    // 0 / 2 stage the first group's X / dY, waves 1 / 3 the second's), doubling the MFMA work per barrier pair
    const bool two_groups = cbi == 1 && cbo == 1;

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;

    __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fin), 0, (int)(uint32_t)(a.n_in * a.cin * 2), WG_FLAGS);
    __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)(uint32_t)(a.n_out * a.cout * 2), WG_FLAGS);
    const int64_t ng = (a.n_out + 31) >> 5;
    const int slot = lane & 7, rsub = lane >> 3;
    // staging roles: waves 0-1 the X chunks 0-1, waves 2-3 the dY chunks 0-1
    const bool stage_x = wave_u < 2;
    const int cw = wave_u & 1;
    // my tiles: t = wave, wave + 4, ...; tile t -> (m = t / TN, n = t % TN)
    int tm[4], tn[4];
    bool live[4];
    // transposing read: my address inside a 4-row x 16-column block, and the block's first column within the tile
    const int tr_off = (((lane & 15) >> 2) * 64) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tt = wave_u + 4 * i;
        live[i] = tt < NTILES;
        tm[i] = live[i] ? tt / TN : 0;
        tn[i] = live[i] ? tt % TN : 0;
    }

    for (int64_t gbase = s; gbase < ng; gbase += 64 * SPLIT) {
      const int64_t gmine = gbase + (int64_t)lane * SPLIT;
      const uint32_t gml = (gmine < ng) ? (a.gmask ? a.gmask[gmine] : 0xffffffffu) : 0u;
      unsigned long long todo = __ballot((gml >> k) & 1u);
      while (todo) {
        const int bit = __ffsll(todo) - 1;
        todo &= todo - 1;
        int64_t g = gbase + (int64_t)bit * SPLIT;
        bool second = false;
        if (two_groups) {
            int64_t g2 = -1;
            if (todo) {
                g2 = gbase + (int64_t)(__ffsll(todo) - 1) * SPLIT;
                todo &= todo - 1;
            }
            second = g2 >= 0;
            if (cw == 1) g = g2;                           // waves 1 and 3 stage the second group (chunk 0 of it)
        }
        const int chunk = two_groups ? 0 : cw;
        uint32_t vo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t pos = g * 32 + 8 * i + rsub;
            const bool ok = g >= 0 && pos < a.n_out;
            if (stage_x) {
                const int idx = ok ? a.nbr[pos * a.K + k] : -1;
                vo[i] = (idx >= 0 && chunk < cbi) ? (uint32_t)idx * (uint32_t)(a.cin * 2) + (uint32_t)((cin0 + chunk * 64) * 2 + slot * 16) : WG_OOB;
            } else {
                const int64_t row = ok ? (a.order ? a.order[pos] : pos) : 0;
                vo[i] = (ok && chunk < cbo) ? (uint32_t)row * (uint32_t)(a.cout * 2) + (uint32_t)((cout0 + chunk * 64) * 2 + slot * 16) : WG_OOB;
            }
        }
        unsigned short* dst = (stage_x ? As : Bs) + cw * 2048;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (stage_x) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(dst + i * 512), 16, vo[i], 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_ptr_t)(dst + i * 512), 16, vo[i], 0, 0, 0);
        }
        __syncthreads();
        const int nks = (two_groups && second) ? 4 : 2;
        for (int ks4 = 0; ks4 < nks; ++ks4) {                  // rows 16 ks + 8 h + j of group slot gs
            const int ks = ks4 & 1, gs = ks4 >> 1;
            const int row0 = 16 * ks + 8 * h;                   // first of my 8 rows (k-values) in this MFMA step
            bf16x8 bop;
            int bn_loaded = -1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!live[i]) continue;                         // wave-uniform: EXEC stays all ones for the transposing reads
                const unsigned short* ap = As + ((tm[i] >> 1) + gs) * 2048 + row0 * 64 + (tm[i] & 1) * 32 + tr_off;
                const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap));
                const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap + 4 * 64));
                const s16x8 a8 = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
                if (tn[i] != bn_loaded) {                      // 128 x 128: all of the wave's tiles share n
                    const unsigned short* bp = Bs + ((tn[i] >> 1) + gs) * 2048 + row0 * 64 + (tn[i] & 1) * 32 + tr_off;
                    const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(bp));
                    const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(bp + 4 * 64));
                    const s16x8 b8 = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
                    bop = __builtin_bit_cast(bf16x8, b8);
                    bn_loaded = tn[i];
                }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a8), bop, acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
      }
    }

    float* P = a.partial + ((int64_t)s * a.K + k) * a.cin * a.cout;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!live[i]) continue;
        const int co = cout0 + tn[i] * 32 + r;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int ci = cin0 + tm[i] * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            P[(int64_t)ci * a.cout + co] = acc[i][reg];
        }
    }
#endif
}

// A pipelined form of this kernel (two or three image sets, the gathers of the next one / two row groups issued before the
// MFMAs of the current one, one barrier per group — the slice kernel's loop shape) was built and measured in round 2: SLOWER,
// 850 k-row shell 128 x 128: 264 TFLOP/s single-buffered, 240 with two image sets, 191 with three (64 x 64: 126 / 116 / 91).
// At 4.1 TB/s of 256-byte gathers the kernel sits at the gather rate of the memory system; what keeps that rate up is the
// number of workgroups per CU (16 KB of LDS each), which every extra image set cuts.
// 64 -> 64 with bf16 operands and 27 offsets: a row pair is 2 x 128 B for 8 k FLOP and the kernel above runs at the gather rate
// of the memory system (~4 TB/s of rows), so the only lever is bytes: a workgroup owns O consecutive offsets and a group's
// 32 dY rows are staged ONCE for all of the O that occur in the group (4 + 4 L KB per L live offsets instead of 8 L), each
// offset's X rows in its own image.  Single-buffered like the kernel above (occupancy carries the latency): one staging
// instruction per thread and image — wave w stages rows 8 w .. 8 w + 7 of every image — then wave (tm, tn) runs its 32 x 32
// tile of every live offset against the shared dY operand: per MFMA k-step one dY fragment, L X fragments, L MFMAs.
template <int O>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_slice_kernel(const WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* Xs = reinterpret_cast<unsigned short*>(smem);             // [O][32 rows][64 ch]
    unsigned short* Ys = Xs + O * 2048;                                       // [32 rows][64 co]
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    const int SPLIT = a.split;
    const int set = blockIdx.x / SPLIT, s = blockIdx.x % SPLIT;
    const int k0 = set * O;
    const uint32_t set_mask = (((1u << O) - 1u) << k0) & ((1u << a.K) - 1u);
    const int tm = wave_u >> 1, tn = wave_u & 1;

    f32x16 acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[o][j] = 0.0f;

    __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fin), 0, (int)(uint32_t)(a.n_in * a.cin * 2), WG_FLAGS);
    __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)(uint32_t)(a.n_out * a.cout * 2), WG_FLAGS);
    const int64_t ng = (a.n_out + 31) >> 5;
    const int slot = lane & 7, rsub = lane >> 3;
    const int tr_off = (((lane & 15) >> 2) * 64) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    for (int64_t gbase = s; gbase < ng; gbase += 64 * (int64_t)SPLIT) {
        const int64_t gmine = gbase + (int64_t)lane * SPLIT;
        const uint32_t gml = (gmine < ng) ? (a.gmask ? a.gmask[gmine] : 0xffffffffu) : 0u;
        unsigned long long todo = __ballot((gml & set_mask) != 0u);
        while (todo) {
            const int bit = __ffsll(todo) - 1;
            todo &= todo - 1;
            const int64_t g = gbase + (int64_t)bit * SPLIT;
            const uint32_t rem = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)__shfl((int)gml, bit, 64) & set_mask) >> k0));
            const int64_t pos = g * 32 + 8 * wave_u + rsub;
            const bool ok = pos < a.n_out;
            {
                const int yrow = ok ? (a.order ? a.order[pos] : (int)pos) : -1;
                const uint32_t vo = yrow >= 0 ? (uint32_t)yrow * (uint32_t)(a.cout * 2) + (uint32_t)(slot * 16) : WG_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_ptr_t)(Ys + wave_u * 512), 16, vo, 0, 0, 0);
            }
#pragma unroll
            for (int o = 0; o < O; ++o) {
                if (!((rem >> o) & 1u)) continue;
                const int idx = ok ? a.nbr[pos * a.K + k0 + o] : -1;
                const uint32_t vo = idx >= 0 ? (uint32_t)idx * (uint32_t)(a.cin * 2) + (uint32_t)(slot * 16) : WG_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(Xs + o * 2048 + wave_u * 512), 16, vo, 0, 0, 0);
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int row0 = 16 * ks + 8 * h;
                const unsigned short* bp = Ys + row0 * 64 + tn * 32 + tr_off;
                const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(bp));
                const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(bp + 4 * 64));
                const s16x8 b8 = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
                const bf16x8 bop = __builtin_bit_cast(bf16x8, b8);
#pragma unroll
                for (int o = 0; o < O; ++o) {
                    if (!((rem >> o) & 1u)) continue;                       // wave-uniform: EXEC stays all ones
                    const unsigned short* ap = Xs + o * 2048 + row0 * 64 + tm * 32 + tr_off;
                    const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap));
                    const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap + 4 * 64));
                    const s16x8 a8 = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
                    acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a8), bop, acc[o], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }

#pragma unroll
    for (int o = 0; o < O; ++o) {
        if (k0 + o >= a.K) break;
        float* P = a.partial + ((int64_t)s * a.K + (k0 + o)) * a.cin * a.cout;
        const int co = tn * 32 + r;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int ci = tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            P[(int64_t)ci * a.cout + co] = acc[o][reg];
        }
    }
#endif
}

// thin shapes (cin or cout not a multiple of 32: q-map branches, input layer, narrow heads): cin * cout <= 4096.
// One workgroup per (offset, split).  With >= 256 (ci, co) pairs a thread owns up to 16 pairs and walks the
// split's rows; with fewer pairs (64 -> 1: the occupancy head on 5 M candidates) the spare threads take
// every RL-th row instead and the row lanes are summed through LDS at the end.
__global__ __launch_bounds__(256) void conv_wgrad_thin_kernel(const WgradArgs a) {
    __shared__ float red[256];
    const int k = blockIdx.x / WG_SPLIT_THIN, s = blockIdx.x % WG_SPLIT_THIN;
    const int pairs = a.cin * a.cout;
    int pp = 1;
    while (pp < pairs && pp < 256) pp <<= 1;              // pairs rounded up to a power of two, at most 256
    const int RL = 256 / pp;                                // row lanes (1 when pairs >= 256)
    const int pid = threadIdx.x % pp, rl = threadIdx.x / pp;
    float acc[16];
    int ci[16], co[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        acc[q] = 0.0f;
        const int e = min(pid + 256 * q, pairs - 1);
        ci[q] = e / a.cout;
        co[q] = e - ci[q] * a.cout;
    }
    const int64_t per = (a.n_out + WG_SPLIT_THIN - 1) / WG_SPLIT_THIN;
    const int64_t lo = s * per, hi = min(a.n_out, lo + per);
    for (int64_t pos = lo + rl; pos < hi; pos += RL) {
        const int idx = a.nbr[pos * a.K + k];
        if (idx < 0) continue;
        const int64_t row = a.order ? a.order[pos] : pos;
        const float* x = a.fin + (int64_t)idx * a.cin;
        const float* y = a.dy + row * a.cout;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (256 * q < pairs) acc[q] = fmaf(x[ci[q]], y[co[q]], acc[q]);
        }
    }
    float* P = a.partial + ((int64_t)s * a.K + k) * pairs;
    if (RL > 1) {                                           // pairs < 256: only acc[0] is in use
        red[threadIdx.x] = acc[0];
        __syncthreads();
        if (rl == 0 && pid < pairs) {
            float sum = 0.0f;
            for (int l = 0; l < RL; ++l) sum += red[l * pp + pid];        // fixed order
            P[pid] = sum;
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = threadIdx.x + 256 * q;
        if (e < pairs) P[e] = acc[q];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int64_t elems, int nsplit,
                                                           float* __restrict__ dw) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    float sum = 0.0f;
    for (int s = 0; s < nsplit; ++s) sum += partial[(int64_t)s * elems + e];        // fixed order
    dw[e] = sum;
}

// the same sums, four elements per thread with 16-byte loads (elems % 4 == 0, 16-byte aligned partials: the MFMA shapes)
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float4* __restrict__ partial, int64_t elems4, int nsplit,
                                                            float4* __restrict__ dw) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems4) return;
    float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int s = 0; s < nsplit; ++s) {                                              // fixed order
        const float4 v = partial[(int64_t)s * elems4 + e];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    dw[e] = sum;
}

}  // namespace pcc

using namespace pcc;

extern "C" {

int pcc_kernel_map_transpose(const int32_t* nbr, int64_t n_out, int32_t K, int64_t n_in, int32_t* nbr_t, uint32_t* row_mask_t,
                             void* stream) {
    PCC_REQUIRE(K >= 1 && K <= 27, "pcc_kernel_map_transpose: K out of range");
    PCC_REQUIRE(n_in < (1ll << 31) && n_out < (1ll << 31), "pcc_kernel_map_transpose: too many rows");
    hipStream_t st = as_stream(stream);
    if (n_in > 0) {
        PCC_CHECK_HIP(hipMemsetAsync(nbr_t, 0xFF, (size_t)n_in * K * sizeof(int32_t), st));
        PCC_CHECK_HIP(hipMemsetAsync(row_mask_t, 0, (size_t)n_in * sizeof(uint32_t), st));
    }
    if (n_out <= 0 || n_in <= 0) return PCC_OK;
    hipLaunchKernelGGL(map_transpose_kernel, dim3(blocks_for(n_out * K, 256)), dim3(256), 0, st, nbr, n_out, K, nbr_t, row_mask_t);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

static inline void launch_wgrad_reduce(const float* partial, int64_t elems, int nsplit, float* dw, hipStream_t st) {
    if (elems % 4 == 0 && ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(dw)) & 15) == 0)
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(blocks_for(elems / 4, 256)), dim3(256), 0, st,
                           reinterpret_cast<const float4*>(partial), elems / 4, nsplit, reinterpret_cast<float4*>(dw));
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks_for(elems, 256)), dim3(256), 0, st, partial, elems, nsplit, dw);
}
static inline bool wgrad_mfma(int cin, int cout) { return cin % 32 == 0 && cout % 32 == 0; }
static inline bool wgrad_rowsplit(int cin, int cout) { return wgrad_mfma(cin, cout) && cin <= 64 && cout <= 64; }
static inline int wgrad_splits(int64_t n_out) {            // at least ~48 row groups per workgroup
    const int64_t want = ((n_out + 31) / 32) / 48;
    return (int)(want < 8 ? 8 : (want > WG_SPLIT_MAX ? WG_SPLIT_MAX : want));
}
// slice kernel (nine offsets = one dz plane per workgroup, dY rows staged once per group): the default for 64-channel-class
// blocks of a 27-offset kernel; PCC_WGRAD_SLICE=0 keeps the one-offset kernel for A/B runs
static inline bool wgrad_slice(int cin, int cout, int K) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("PCC_WGRAD_SLICE"); on = (e && e[0] == '0') ? 0 : 1; }
    return on && wgrad_rowsplit(cin, cout) && K == 27;
}
constexpr int WG_SPLIT_SLICE_MAX = 256;
static inline int wgrad_slice_splits(int64_t n_out) {      // three workgroup sets only: more splits to fill 256 CUs x 4
    const int64_t want = ((n_out + 31) / 32) / 24;
    return (int)(want < 8 ? 8 : (want > WG_SPLIT_SLICE_MAX ? WG_SPLIT_SLICE_MAX : want));
}
static inline int wgrad_partials(int cin, int cout, int split, int K) {
    if (!wgrad_mfma(cin, cout)) return WG_SPLIT_THIN;
    if (wgrad_slice(cin, cout, K)) return split;
    return split * (wgrad_rowsplit(cin, cout) ? 4 : 1);
}

int64_t pcc_conv_wgrad_scratch_elems(int32_t K, int32_t cin, int32_t cout) {
    return (int64_t)wgrad_partials(cin, cout, WG_SPLIT_MAX, 0) * K * cin * cout;      // the larger of the two layouts
}

int pcc_conv_wgrad(const float* fin, int64_t n_in, int32_t cin, const float* dy, int64_t n_out, int32_t cout, const int32_t* nbr,
                   const int32_t* order, const uint32_t* group_mask32, int32_t K, float* dw, float* scratch, int64_t scratch_elems,
                   void* stream) {
    PCC_REQUIRE(K >= 1 && K <= 27 && cin >= 1 && cout >= 1, "pcc_conv_wgrad: bad shape");
    PCC_REQUIRE(nbr != nullptr, "pcc_conv_wgrad: neighbour table required (kernel_size 1: pass the identity map)");
    PCC_REQUIRE(scratch_elems >= pcc_conv_wgrad_scratch_elems(K, cin, cout), "pcc_conv_wgrad: scratch too small");
    hipStream_t st = as_stream(stream);
    const int64_t elems = (int64_t)K * cin * cout;
    if (n_out <= 0) {
        PCC_CHECK_HIP(hipMemsetAsync(dw, 0, (size_t)elems * sizeof(float), st));
        return PCC_OK;
    }
    WgradArgs a;
    a.fin = fin; a.dy = dy; a.nbr = nbr; a.order = order; a.gmask = group_mask32; a.partial = scratch;
    a.n_in = n_in; a.n_out = n_out; a.cin = cin; a.cout = cout; a.K = K; a.split = wgrad_splits(n_out);
    if (cin % 32 == 0 && cout % 32 == 0) {
        PCC_REQUIRE((uint64_t)n_in * cin * 4 <= WG_OOB && (uint64_t)n_out * cout * 4 <= WG_OOB,
                    "pcc_conv_wgrad: operands of 4 GiB and more are not supported yet");
        PCC_REQUIRE(((reinterpret_cast<uintptr_t>(fin) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0,
                    "pcc_conv_wgrad: fin and dy must be 16-byte aligned (16-byte LDS-DMA loads)");
        if (wgrad_slice(cin, cout, K)) {
            static int O = -1;       // PCC_WGRAD_SLICE_O=3|5|9: offsets per workgroup (A/B; accumulators = 16 O registers per lane)
            if (O < 0) { const char* e = getenv("PCC_WGRAD_SLICE_O"); O = e ? atoi(e) : 5; }
            a.split = wgrad_slice_splits(n_out);
            static int ahead = -1;   // PCC_WGRAD_AHEAD=1|2: prefetch distance of the gathers (A/B)
            if (ahead < 0) { const char* e = getenv("PCC_WGRAD_AHEAD"); ahead = (e && e[0] == '1') ? 1 : 2; }
#define PCC_SLICE(OO)                                                                                                              \
    do {                                                                                                                           \
        const dim3 grid((unsigned)(((27 + OO - 1) / OO) * a.split));                                                               \
        if (ahead == 1) hipLaunchKernelGGL((conv_wgrad_slice_kernel<OO, 1>), grid, dim3(256), 2 * 2 * 2048 * sizeof(float), st, a); \
        else hipLaunchKernelGGL((conv_wgrad_slice_kernel<OO, 2>), grid, dim3(256), 2 * 3 * 2048 * sizeof(float), st, a);            \
    } while (0)
            if (O == 3) PCC_SLICE(3);
            else if (O == 4) PCC_SLICE(4);
            else if (O == 6) PCC_SLICE(6);
            else if (O == 9) PCC_SLICE(9);
            else PCC_SLICE(5);
#undef PCC_SLICE
        } else {
            const dim3 grid((unsigned)(K * a.split), (unsigned)((cin + 127) / 128), (unsigned)((cout + 127) / 128));
            hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), 8 * 1024 * sizeof(float), st, a);
        }
    } else {
        PCC_REQUIRE((int64_t)cin * cout <= 4096, "pcc_conv_wgrad: thin path handles cin * cout <= 4096 (got %d x %d)", cin, cout);
        hipLaunchKernelGGL(conv_wgrad_thin_kernel, dim3((unsigned)(K * WG_SPLIT_THIN)), dim3(256), 0, st, a);
    }
    launch_wgrad_reduce(scratch, elems, wgrad_partials(cin, cout, a.split, K), dw, st);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

int pcc_conv_wgrad_bf16(const uint16_t* fin, int64_t n_in, int32_t cin, const uint16_t* dy, int64_t n_out, int32_t cout,
                        const int32_t* nbr, const int32_t* order, const uint32_t* group_mask32, int32_t K, float* dw, float* scratch,
                        int64_t scratch_elems, void* stream) {
    PCC_REQUIRE(K >= 1 && K <= 27, "pcc_conv_wgrad_bf16: bad K");
    PCC_REQUIRE(cin % 64 == 0 && cout % 64 == 0 && cin >= 64 && cout >= 64, "pcc_conv_wgrad_bf16: cin and cout must be multiples of 64 (got %d, %d)", cin, cout);
    PCC_REQUIRE(nbr != nullptr, "pcc_conv_wgrad_bf16: neighbour table required");
    PCC_REQUIRE((uint64_t)n_in * cin * 2 <= WG_OOB && (uint64_t)n_out * cout * 2 <= WG_OOB, "pcc_conv_wgrad_bf16: operands of 4 GiB and more are not supported");
    PCC_REQUIRE(((reinterpret_cast<uintptr_t>(fin) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0,
                "pcc_conv_wgrad_bf16: fin and dy must be 16-byte aligned (16-byte LDS-DMA loads)");
    hipStream_t st = as_stream(stream);
    const int64_t elems = (int64_t)K * cin * cout;
    if (n_out <= 0) {
        PCC_CHECK_HIP(hipMemsetAsync(dw, 0, (size_t)elems * sizeof(float), st));
        return PCC_OK;
    }
    WgradArgs a;
    a.fin = reinterpret_cast<const float*>(fin); a.dy = reinterpret_cast<const float*>(dy); a.nbr = nbr; a.order = order;
    a.gmask = group_mask32; a.partial = scratch; a.n_in = n_in; a.n_out = n_out; a.cin = cin; a.cout = cout; a.K = K;
    a.split = wgrad_splits(n_out);
    PCC_REQUIRE(scratch_elems >= (int64_t)a.split * elems, "pcc_conv_wgrad_bf16: scratch too small");
    // 64 -> 64, 27 offsets: the slice kernel (PCC_WGRAD_BF16_SLICE_O=0 keeps the one-offset kernel; 3 | 5 | 9 offsets per
    // workgroup).  850 k-row / 265 k-row shells, same box: one-offset 122-124 / 129 TFLOP/s, O = 3 137 / 135 (71 registers, 7 waves
    // per SIMD), O = 5 121 / 119 (103 registers), O = 9 140 / 125 (167 registers): fewer bytes and fewer waves trade evenly
    static int slice_o = -1;
    if (slice_o < 0) {
        const char* e = getenv("PCC_WGRAD_BF16_SLICE_O");
        slice_o = e ? atoi(e) : 3;
        if (slice_o != 0 && slice_o != 3 && slice_o != 5 && slice_o != 9) slice_o = 3;
    }
    if (slice_o && cin == 64 && cout == 64 && K == 27) {
        a.split = wgrad_slice_splits(n_out);
        PCC_REQUIRE(scratch_elems >= (int64_t)a.split * elems, "pcc_conv_wgrad_bf16: scratch too small");
        const unsigned sets = (unsigned)((K + slice_o - 1) / slice_o);
        const size_t lds = (size_t)(slice_o + 1) * 2048 * sizeof(unsigned short);
        if (slice_o == 3) hipLaunchKernelGGL(conv_wgrad_bf16_slice_kernel<3>, dim3(sets * a.split), dim3(256), lds, st, a);
        else if (slice_o == 5) hipLaunchKernelGGL(conv_wgrad_bf16_slice_kernel<5>, dim3(sets * a.split), dim3(256), lds, st, a);
        else hipLaunchKernelGGL(conv_wgrad_bf16_slice_kernel<9>, dim3(sets * a.split), dim3(256), lds, st, a);
        launch_wgrad_reduce(scratch, elems, a.split, dw, st);
        PCC_LAUNCH_CHECK();
        return PCC_OK;
    }
    const dim3 grid((unsigned)(K * a.split), (unsigned)((cin + 127) / 128), (unsigned)((cout + 127) / 128));
    hipLaunchKernelGGL(conv_wgrad_bf16_kernel, grid, dim3(256), 4 * 2048 * sizeof(unsigned short), st, a);
    launch_wgrad_reduce(scratch, elems, a.split, dw, st);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
}

}  // extern "C"
