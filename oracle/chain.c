/* Oracle, order-matched mode: sparse convolution as ONE fused multiply-add chain per output element.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/__init__.py).  The default oracle (oracle/nn.py) sums a convolution the way a CPU
 * BLAS does: per kernel offset a gather -> sgemm -> index_add_.  That is an independent restatement of
 *     out[j] = bias + sum_k in[nbr(j, k)] @ W[k]          (MinkowskiEngine; reference call sites model/transforms.py:35-57,
 *                                                           model/blocks.py:29-53,152-181, model/entropy_models.py:284-306)
 * but fp32 addition is not associative, so the product (csrc/conv.hip) and that oracle agree to ~1e-7 relative and no
 * closer, and every discrete decision downstream (a latent on a rounding boundary, a scale on a table boundary, a top-k
 * near-tie) can fall on different sides.  This file states the sum in the ORDER the product's kernels document for
 * themselves (csrc/conv.hip:1-16, 689-716, 1109-1162), as plain scalar arithmetic:
 *
 *     acc = +0
 *     for k ascending, neighbour present:            (an absent neighbour contributes nothing)
 *         for the input channels in `visit` order:
 *             acc = fmaf(in[nbr(j, k)][ci], W[k][ci][co], acc)        -- one rounding per step
 *
 * `visit` is 0, 1, 2, ... for the thin kernels (conv_thin_kernel, and im2col_thin + MFMA which is built to equal it), and
 * within every group of 8 channels 0, 4, 1, 5, 2, 6, 3, 7 for the MFMA kernels (lane halves h = 0 / 1 of
 * v_mfma_f32_32x32x2_f32 hold channels 8 kk + s and 8 kk + 4 + s of sub-block kk, s = 0..3: csrc/conv.hip:598-608,689-716;
 * the hardware instruction is the same chain of fused multiply-adds, k = 0 then k = 1: tools/micro/mfma_shapes_bitwise.hip).
 * With it the oracle's latents, streams and decoded voxel sets can be compared with the product's for EQUALITY
 * (tests/test_exact_parity.py); the BLAS-order oracle stays the independent check at 1e-3 dB.
 *
 * Vectorised over output channels only (8 independent chains per AVX2 register: lanes never mix), rows six at a time
 * so that a weight vector is loaded once per 6 chains.  The padding of a short group of six is a row of zeros:
 * fmaf(0, w, acc) == acc exactly for finite w and an accumulator that started at +0 (it can never be -0), which is
 * also how the product's 32-row tiles treat a row that lacks an offset its tile has.
 * Build: oracle/Makefile (gcc -O2 -mavx2 -mfma -ffp-contract=off -pthread).
 */
#include <immintrin.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RB 6        /* rows per register block */
#define CB 16       /* output channels per register block (2 x __m256) */
#define WIN 96      /* output rows per window */

typedef struct {
    const float* fin;
    const float* wp;          /* packed [K][strips][cin (visit order)][CB], zero padded columns */
    const int64_t* nbr;       /* [n_out][K] or NULL (K = 1, identity) */
    const int32_t* visit;     /* [cin] channel visited at step t */
    const float* zero_row;    /* [cin] zeros */
    float* out;               /* [n_out][cout] */
    int64_t n_out;
    int cin, cout, K, strips;
    int64_t next;             /* atomic cursor over row blocks */
} job_t;

/* One window of up to WIN consecutive output rows.  Per kernel offset the window's rows that HAVE the offset are taken six
 * at a time (a short last group is padded with a row of zeros whose accumulator is thrown away), so hardly any chain step
 * multiplies zeros; between two offsets a row's 16 partial sums wait in a small buffer.  Per element: k ascending, one
 * contiguous run of fused multiply-adds per present offset, continuing from the stored partial sum — the chain above. */
static void window_rows(const job_t* J, int64_t r0, int nr) {
    const int cin = J->cin, cout = J->cout, K = J->K;
    uint8_t list[32][WIN];              /* rows of the window that have offset k (K <= 27 on this path, 32 for safety) */
    int cnt[32];
    for (int k = 0; k < K; ++k) {
        int c = 0;
        for (int r = 0; r < nr; ++r) {
            const int64_t idx = J->nbr ? J->nbr[(r0 + r) * K + k] : (r0 + r);
            if (idx >= 0) list[k][c++] = (uint8_t)r;
        }
        cnt[k] = c;
    }
    float accbuf[WIN + 1][CB] __attribute__((aligned(32)));        /* slot WIN: the padding rows' scratch */
    for (int s = 0; s < J->strips; ++s) {
        memset(accbuf, 0, sizeof(accbuf));
        for (int k = 0; k < K; ++k) {
            const float* w = J->wp + ((int64_t)k * J->strips + s) * cin * CB;
            const int32_t* visit = J->visit;
            for (int g = 0; g < cnt[k]; g += RB) {
                const float* x[RB];
                float* a[RB];
                for (int i = 0; i < RB; ++i) {
                    if (g + i < cnt[k]) {
                        const int r = list[k][g + i];
                        const int64_t idx = J->nbr ? J->nbr[(r0 + r) * K + k] : (r0 + r);
                        x[i] = J->fin + idx * cin;
                        a[i] = accbuf[r];
                    } else {
                        x[i] = J->zero_row;
                        a[i] = accbuf[WIN];
                    }
                }
                /* 12 named accumulators: they must live in registers (an indexed array of vectors goes to the stack) */
                __m256 a00 = _mm256_load_ps(a[0]), a01 = _mm256_load_ps(a[0] + 8), a10 = _mm256_load_ps(a[1]), a11 = _mm256_load_ps(a[1] + 8);
                __m256 a20 = _mm256_load_ps(a[2]), a21 = _mm256_load_ps(a[2] + 8), a30 = _mm256_load_ps(a[3]), a31 = _mm256_load_ps(a[3] + 8);
                __m256 a40 = _mm256_load_ps(a[4]), a41 = _mm256_load_ps(a[4] + 8), a50 = _mm256_load_ps(a[5]), a51 = _mm256_load_ps(a[5] + 8);
                const float *x0 = x[0], *x1 = x[1], *x2 = x[2], *x3 = x[3], *x4 = x[4], *x5 = x[5];
                for (int t = 0; t < cin; ++t) {
                    const int ci = visit[t];
                    const __m256 w0 = _mm256_load_ps(w + (int64_t)t * CB), w1 = _mm256_load_ps(w + (int64_t)t * CB + 8);
                    __m256 xv;
#define STEP(X, A0, A1) xv = _mm256_broadcast_ss((X) + ci); A0 = _mm256_fmadd_ps(xv, w0, A0); A1 = _mm256_fmadd_ps(xv, w1, A1)
                    STEP(x0, a00, a01); STEP(x1, a10, a11); STEP(x2, a20, a21);
                    STEP(x3, a30, a31); STEP(x4, a40, a41); STEP(x5, a50, a51);
#undef STEP
                }
                _mm256_store_ps(a[0], a00); _mm256_store_ps(a[0] + 8, a01); _mm256_store_ps(a[1], a10); _mm256_store_ps(a[1] + 8, a11);
                _mm256_store_ps(a[2], a20); _mm256_store_ps(a[2] + 8, a21); _mm256_store_ps(a[3], a30); _mm256_store_ps(a[3] + 8, a31);
                _mm256_store_ps(a[4], a40); _mm256_store_ps(a[4] + 8, a41); _mm256_store_ps(a[5], a50); _mm256_store_ps(a[5] + 8, a51);
            }
        }
        const int c0 = s * CB, nc = cout - c0 < CB ? cout - c0 : CB;
        for (int r = 0; r < nr; ++r) memcpy(J->out + (r0 + r) * cout + c0, accbuf[r], (size_t)nc * sizeof(float));
    }
}

static void* worker(void* arg) {
    job_t* J = (job_t*)arg;
    const int64_t nwin = (J->n_out + WIN - 1) / WIN;
    for (;;) {
        const int64_t b0 = __atomic_fetch_add(&J->next, 4, __ATOMIC_RELAXED);       /* 4 windows at a time */
        if (b0 >= nwin) break;
        const int64_t b1 = b0 + 4 < nwin ? b0 + 4 : nwin;
        for (int64_t b = b0; b < b1; ++b) {
            const int64_t r0 = b * WIN;
            window_rows(J, r0, (int)(J->n_out - r0 < WIN ? J->n_out - r0 : WIN));
        }
    }
    return NULL;
}

/* out[j][co] = the chain above, no bias (the caller adds it: the product's epilogue adds the bias to the finished
 * accumulator, csrc/conv.hip:861).  mfma_order != 0 needs cin % 8 == 0.  Returns 0, or -1 on a bad argument / allocation. */
int pcc_oracle_conv_chain(const float* fin, int64_t n_in, int32_t cin, const float* w, const int64_t* nbr, int64_t n_out,
                          int32_t K, int32_t cout, int32_t mfma_order, float* out, int32_t nthreads) {
    (void)n_in;
    if (cin <= 0 || cout <= 0 || K <= 0 || K > 32 || n_out < 0) return -1;
    if (mfma_order && cin % 8 != 0) return -1;
    if (!nbr && K != 1) return -1;
    if (n_out == 0) return 0;
    const int strips = (cout + CB - 1) / CB;
    int32_t* visit = (int32_t*)malloc(sizeof(int32_t) * (size_t)cin);
    float* zero_row = (float*)calloc((size_t)cin, sizeof(float));
    float* wp = (float*)aligned_alloc(32, sizeof(float) * (size_t)K * strips * cin * CB);
    if (wp) memset(wp, 0, sizeof(float) * (size_t)K * strips * cin * CB);
    if (!visit || !zero_row || !wp) { free(visit); free(zero_row); free(wp); return -1; }
    static const int V8[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    for (int t = 0; t < cin; ++t) visit[t] = mfma_order ? (t & ~7) + V8[t & 7] : t;
    for (int k = 0; k < K; ++k)
        for (int s = 0; s < strips; ++s)
            for (int t = 0; t < cin; ++t)
                for (int c = 0; c < CB; ++c) {
                    const int co = s * CB + c;
                    if (co < cout) wp[(((int64_t)k * strips + s) * cin + t) * CB + c] = w[((int64_t)k * cin + visit[t]) * cout + co];
                }
    job_t J = {fin, wp, nbr, visit, zero_row, out, n_out, cin, cout, K, strips, 0};
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if (n_out < 8 * WIN) nthreads = 1;
    pthread_t th[64];
    int started = 0;
    for (int i = 1; i < nthreads; ++i)
        if (pthread_create(&th[started], NULL, worker, &J) == 0) ++started;
    worker(&J);
    for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
    free(visit); free(zero_row); free(wp);
    return 0;
}

/* The plain scalar statement of the same chain (no blocking, no vectors, no threads): what the function above must
 * equal bit for bit (tests/test_oracle_chain.py), and the form to read. */
int pcc_oracle_conv_chain_scalar(const float* fin, int32_t cin, const float* w, const int64_t* nbr, int64_t n_out, int32_t K,
                                 int32_t cout, int32_t mfma_order, float* out) {
    static const int V8[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    if (mfma_order && cin % 8 != 0) return -1;
    if (!nbr && K != 1) return -1;
    for (int64_t j = 0; j < n_out; ++j)
        for (int co = 0; co < cout; ++co) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) {
                const int64_t idx = nbr ? nbr[j * K + k] : j;
                if (idx < 0) continue;
                for (int t = 0; t < cin; ++t) {
                    const int ci = mfma_order ? (t & ~7) + V8[t & 7] : t;
                    acc = __builtin_fmaf(fin[idx * cin + ci], w[((int64_t)k * cin + ci) * cout + co], acc);
                }
            }
            out[j * cout + co] = acc;
        }
    return 0;
}

/* Narrow heads (csrc/conv.hip:1206-1280): out[j][c] = sum_k ascending, neighbour present, of scores[nbr(j,k)][k*cout + c],
 * plain additions from +0; the caller adds the bias afterwards (gather_sum_kernel: acc + bias). */
int pcc_oracle_gather_sum(const float* scores, int32_t ld, const int64_t* nbr, int64_t n_out, int32_t K, int32_t cout, float* out) {
    for (int64_t j = 0; j < n_out; ++j)
        for (int c = 0; c < cout; ++c) {
            volatile float acc = 0.0f;          /* volatile: one rounding per addition, whatever the optimiser would like */
            for (int k = 0; k < K; ++k) {
                const int64_t idx = nbr[j * K + k];
                if (idx >= 0) acc = acc + scores[idx * ld + k * cout + c];
            }
            out[j * cout + c] = acc;
        }
    return 0;
}
