/*
 * libpcc_hip.so — C-ABI of the MI355X-native encode/decode operators of the joint
 * geometry+attribute point-cloud codec (ColorModel.compress / decompress / forward).
 *
 * The reference (/root/reference) has no FFI of its own: its operators are Python calls
 * into MinkowskiEngine (model/model.py:3, model/transforms.py:3, model/blocks.py:3,
 * model/entropy_models.py:5) and compressai (model/entropy_models.py:9-10).  Each entry
 * point below names the reference call site(s) it replaces.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers unless marked "host".
 *   - all buffers are caller-owned (torch tensors on the Python side); the library never
 *     allocates device memory.  `stream` is a hipStream_t passed as void*.
 *   - every function returns 0 on success or a negative PCC_ERR_* code; the message is
 *     available from pcc_last_error() (thread-local).  No exceptions cross the ABI.
 *   - coordinates: int32 [N,4] rows (batch, x, y, z), |x|,|y|,|z| <= PCC_COORD_LIMIT, batch <= PCC_BATCH_LIMIT.
 *   - features: fp32 row-major [N, C].  Neighbour tables: int32 [N_out, K], -1 = absent.
 *   - hash table = (keys uint64[cap], vals int32[cap], tensor_stride), cap a power of two from
 *     pcc_hash_capacity(); a table is immutable once built and may be read concurrently.
 *     `tensor_stride` is the grid pitch of the coordinate set the table indexes (every coordinate a
 *     multiple of it; 1 for arbitrary integer coordinates): the slot function keeps 8 grid steps
 *     along z in one cache line, so build and every later probe must name the same value (any
 *     value >= 1 is correct, the set's own stride is the fast one).  pcc_stride_map / pcc_children /
 *     pcc_kernel_map derive it from their `ts` / `step` arguments.
 */
#ifndef PCC_HIP_H
#define PCC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCC_OK 0
#define PCC_ERR_ARG (-1)
#define PCC_ERR_HIP (-2)
#define PCC_ERR_UNSUPPORTED (-3)
#define PCC_ERR_DATA (-4)

#define PCC_ACT_NONE 0
#define PCC_ACT_RELU 1       /* ME.MinkowskiReLU */
#define PCC_ACT_LEAKY_RELU 2 /* ME.MinkowskiLeakyReLU, slope 0.01 */

int pcc_version(void);
const char* pcc_last_error(void);
/* Number of devices visible to the library and a short name of device `dev` (host). */
int pcc_device_count(void);
int pcc_device_name(int dev, char* out, int out_len);

/* ---------------------------------------------------------------------------------------
 * Coordinate manager (ME coordinate_map_*; every ME.SparseTensor(...) construction, e.g.
 * model/model.py:59,64,122,188).
 * ------------------------------------------------------------------------------------- */
int64_t pcc_hash_capacity(int64_t n);

/* Coordinate range.  A voxel key has 10 bits of batch index and 18 bits per coordinate: coordinates must satisfy
 * |c| <= PCC_COORD_LIMIT and batch indices 0 <= b <= PCC_BATCH_LIMIT — the reference's radix-1e5 keys (model/blocks.py:118 and
 * utils.py:170: coordinates 0 .. 99,999) fit, with the same span on the negative side.  A set with a coordinate outside the
 * range is never hashed with an aliased key: pcc_stride_map / pcc_children check the source rows and every generated
 * coordinate and report *out_count = PCC_COUNT_ERR_RANGE instead of a row count — the error reaches the host with the one
 * value it reads anyway, at no extra synchronisation (the Python side raises ValueError). */
#define PCC_COORD_LIMIT 130000
#define PCC_BATCH_LIMIT 1022
#define PCC_COUNT_ERR_RANGE (-2)

/* Insert rows 0..n-1; table value = row index.  Duplicate coordinates keep the smallest
 * row index and increment *dup_count (device int32, may be NULL). */
int pcc_hash_build(const int32_t* coords, int64_t n, uint64_t* keys, int32_t* vals, int64_t cap,
                   int32_t tensor_stride, int32_t* dup_count, void* stream);

/* out_idx[i] = row of query i or -1.  (features_at_coordinates on on-grid integer queries:
 * model/transforms.py:96,124,262; model/blocks.py:37,50; model/entropy_models.py:326,364,401;
 * torch.isin of model/blocks.py:125.) */
int pcc_hash_lookup(const uint64_t* keys, const int32_t* vals, int64_t cap, int32_t tensor_stride,
                    const int32_t* query, int64_t nq, int32_t* out_idx, void* stream);

/* Scratch (int32 elements) needed by pcc_stride_map / pcc_children / pcc_compact_* for m candidates. */
int64_t pcc_scan_scratch_elems(int64_t m);

/* Output coordinate set of a stride-2 convolution on a tensor of stride `ts`
 * (ME stride map; model/transforms.py:49-51, model/blocks.py:203,
 * model/entropy_models.py:276,280, model/model.py:189-190):
 * unique floor(c / 2ts) * 2ts, in order of first appearance.  out_coords has room for n
 * rows; *out_count (device int64) receives the number of unique rows, or
 * PCC_COUNT_ERR_RANGE if a source or output coordinate is outside the key range.  On return
 * (keys, vals, cap) is the hash table of the OUTPUT set (cap >= pcc_hash_capacity(n)). */
int pcc_stride_map(const int32_t* coords, int64_t n, int32_t ts, uint64_t* keys, int32_t* vals,
                   int64_t cap, int32_t* scratch, int32_t* out_coords, int64_t* out_count,
                   void* stream);

/* Output coordinate set of a generative transposed convolution, kernel `ksize` (2 or 3),
 * stride 2, on a tensor of stride `ts` (ME.MinkowskiGenerativeConvolutionTranspose:
 * model/blocks.py:84, model/entropy_models.py:286,290; ConvolutionTranspose :298,302):
 * unique c + off_k * ts/2.  Candidates are enumerated parent-major for ksize 3 and
 * offset-major for ksize 2; the output keeps first appearances in that order.
 * out_coords has room for n * ksize^3 rows; table as in pcc_stride_map
 * (cap >= pcc_hash_capacity(n * ksize^3)). */
int pcc_children(const int32_t* coords, int64_t n, int32_t ts, int32_t ksize, uint64_t* keys,
                 int32_t* vals, int64_t cap, int32_t* scratch, int32_t* out_coords,
                 int64_t* out_count, void* stream);

/* Kernel map (ME kernel_map, cached per coordinate manager): for every output row j and
 * kernel offset k, nbr[j*K + k] = input row at c_out + sign * off_k * step, or -1.
 * sign = +1 for (strided) convolution with step = input tensor stride; sign = -1 for
 * transposed / generative convolution with step = input stride / 2.
 * row_mask[j] (uint32) has bit k set iff output row j has a neighbour at offset k.
 * *pair_count (device int64, may be NULL) receives the number of (output row, offset) pairs
 * with a neighbour = the "pairs" of the algorithmic FLOP count 2 * pairs * C_in * C_out. */
int pcc_kernel_map(const int32_t* out_coords, int64_t n_out, const uint64_t* in_keys,
                   const int32_t* in_vals, int64_t in_cap, int32_t ksize, int32_t step,
                   int32_t sign, int32_t* nbr, uint32_t* row_mask, int64_t* pair_count,
                   void* stream);

/* The pair count of a kernel map from its row masks alone (sum of popcounts) — for callers that passed
 * pair_count = NULL to pcc_kernel_map and want the figure later (FLOP accounting of a benchmark: the codec itself
 * never reads it, so the product path does not compute it per map). */
int pcc_pair_count(const uint32_t* row_mask, int64_t n_out, int64_t* pair_count, void* stream);

/* Execution order for the MFMA convolution.  Output rows are sorted by
 * (spatial block of 2^block_log2 voxels per axis, neighbour mask); block_log2 < 0 sorts by
 * mask alone.  Rows with the same neighbour pattern become adjacent, so a 32-row MFMA tile
 * multiplies (almost) no all-zero neighbour rows.  Results do not depend on the order.
 *   order[p]        = output row executed at position p
 *   group_mask32[g] = OR of row_mask over positions 32g .. 32g+31
 * The neighbour table itself stays in output-row order: the convolution kernels read row order[p] of it.
 * scratch_bytes from pcc_order_scratch_bytes(n). */
int64_t pcc_order_scratch_bytes(int64_t n);
int pcc_order_rows_by_mask(const uint32_t* row_mask, const int32_t* coords, int64_t n,
                           int32_t block_log2, int32_t tensor_stride, int32_t* order, uint32_t* group_mask32,
                           void* scratch, int64_t scratch_bytes, void* stream);
/* nbr_sorted[p, :] = nbr[order[p], :] — the permuted copy of a table for the weight-gradient kernels of the training
 * path, which index their table by execution position (pcc_conv_wgrad*). */
int pcc_permute_map_rows(const int32_t* nbr, const int32_t* order, int64_t n, int32_t K, int32_t* nbr_sorted, void* stream);

/* pcc_kernel_map + pcc_order_rows_by_mask (mask order over the whole map: block_log2 < 0) in ONE launch for maps of at
 * most pcc_small_map_max() output rows (256; 0 when PCC_SMALL_MAP=0): a single 1024-thread workgroup probes, counts, builds
 * the keys and sorts — three launches otherwise (six and a memset above 16,384 rows), each a few microseconds of work
 * behind its dispatch; from ~300 rows on the probes are too much work for one CU.
 * Outputs as those two calls write them, bit for bit (nbr [n_out, K], row_mask, order, group_mask32);
 * scratch_bytes from pcc_order_scratch_bytes(n_out). */
int64_t pcc_small_map_max(void);
/* One-workgroup forms of the small per-map chains, as a bit mask: 1 = execution order (maps of at most 16,384 rows: counts, keys and
 * sort in one launch), 2 = top-k (one batch item, at most 32,768 rows), 4 = coordinate sets (at most 8,192 candidates).  All on
 * by default (PCC_ORDER_SMALL=0 / PCC_TOPK_SMALL=0 / PCC_UNIQUE_SMALL=0 switch one off at start-up); sets the mask and returns
 * the previous one, a negative argument only reads it.  Outputs do not depend on it. */
int32_t pcc_small_paths(int32_t mask);
int pcc_small_kernel_map(const int32_t* out_coords, int64_t n_out, const uint64_t* in_keys, const int32_t* in_vals, int64_t in_cap,
                         int32_t ksize, int32_t step, int32_t sign, int32_t* nbr, uint32_t* row_mask, int32_t* order,
                         uint32_t* group_mask32, void* scratch, int64_t scratch_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Sparse convolution forward (ME.MinkowskiConvolution / *ConvolutionTranspose forward,
 * 100+ instances: model/transforms.py:35-57,168-234; model/blocks.py:17-24,86-97,194-220;
 * model/entropy_models.py:273-305) with the pointwise ops that follow it fused:
 *   v = bias + sum_k in[nbr(j,k)] @ W[k]                 (k ascending, fixed order)
 *   if film:     v = v * film[j, c] + film[j, cout + c]   (ScaledBlock, blocks.py:37-40)
 *   v = act(v)                                            (MinkowskiReLU / LeakyReLU)
 *   if residual: v += residual[j, c]                      (blocks.py:49-52)
 * W is the ME kernel tensor [K, cin, cout] (fp32).  For cin % 32 == 0 the MFMA path is used
 * and needs `w_packed` from pcc_conv_pack_weights; otherwise (cin in {1,2,3,4,6,8,12,16,24}) the
 * thin path reads `w` directly.  nbr == NULL means kernel_size 1 (identity map, K = 1).
 * MFMA path: if `order` != NULL, position p computes output row order[p] and `group_mask32` holds the
 * masks from pcc_order_rows_by_mask; with order == NULL rows run in natural order (group_mask32 may
 * then be NULL = every offset executed).
 * nbr / film / residual / fout are always indexed by the output row, never by the position.
 * MFMA path (also pcc_conv_fwd_bf16, pcc_conv_wgrad*): the gathered tensors (fin, dy) and
 * w_packed must be 16-byte aligned — rows move by 16-byte LDS-DMA loads; PCC_ERR_ARG otherwise.
 * ------------------------------------------------------------------------------------- */
int64_t pcc_conv_packed_elems(int32_t K, int32_t cin, int32_t cout);
int pcc_conv_pack_weights(const float* w, int32_t K, int32_t cin, int32_t cout, float* w_packed,
                          void* stream);
int pcc_conv_fwd(const float* fin, int64_t n_in, int32_t cin, const float* w, const float* w_packed,
                 const float* bias, const int32_t* nbr, const int32_t* order,
                 const uint32_t* group_mask32, int32_t K, float* fout, int64_t n_out, int32_t cout,
                 int32_t act, const float* film, const float* residual, void* stream);

/* Small launches of pcc_conv_fwd (map convolutions, cin % 32 == 0, whose 32 x 32 output tiles number at most
 * `workgroups`, twice that for outputs narrower than 128 columns) run on conv_small_kernel: one 16 x 16 MFMA block per wave,
 * loader waves running the LDS-DMAs ahead — the time of such a launch is one workgroup's serial MFMA chain, which this makes
 * four times shorter (128 -> 128 on 1,136 rows: 27 us against 66).  Results are bit-identical.  Sets the threshold (default
 * 640, or PCC_CONV_SMALL_MAX; 0 = never) and returns the previous one; a negative argument only reads it. */
int64_t pcc_conv_small_max(int64_t workgroups);

/* bf16-input variant of pcc_conv_fwd (training / BASELINE config 5): features and packed weights are bf16
 * (fin [n_in, cin] bf16, cin a multiple of 64; pcc_conv_pack_weights_bf16: [K, cin/8, cout^32, 8]), products
 * accumulate in fp32 on v_mfma_f32_32x32x16_bf16, bias / FiLM / activation / residual and the output are fp32.
 * Same tiling, row order, offset skipping and accumulation order as the fp32 kernel; half the gather bytes
 * and 1/8 of the MFMA time per channel.  NOT used by compress / decompress (fp32 parity budget). */
int64_t pcc_conv_packed_elems_bf16(int32_t K, int32_t cin, int32_t cout);
int pcc_conv_pack_weights_bf16(const float* w, int32_t K, int32_t cin, int32_t cout, uint16_t* w_packed, void* stream);
int pcc_conv_fwd_bf16(const uint16_t* fin, int64_t n_in, int32_t cin, const uint16_t* w_packed, const float* bias,
                      const int32_t* nbr, const int32_t* order, const uint32_t* group_mask32, int32_t K, float* fout,
                      int64_t n_out, int32_t cout, int32_t act, const float* film, const float* residual, void* stream);

/* Training-path epilogue of a convolution as its own operator (the inference path fuses it into pcc_conv_fwd):
 *   forward   out = act(film ? c * beta + gamma : c) + (residual ? residual : 0)     (blocks.py:37-40,49-52)
 *   backward  du = dout * act'(u);  dc = film ? du * beta : du;  dfilm = [du * c | du]   (d residual = dout)
 * c / out / dout / dc [n, channels], film / dfilm [n, 2 * channels] (beta | gamma); channels % 4 == 0, 16-byte aligned.
 * Same operation order as the torch ops they replace (mul, add, act, add): identical values and gradients. */
int pcc_epilogue_fwd(const float* c, const float* film, const float* residual, int64_t n, int32_t channels, int32_t act,
                     float* out, void* stream);
int pcc_epilogue_bwd(const float* dout, const float* c, const float* film, int64_t n, int32_t channels, int32_t act,
                     float* dc, float* dfilm, void* stream);

/* dY of a convolution read once on the training path (autograd.py, SparseConvFn.backward): out_bf16 (may be NULL) = the
 * bf16 copy the bf16 weight-gradient / backward-data kernels gather (round to nearest even, torch's conversion), colsum
 * (may be NULL) = the column sums = the bias gradient (ME differentiates `out + bias`, reference blocks.py convolutions
 * with bias=True).  Deterministic two-stage reduction; scratch: pcc_cast_colsum_scratch_elems(channels) floats.
 * channels % 4 == 0, <= 1024; x 16-byte aligned. */
int64_t pcc_cast_colsum_scratch_elems(int32_t channels);
int pcc_cast_colsum(const float* x, int64_t n, int32_t channels, uint16_t* out_bf16, float* colsum, float* scratch,
                    int64_t scratch_elems, void* stream);

/* Split-bf16 arithmetic on fp32 data (opt-in; the default convolution multiplies in fp32): every fp32 operand is
 * the exact sum of three bf16 numbers; the weights are pre-split into three planes (pcc_conv_pack_weights_x3,
 * pcc_conv_packed_elems_x3 bf16 elements), the gathered fp32 rows are split in registers, and the six products
 * whose weight is at least 2^-16 of the leading one run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation in a
 * fixed order.  Same semantics, epilogue and determinism guarantees as pcc_conv_fwd; results differ from it by
 * the size of an fp32 rounding per product (dropped terms < 3 x 2^-24 |x||w|).  cin % 32 == 0, cin <= 256, output
 * width (rounded up to 32) a multiple of 64; encoder and decoder must use the same mode. */
int64_t pcc_conv_packed_elems_x3(int32_t K, int32_t cin, int32_t cout);
int pcc_conv_pack_weights_x3(const float* w, int32_t K, int32_t cin, int32_t cout, uint16_t* w_packed, void* stream);
int pcc_conv_fwd_x3(const float* fin, int64_t n_in, int32_t cin, const uint16_t* w_packed, const float* bias,
                    const int32_t* nbr, const int32_t* order, const uint32_t* group_mask32, int32_t K, float* fout,
                    int64_t n_out, int32_t cout, int32_t act, const float* film, const float* residual, void* stream);

/* Thin inputs, wide outputs (cin <= 8, cout % 32 == 0: the first layers of the q-map heads and of g_a) on the matrix cores:
 * out[j, :] = the K * cin inputs of output row j's neighbourhood (zeros for absent neighbours), zero-padded to k2 (a multiple
 * of 32) columns, stored in the order pcc_conv_fwd's MFMA loop contracts channels — physical column 8 g + s holds logical
 * column 8 g + 2 (s & 3) + (s >> 2), logical = k * cin + ci.  Followed by a kernel_size-1 pcc_conv_fwd over `out` with the
 * weights re-laid-out the same way ([1, k2, cout]: row 8 g + [0,4,1,5,2,6,3,7][t] = W[k, ci, :] of logical 8 g + t), the sum
 * per output element runs over (k, ci) ascending like the scalar thin kernel's: identical bits, the fp32 MFMA being the
 * same fused multiply-add chain as v_fma_f32. */
int pcc_im2col_thin(const float* fin, int32_t cin, const int32_t* nbr, int64_t n_out, int32_t K, float* out, int32_t k2,
                    void* stream);

/* Narrow-head convolution, second half (cout <= 4 on wide inputs: the occupancy logit of
 * model/blocks.py:94-98,142 and the q-map heads).  The caller first computes
 * scores[i, k*cout + c] = in[i] . W[k][:, c] for every INPUT row with one kernel_size-1 call of
 * pcc_conv_fwd on the re-laid-out kernel [cin, K*cout]; this call adds, per output row, the
 * <= K scalars its neighbours contribute: out[j,c] = act(bias[c] + sum_k scores[nbr(j,k), k*cout+c]).
 * `nbr` is the natural-order table of pcc_kernel_map; `ld` = row stride of scores in floats. */
int pcc_gather_sum_fwd(const float* scores, int32_t ld, const int32_t* nbr, int32_t K, int32_t cout,
                       const float* bias, float* out, int64_t n_out, int32_t act, void* stream);

/* ---------------------------------------------------------------------------------------
 * Row movement: lookup-gather, pruning.
 * ------------------------------------------------------------------------------------- */
/* out[i,:] (+)= idx[i] >= 0 ? src[idx[i],:] : 0   (features_at_coordinates after
 * pcc_hash_lookup; accumulate != 0 adds into out: model/transforms.py:96,262). */
int pcc_gather_rows(const float* src, int32_t c, const int32_t* idx, int64_t n, float* out,
                    int32_t accumulate, void* stream);

/* out[idx[i],:] = src[i,:] for idx[i] >= 0 (re-indexing a tensor onto another map). */
int pcc_scatter_rows(const float* src, int32_t c, const int32_t* idx, int64_t n, float* out,
                     void* stream);
/* out[idx[i],:] += src[i,:] for idx[i] >= 0: the backward of pcc_gather_rows (training path; the
 * reference gets it from torch's indexing autograd behind features_at_coordinates, loss.py:103-107). */
int pcc_scatter_add_rows(const float* src, int32_t c, const int32_t* idx, int64_t n, float* out,
                         void* stream);

/* ME.MinkowskiPruning (model/blocks.py:90,126): order-preserving compaction of the rows
 * with mask != 0.  Either of feats/out_feats and coords/out_coords may be NULL.
 * new_index[i] (optional) = output row of input row i or -1.  *out_count: device int64. */
int pcc_compact_rows(const uint8_t* mask, int64_t n, const int32_t* coords, int32_t* out_coords,
                     const float* feats, int32_t c, float* out_feats, int32_t* new_index,
                     int32_t* scratch, int64_t* out_count, void* stream);

/* ---------------------------------------------------------------------------------------
 * Per-batch top-k on one logit per row (GenerativeUpBlock._topk_prediction,
 * model/blocks.py:130-150, torch.topk).  Row i belongs to batch coords[i*4]; batch b keeps
 * its k[b] largest logits (logits[i*ld]); exact ties are broken by ascending voxel key.
 * NaN sorts above +inf (torch.topk).  state: >= pcc_topk_state_elems(nbatch) int32 elements (per item the selection's words and a
 * 256-bin histogram; for one item also the 256 x 256 per-workgroup bins of the logit passes).
 * ------------------------------------------------------------------------------------- */
int64_t pcc_topk_state_elems(int32_t nbatch);
int pcc_topk_mask(const float* logits, int32_t ld, const int32_t* coords, int64_t n, int32_t nbatch,
                  const int32_t* k, uint8_t* mask, int32_t* state, void* stream);

/* rows-per-batch histogram (AnalysisTransform.count_per_batch, model/transforms.py:65-71). */
int pcc_count_per_batch(const int32_t* coords, int64_t n, int32_t nbatch, int32_t* counts, void* stream);

/* Canonical (b,x,y,z)-lexicographic order (utils.sort_tensor / sort_points,
 * utils.py:155-204): perm[r] = input row that comes r-th.  scratch_bytes from the query. */
int64_t pcc_sort_scratch_bytes(int64_t n);
int pcc_sort_coords(const int32_t* coords, int64_t n, int32_t* perm, void* scratch,
                    int64_t scratch_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Convolution backward (training path; the reference gets it from MinkowskiEngine's autograd
 * functions behind every ME.Minkowski*Convolution*, train.py:194-206).
 *   dX = pcc_conv_fwd over the TRANSPOSED map with transposed weights W^T[k] = W[k]^T:
 *        pcc_kernel_map_transpose gives nbr_t[i, k] = output row j with nbr[j, k] == i (or -1) and
 *        the per-input-row offset masks (feed them to pcc_order_rows_by_mask like a forward map).
 *   dW[k] = sum over the pairs of offset k of X[i]^T dY[j]: pcc_conv_wgrad (nbr / order / group masks
 *        of the FORWARD map in execution order; order == NULL: natural order).  fp32 MFMA for
 *        cin, cout multiples of 32, a scalar kernel for thin shapes (cin * cout <= 4096).  The
 *        reduction order is fixed (bitwise reproducible).  scratch: pcc_conv_wgrad_scratch_elems floats.
 *   db = column sums of dY (caller).
 * ------------------------------------------------------------------------------------- */
int pcc_kernel_map_transpose(const int32_t* nbr, int64_t n_out, int32_t K, int64_t n_in, int32_t* nbr_t,
                             uint32_t* row_mask_t, void* stream);
int64_t pcc_conv_wgrad_scratch_elems(int32_t K, int32_t cin, int32_t cout);
int pcc_conv_wgrad(const float* fin, int64_t n_in, int32_t cin, const float* dy, int64_t n_out, int32_t cout,
                   const int32_t* nbr, const int32_t* order, const uint32_t* group_mask32, int32_t K, float* dw,
                   float* scratch, int64_t scratch_elems, void* stream);
/* bf16 operands (fin, dy bf16; cin, cout multiples of 64), fp32 accumulation and result; same scratch size. */
int pcc_conv_wgrad_bf16(const uint16_t* fin, int64_t n_in, int32_t cin, const uint16_t* dy, int64_t n_out, int32_t cout,
                        const int32_t* nbr, const int32_t* order, const uint32_t* group_mask32, int32_t K, float* dw,
                        float* scratch, int64_t scratch_elems, void* stream);

/* ---------------------------------------------------------------------------------------
 * Latent-coordinate side channel of file mode.  Replaces ColorModel.gpcc_encode / gpcc_decode
 * (model/model.py:318-395), which shell out to the external MPEG G-PCC binary `tmc3`; this is the
 * build's own lossless octree ("PCO1", NOT G-PCC compatible; container in octree.py).
 *
 * pcc_octree_occupancy: coords [n,4] (batch ignored), grid = (c - origin) / stride must lie in
 *   [0, 2^depth)^3.  Child index per level = (xbit << 2) | (ybit << 1) | zbit.  Level L
 *   (0 = root .. depth-1) receives one occupancy byte per occupied node, ascending Morton order,
 *   at occupancy[L * n ...]; level_counts (device, depth + 2 ints) = nodes per level, then the
 *   number of distinct leaves (== n unless the input holds duplicates), then the number of
 *   inputs outside the grid / off the stride lattice (must be 0).  origin: 3 host ints.
 * pcc_octree_expand: the inverse.  occupancy = the levels back to back (device), level_counts
 *   = depth host int64; writes coords_out [n_points,4] = (batch, x, y, z) in ascending Morton
 *   order.  The caller has already checked that the popcounts of level L sum to the size of
 *   level L+1 (n_points for the last); a stream that lies cannot write out of bounds.
 * ------------------------------------------------------------------------------------- */
int64_t pcc_octree_scratch_bytes(int64_t n);
int pcc_octree_occupancy(const int32_t* coords, int64_t n, int32_t stride, const int32_t* origin,
                         int32_t depth, uint8_t* occupancy, int32_t* level_counts, void* scratch,
                         int64_t scratch_bytes, void* stream);
int pcc_octree_expand(const uint8_t* occupancy, const int64_t* level_counts, int32_t depth,
                      int32_t stride, const int32_t* origin, int32_t batch, int64_t n_points,
                      int32_t* coords_out, void* scratch, int64_t scratch_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Quality metrics: nearest-neighbour association on integer grids.  Replaces the open3d KD-tree
 * queries of PointCloudMetric (metrics/metric.py:36-43).  For every query row (b,x,y,z) finds the
 * nearest target voxel through the target's hash table (pcc_hash_build): nn_idx = target row,
 * nn_d2 = squared Euclidean distance (sum over axes, NOT the reference's per-axis mean);
 * equidistant candidates resolve to the smallest (x,y,z); tie_count / tie_rgb (optional) = number
 * of equidistant nearest voxels and the sum of their colours (target_rgb [n,3], f64 like the
 * reference's numpy arrays) for the reference's
 * tie-averaging mode (metric.py:121-146).  Queries whose neighbour is farther than max_radius
 * (Chebyshev shells searched: 0..max_radius) get nn_idx = nn_d2 = -1 and must be retried wider.
 * ------------------------------------------------------------------------------------- */
int pcc_nn_search(const int32_t* query, int64_t nq, const uint64_t* keys, const int32_t* vals, int64_t cap,
                  int32_t tensor_stride, const double* target_rgb, int32_t max_radius, int32_t* nn_idx, int64_t* nn_d2,
                  int32_t* tie_count, double* tie_rgb, void* stream);

/* ---------------------------------------------------------------------------------------
 * Entropy model, device side (compressai EntropyBottleneck / GaussianConditional,
 * model/entropy_models.py:313,330,352-353,371-372,393,407-408).  Features are [N, C]
 * row-major; symbol / index / likelihood planes are channel-major [C, N] — the order in
 * which the reference flattens (1, C, N) tensors into one rANS stream.
 * ------------------------------------------------------------------------------------- */
/* symbols[c,n] = rint(z[n,c] - median[c]);  z_hat[n,c] = symbols + median  (either may be NULL). */
int pcc_eb_quantize(const float* z, int64_t n, int32_t c, const float* medians, int32_t* symbols,
                    float* z_hat, void* stream);
/* z_hat[n,c] = symbols[c,n] + median[c] */
int pcc_eb_dequantize(const int32_t* symbols, int64_t n, int32_t c, const float* medians,
                      float* z_hat, void* stream);
/* Factorized-density likelihood max(|sigmoid(s*u) - sigmoid(s*l)|, 1e-9) of v = z_hat
 * (B.2).  eb_params: per channel 58 floats = softplus(matrix_i), bias_i, tanh(factor_i)
 * flattened in the order m0[3] b0[3] f0[3] m1[9] b1[3] f1[3] m2[9] b2[3] f2[3] m3[9] b3[3]
 * f3[3] m4[3] b4[1].  lik is [C, N]. */
int pcc_eb_likelihood(const float* z_hat, int64_t n, int32_t c, const float* eb_params, float* lik,
                      void* stream);
/* params [N, 2C] = (scales | means) rows aligned with y (h_s output looked up at y's
 * coordinates).  indexes[c,n] = 63-level table index of max(scale, 0.11);
 * symbols[c,n] = rint(y - mean).  scale_table: `levels` ascending floats. */
int pcc_gc_encode_prep(const float* y, const float* params, int64_t n, int32_t c,
                       const float* scale_table, int32_t levels, int32_t* symbols,
                       int32_t* indexes, void* stream);
/* The same preparation with the planes the host coder reads directly, already in stream order: output column j
 * takes row perm[j] (perm may be NULL = identity; the reference sorts its tensors canonically before coding,
 * utils.py:155-180, model/entropy_models.py:357-372), symbols as int16 [c,n] (may be NULL: indexes only, the
 * decoder's case), indexes as uint8 [c,n] (levels <= 256).  *overflow (device int32, required with symbols) is
 * set to 1 when a symbol does not fit int16 — the caller then uses pcc_gc_encode_prep. */
int pcc_gc_encode_prep_packed(const float* y, const float* params, int64_t n, int32_t c,
                              const float* scale_table, int32_t levels, const int32_t* perm,
                              int16_t* symbols, uint8_t* indexes, int32_t* overflow, void* stream);
int pcc_gc_dequantize_i16(const int16_t* symbols, const float* params, int64_t n, int32_t c,
                          float* y_hat, void* stream);
/* y_hat[n,c] = symbols[c,n] + mean[n,c]  (GaussianConditional.decompress/dequantize). */
int pcc_gc_dequantize(const int32_t* symbols, const float* params, int64_t n, int32_t c, float* y_hat,
                      void* stream);
/* eval-mode forward: y_hat = rint(y - mean) + mean, lik[c,n] = Gaussian bin mass, floor 1e-9. */
int pcc_gc_forward(const float* y, const float* params, int64_t n, int32_t c, float* y_hat, float* lik,
                   void* stream);

/* ---------------------------------------------------------------------------------------
 * Entropy coder, host side (compressai _CXX / ans: encode_with_indexes,
 * decode_with_indexes, pmf_to_quantized_cdf — same argument order).  All pointers HOST.
 * cdfs is [n_cdfs, cdf_stride] int32.  One rANS stream per call.
 * ------------------------------------------------------------------------------------- */
/* Returns bytes written (>= 8), PCC_ERR_ARG if out_cap is too small (needs <= 4*(3n+4)). */
int64_t pcc_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, int64_t n,
                                     const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                                     const int32_t* offsets, uint8_t* out, int64_t out_cap);
int pcc_rans_decode_with_indexes(const uint8_t* data, int64_t nbytes, const int32_t* indexes, int64_t n,
                                 const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                                 const int32_t* offsets, int32_t* out_symbols);
/* The same coder on the packed planes of pcc_gc_encode_prep_packed (int16 symbols, uint8 indexes): identical
 * bytes.  The decoder sets *narrowed (host int32, required) to 1 when a decoded symbol does not fit int16 —
 * out_symbols is then unusable and the caller repeats the decode with pcc_rans_decode_with_indexes. */
int64_t pcc_rans_encode_with_indexes_i16u8(const int16_t* symbols, const uint8_t* indexes, int64_t n,
                                           const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                                           const int32_t* offsets, uint8_t* out, int64_t out_cap);
int pcc_rans_decode_with_indexes_u8i16(const uint8_t* data, int64_t nbytes, const uint8_t* indexes, int64_t n,
                                       const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                                       const int32_t* offsets, int16_t* out_symbols, int32_t* narrowed);
/* cdf has n+1 entries. */
int pcc_pmf_to_quantized_cdf(const float* pmf, int32_t n, int32_t precision, int32_t* cdf);

#ifdef __cplusplus
}
#endif
#endif /* PCC_HIP_H */
