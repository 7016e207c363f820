// Device-wide primitives shared by the coordinate kernels: stable LSD radix sort of (key, int32 value) pairs
// and prefix scans of int32 flags.  Hand-written for gfx950 (64-wide wavefronts: ranks inside a wave come from
// ballots, per-wave digit counters live in LDS); no library dependency.
#pragma once
#include "common.h"

namespace pcc {

static inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

// --- scans (coords.hip) -----------------------------------------------------------------------------
// pos[i] = sum of flags[0..i) (exclusive) or flags[0..i] (inclusive != 0); flags and pos may alias.
// block_sums: scan_block_sums_elems(m) ints of scratch.  *total (device int64, may be NULL) = sum of all flags — or
// COUNT_ERR_RANGE when `err` (device int32, optional) is non-zero by the time the block sums are scanned.
int64_t scan_block_sums_elems(int64_t m);
int scan_flags(const int32_t* flags, int64_t m, int32_t* pos, int32_t* block_sums, int64_t* total, int inclusive,
               hipStream_t st, const int32_t* err = nullptr);

// --- radix sort (sort.hip) --------------------------------------------------------------------------
// Sorts n pairs by bits [begin_bit, end_bit) of the key, ascending, stable.  keys_a / vals_a hold the input and
// are used as ping-pong space together with keys_b / vals_b; the sorted pairs end up in the *_a arrays when the
// number of passes is even and in the *_b arrays when it is odd — radix_sort_result_in_b() tells which.
// vals_a == NULL on entry means "values = 0 .. n-1" (they are then materialised by the first pass).
// counters: radix_sort_counter_bytes(n) bytes of scratch.
// digit_bits: 8, or 9 (32-bit keys only: the 27-bit execution-order keys sort in three passes instead of four).
int radix_sort_passes(int begin_bit, int end_bit, int digit_bits = 8);
static inline bool radix_sort_result_in_b(int begin_bit, int end_bit, int digit_bits = 8) { return radix_sort_passes(begin_bit, end_bit, digit_bits) & 1; }
int64_t radix_sort_counter_bytes(int64_t n);
int radix_sort_order_digit_bits(int64_t n);      // digit width the 27-bit execution-order sort of n keys should use
int radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* keys_b, int32_t* vals_a, int32_t* vals_b, bool vals_are_iota, int64_t n,
                         int begin_bit, int end_bit, void* counters, hipStream_t st, int digit_bits = 8);
int radix_sort_pairs_u64(uint64_t* keys_a, uint64_t* keys_b, int32_t* vals_a, int32_t* vals_b, bool vals_are_iota, int64_t n,
                         int begin_bit, int end_bit, void* counters, hipStream_t st);

}  // namespace pcc
