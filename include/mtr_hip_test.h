/*
 * mtr_hip_test.h — entry points of libmtr_hip.so that exist for parity tests and debugging only (the same kernels the
 * batch path of include/mtr_hip.h runs, reachable stage by stage).  Not part of the drop-in boundary: a maintainer of
 * the reference binds include/mtr_hip.h alone.
 */
#ifndef MTR_HIP_TEST_H
#define MTR_HIP_TEST_H

#include "mtr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* K1 alone = fill_directional_index_with_end (fill_directional_index.c:549-602) for every read of the
 * uploaded batch.  Returns per read the surviving candidate ranges (start ascending): start, end, w
 * and the DI value's IEEE-754 bit pattern.  Arrays are malloc'ed; free() them. */
mtr_status mtr_test_ranges(mtr_ctx *ctx, int32_t **out_counts, int32_t **out_start, int32_t **out_end,
                           int32_t **out_w, uint64_t **out_di_bits, int64_t *out_total);

/* wrap_around_DP_sub (wrap_around_DP.c:222-354) for n_tasks (read, window, unit, scores) tasks on the
 * uploaded batch.  unit codes 0..3, units concatenated, unit_off[n_tasks+1].  out8[8*t..] =
 * rep_start, rep_end, repeat_len, Num_freq_unit, matches, mismatches, insertions, deletions. */
mtr_status mtr_test_wrap_dp(mtr_ctx *ctx, int32_t n_tasks, const int32_t *read_idx, const int32_t *query_start,
                            const int32_t *query_end, const uint8_t *units, const int32_t *unit_off,
                            const int32_t *gain, const int32_t *mismatch, const int32_t *indel, int32_t *out8);

/* How the last launch ran: 0 = one kernel, a wavefront per read; 1 = range-parallel; 2 = the staged chain of kernels.
 * (The records do not depend on it; tests pin the policy of mtr_hip.h's mtr_set_overlapped_launches with it.) */
int32_t mtr_test_last_mode(const mtr_ctx *ctx);

/* Event trace of the last run (debug aid for parity work): enable before mtr_run_resident.
 * Each event is 16 int32: [0]=type (2 search, 3 DP, 4 polish, 5 revise, 6 record, 7 per-read cost, 8 per-walk cost), [1]=read index,
 * then type-specific fields (the trace_ev() calls in mtr_amd/csrc/k2_units.hip.inc). */
mtr_status mtr_set_trace(mtr_ctx *ctx, int32_t max_events);
mtr_status mtr_get_trace(mtr_ctx *ctx, int32_t **out_events, int64_t *out_n);

#ifdef __cplusplus
}
#endif
#endif /* MTR_HIP_TEST_H */
