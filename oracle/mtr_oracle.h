/*
 * mtr_oracle — TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the per-read hot path of reference mTR
 * (handle_one_read.c:190-266 and everything below it) under the "isolated semantics" contract of
 * each read is processed as if it were the only read in its process.
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check /
 * time the HIP path on machines where /root/reference does not exist.  Nothing under mtr_amd/
 * may include, link or call it.
 *
 * Parity status: PINNED — checked line by line against the capture points G1/G3/G3p/G3r/G4/G5
 * recorded from the unmodified reference by oracle/ref_capture.c (tests/test_oracle_golden.py,
 * tests/golden/), and against the md5 fingerprints of SURVEY.md Appendix C.
 */
#ifndef MTR_ORACLE_H
#define MTR_ORACLE_H
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTRO_MAX_PERIOD 500          /* mTR.h:35 */
#define MTRO_MAX_INPUT_LENGTH 1000000 /* mTR.h:31 */

/* One candidate repeat = the 17 arguments of insert_an_alignment_into_set (mTR.h:151-168),
 * minus readID/inputLen which are per read. */
typedef struct {
    int32_t rep_start, rep_end, repeat_len, rep_period, num_freq_unit;
    int32_t num_matches, num_mismatches, num_insertions, num_deletions;
    int32_t kmer, match_gain, mismatch_penalty, indel_penalty;
    char    unit[MTRO_MAX_PERIOD * 2 + 4]; /* NUL-terminated; room for a unit the revision doubled */
    int32_t unit_score[MTRO_MAX_PERIOD];
} mtro_record;

/* per-read work counters (used for DESIGN.md's C_alg / B_alg and by bench.py) */
typedef struct {
    int64_t dp_calls, dp_cells, dp_rows, dp_max_cells;
    int64_t revise_dp_calls, revise_dp_cells;
    int64_t kmer_tables, kmer_lookups, searches_passing_maxfreq;
    int64_t ranges_candidate, ranges_executed, records;
    int64_t di_passes, di_positions;
} mtro_stats;

typedef struct mtro_ctx mtro_ctx;

mtro_ctx *mtro_create(int manhattan, float min_match_ratio);
void      mtro_destroy(mtro_ctx *);
/* capture stream in the JSONL format of oracle/ref_capture.c (NULL = off); level as there */
void      mtro_set_capture(mtro_ctx *, FILE *cap, int level);
/* on = the reference's behaviour on a multi-read file instead of isolated semantics: the reads must then be given
 * in file order (results of a read depend on the longer reads before it).  Checked against the reference run on
 * whole files (tests/test_oracle_golden.py, capture points G1..G4; stdout up to the heap-order chaining ties). */
void      mtro_set_file_order(mtro_ctx *, int on);
const mtro_stats *mtro_get_stats(const mtro_ctx *);
void      mtro_reset_stats(mtro_ctx *);

/* codes[0..L): 0..3 = A C G T.  Returns the number of records (insertion order) written to
 * *out (malloc'ed, caller frees with free()), or <0 on error. */
int mtro_process_read(mtro_ctx *, const char *read_id, const uint8_t *codes, int L, mtro_record **out);

/* chaining.cpp:243-363 with insertion-order ties: writes the indices of the records of the best
 * chain, in print order, to chain_idx (capacity n); returns the chain length. */
int mtro_chain(const mtro_record *recs, int n, int *chain_idx);

/* chaining.cpp:125-171 (+ wrap_around_DP.c:57-213 when print_alignment): prints one chain. */
void mtro_print_chain(FILE *fp, const char *read_id, int L, const uint8_t *codes, const mtro_record *recs,
                      const int *chain_idx, int n_chain, int print_alignment);

/* the same for the read the context processed last, printed from the context's own read buffer (file-order mode: a
 * repeat can end on org[L], a base an earlier read left there) */
void mtro_print_chain_of_last_read(mtro_ctx *, FILE *fp, const char *read_id, const mtro_record *recs,
                                   const int *chain_idx, int n_chain, int print_alignment);

/* stand-alone pieces, exported so that tests can drive the HIP kernels' building blocks */
typedef struct {
    int32_t rep_start, rep_end, repeat_len, num_freq_unit, num_matches, num_mismatches, num_insertions, num_deletions;
} mtro_dp_result;
/* wrap_around_DP_sub (wrap_around_DP.c:222-354) on codes[] with org[p>=L]=0 */
int mtro_wrap_dp(const uint8_t *codes, int L, int query_start, int query_end, const uint8_t *unit, int unit_len,
                 int G, int MM, int D, mtro_dp_result *res);
/* fill_directional_index_with_end (+ de-dup); arrays of length L; returns number of usable ranges */
int mtro_ranges(mtro_ctx *, const uint8_t *codes, int L, double *di, int32_t *end, int32_t *w);
/* the MT19937 stream (MT.h) after init_genrand(0): out[i] = genrand_int32() % 4 */
void mtro_mt_bases(uint8_t *out, int n);

/* FASTA reader with the reference's quirks (handle_one_file.c:201-269).  Returns number of reads;
 * ids[i], seqs[i] (codes 0..3), lens[i] are malloc'ed.  Exits like the reference on a bad character. */
int mtro_read_fasta(const char *path, char ***ids, uint8_t ***seqs, int **lens);

#ifdef __cplusplus
}
#endif
#endif
