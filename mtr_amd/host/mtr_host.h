/* mtr_host.h — host driver of the MI355X mTR: same command line, FASTA input and report/alignment output as
 * reference mTR (main.c, handle_one_file.c, chaining.cpp, pretty_print_alignment), with the per-read hot path
 * delegated to libmtr_hip.so through the C-ABI of include/mtr_hip.h.  Plain C, like the reference's host side. */
#ifndef MTR_HOST_H
#define MTR_HOST_H
#include <stdint.h>
#include <stdio.h>
#include "mtr_hip.h"

#define MTRH_BLK 4096                 /* fgets chunk of the reference reader (mTR.h:57) */
#define MTRH_OVERLAP 10               /* MAX_LEN_overlapping (mTR.h:39) */
#define MTRH_ALIGN_WIDTH 50           /* ALIGNMENT_WIDTH_PRINTING (mTR.h:38) */

typedef struct {
    char    *id;                      /* header line after '>' */
    uint8_t *codes;                   /* 0..3 */
    int32_t  len;
    uint8_t  after[2];                /* orgInputString[len], [len+1]: 0 = 'A' (isolated semantics); with -B the bases an
                                       * earlier, longer read left there (mtr_get_bases_after_read) */
} mtrh_read;

/* Streaming FASTA reader with the reference's rules (handle_one_file.c:169-269): 4096-byte fgets chunks,
 * a line starting with '>' opens a record and the rest of it is the ID, ACGT/acgt only (anything else is
 * fatal: "Invalid character"), processing stops at the first empty record. */
typedef struct mtrh_fasta mtrh_fasta;
mtrh_fasta *mtrh_fasta_open(const char *path);
/* reads up to max_reads records / max_bases bases; returns the number read (0 at the end) */
int  mtrh_fasta_next_batch(mtrh_fasta *f, mtrh_read *out, int max_reads, int64_t max_bases);
void mtrh_fasta_close(mtrh_fasta *f);
void mtrh_read_free(mtrh_read *r);

/* chaining.cpp:243-363 with alignments taken in insertion order: fills chain[] (capacity n) with the indices
 * of the records of the maximum-score chain in print order; returns its length */
int  mtrh_chain(const mtr_record *recs, int n, int *chain);

/* chaining.cpp:125-171: one report line per chained repeat (+ the alignment block with -a) */
void mtrh_print_chain(FILE *fp, const mtrh_read *rd, const mtr_record *recs, const int *chain, int n_chain, int print_alignment);
/* the same with the alignment blocks taken from mtr_alignments() (device) instead of a DP on the host */
void mtrh_print_chain_ops(FILE *fp, const mtrh_read *rd, const mtr_record *recs, const int *chain, int n_chain,
                          const uint8_t *ops, const int64_t *off, const int32_t *ends, int64_t first_task);
#endif
