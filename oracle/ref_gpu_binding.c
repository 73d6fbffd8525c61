/*
 * ref_gpu_binding.c — TEST INFRASTRUCTURE ONLY: the binding of INTEGRATION.md, compiled.
 *
 * This is the handle_one_file() a maintainer of reference mTR would write to call libmtr_hip.so (the same text as
 * in INTEGRATION.md).  oracle/Makefile links it with the UNMODIFIED reference objects compiled from the sources where
 * they lie (main.c, chaining.cpp, wrap_around_DP.c's printer, handle_one_file.c's reader with its own handle_one_file
 * renamed on the command line) into oracle/_ref/mTR_ref_gpu: the reference's front end, reader, chaining and
 * printers around this repository's per-read path.  tests/test_gpu_cli.py runs it on the GPU box and compares its
 * stdout with the reference's own.  Nothing of the reference is copied: its header and objects are used in place.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "mTR.h"                           /* the reference's header, -I$(REF) */
#include "mtr_hip.h"                       /* this repository: include/mtr_hip.h; link with -lmtr_hip */

void malloc_global_variables(void);
void free_global_variables(void);
FILE *init_handle_one_file(char *inputFile);
void return_one_read(FILE *fp, Read *currentRead);

int handle_one_file(char *inputFile, int print_alignment)
{
    malloc_global_variables();             /* still needed by chaining's -a printer (orgInputString, WrapDP) */
    mtr_ctx *gpu;
    if (mtr_create(0, Manhattan_Distance, min_match_ratio, &gpu) != MTR_OK) {
        fprintf(stderr, "no HIP device\n"); exit(EXIT_FAILURE);
    }
    Read *currentRead = malloc(sizeof(Read));
    FILE *fp = init_handle_one_file(inputFile);
    enum { BATCH = 65536 };
    uint8_t *bases = malloc((size_t)BATCH * 4096); size_t cap = (size_t)BATCH * 4096, used = 0;
    int64_t *off = malloc(sizeof(int64_t) * BATCH); int32_t *len = malloc(sizeof(int32_t) * BATCH);
    char (*ids)[MAX_ID_LENGTH] = malloc((size_t)BATCH * MAX_ID_LENGTH);
    int n = 0, last = 0;
    while (!last) {
        return_one_read(fp, currentRead);                       /* handle_one_file.c:201-269, unchanged */
        last = (currentRead->len == 0);
        if (!last) {
            if (used + currentRead->len > cap) { cap = 2 * (used + currentRead->len); bases = realloc(bases, cap); }
            off[n] = used; len[n] = currentRead->len; strcpy(ids[n], currentRead->ID);
            for (int i = 0; i < currentRead->len; i++) bases[used + i] = (uint8_t)currentRead->codedString[i];
            used += currentRead->len; n++;
        }
        if (n == BATCH || (last && n > 0)) {
            mtr_record *rec; int32_t *cnt; int64_t total;
            if (mtr_process_batch(gpu, bases, off, len, n, &rec, &cnt, &total) != MTR_OK) {
                fprintf(stderr, "%s\n", mtr_last_error(gpu)); exit(EXIT_FAILURE);
            }
            int64_t p = 0;
            for (int r = 0; r < n; r++) {                       /* = the tail of handle_one_TR, handle_one_read.c:239-252 */
                for (int i = 0; i < len[r]; i++) orgInputString[i] = bases[off[r] + i];   /* for pretty_print_alignment */
                for (int k = 0; k < cnt[r]; k++, p++)
                    insert_an_alignment_into_set(ids[r], len[r], rec[p].rep_start, rec[p].rep_end, rec[p].repeat_len,
                        rec[p].rep_period, rec[p].num_freq_unit, rec[p].num_matches, rec[p].num_mismatches,
                        rec[p].num_insertions, rec[p].num_deletions, rec[p].kmer, rec[p].match_gain,
                        rec[p].mismatch_penalty, rec[p].indel_penalty, rec[p].unit, rec[p].unit_score);
                chaining(print_alignment);
            }
            mtr_free_results(rec, cnt);
            n = 0; used = 0;
        }
    }
    fclose(fp); mtr_destroy(gpu); free_global_variables(); free(currentRead);
    return tmp_read_cnt;
}
