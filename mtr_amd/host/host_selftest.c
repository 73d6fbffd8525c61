/* host_selftest.c — drives the host-side logic (FASTA reader, chaining, printers) WITHOUT the GPU: records come
 * from a text file (one per line: read_index rep_start rep_end repeat_len period copies mat mis ins del k G MM D unit)
 * instead of libmtr_hip.so.  Used by tests/test_host_driver.py to check chaining + printing against the
 * reference's golden stdout on machines without a GPU. */
#include "mtr_host.h"
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: host_selftest fasta records.txt [-a]\n"); return 2; }
    const int print_alignment = argc > 3 && strcmp(argv[3], "-a") == 0;
    FILE *rf = fopen(argv[2], "r");
    if (!rf) { perror(argv[2]); return 2; }
    mtr_record *recs = NULL; int *owner = NULL; int n = 0, cap = 0;
    char unit[2048];
    for (;;) {
        mtr_record r; int idx;
        memset(&r, 0, sizeof(r));
        int got = fscanf(rf, "%d %d %d %d %d %d %d %d %d %d %d %d %d %d %2047s", &idx, &r.rep_start, &r.rep_end, &r.repeat_len, &r.rep_period,
                         &r.num_freq_unit, &r.num_matches, &r.num_mismatches, &r.num_insertions, &r.num_deletions, &r.kmer,
                         &r.match_gain, &r.mismatch_penalty, &r.indel_penalty, unit);
        if (got != 15) break;
        strncpy(r.unit, unit, sizeof(r.unit) - 1);
        if (n == cap) { cap = cap ? cap * 2 : 64; recs = (mtr_record *)realloc(recs, sizeof(mtr_record) * (size_t)cap); owner = (int *)realloc(owner, sizeof(int) * (size_t)cap); }
        recs[n] = r; owner[n] = idx; n++;
    }
    fclose(rf);
    mtrh_fasta *fa = mtrh_fasta_open(argv[1]);
    mtrh_read rd; int ridx = 0, p = 0;
    while (mtrh_fasta_next_batch(fa, &rd, 1, 1LL << 40) == 1) {
        int q = p; while (q < n && owner[q] == ridx) q++;
        if (q > p) {
            int *chain = (int *)malloc(sizeof(int) * (size_t)(q - p));
            int nc = mtrh_chain(recs + p, q - p, chain);
            mtrh_print_chain(stdout, &rd, recs + p, chain, nc, print_alignment);
            free(chain);
        }
        p = q; ridx++;
        mtrh_read_free(&rd);
    }
    mtrh_fasta_close(fa);
    free(recs); free(owner);
    return 0;
}
