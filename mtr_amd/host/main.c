/* main.c — mTR [-acp] [-m ratio] <fasta file name>: the reference's command line (main.c:48-123) over the
 * MI355X hot path.  Reads are taken in batches, handed to libmtr_hip.so (mtr_process_batch), and every read's
 * records are chained and printed in input order. */
#define _POSIX_C_SOURCE 200809L
#include "mtr_host.h"
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/time.h>

#define BATCH_READS 65536
#define BATCH_BASES (512LL << 20)

static void usage(void)
{
    fprintf(stderr, "mTR [-acp] [-m ratio] <fasta file name> \n");
    fprintf(stderr, "-a: Output the alignment between the input sequence and predicted tandem repeat. \n");
    fprintf(stderr, "-c: Print the computation time of each step.\n");
    fprintf(stderr, "-m ratio: Give a minimum match ratio ranging from 0 to 1.\n");
    fprintf(stderr, "-p: Use Pearson's correlation coefficient distance in place of Manhattan distance.\n");
}

static double now(void) { struct timeval t; gettimeofday(&t, NULL); return t.tv_sec + t.tv_usec * 1.0E-6; }

int main(int argc, char **argv)
{
    int print_time = 0, print_alignment = 0, manhattan = 1, device = 0;
    float min_match_ratio = 0.6f;                 /* MIN_MATCH_RATIO, mTR.h:32 */
    int opt;
    while ((opt = getopt(argc, argv, "acm:pd:")) != -1) {
        switch (opt) {
        case 'a': print_alignment = 1; break;
        case 'c': print_time = 1; break;
        case 'm':
            min_match_ratio = (float)atof(optarg);
            if (!(0 <= min_match_ratio && min_match_ratio <= 1)) { fprintf(stderr, "The input minimum match ratio must range from 0 to 1.\n"); exit(EXIT_FAILURE); }
            break;
        case 'p': manhattan = 0; fprintf(stderr, "Pearson's correlation coefficient distance in place of Manhattan distance.\n"); break;
        case 'd': device = atoi(optarg); break;    /* extension: GPU ordinal */
        default: usage(); exit(EXIT_FAILURE);
        }
    }
    if (optind >= argc) { fprintf(stderr, "The input file name is expected argument after options\n"); exit(EXIT_FAILURE); }

    const double t_all = now();
    mtr_ctx *ctx = NULL;
    mtr_status st = mtr_create(device, manhattan, min_match_ratio, &ctx);
    if (st != MTR_OK) { fprintf(stderr, "fatal error: no usable HIP device (mtr_create returned %d); this build has no CPU path\n", (int)st); exit(EXIT_FAILURE); }

    mtrh_fasta *fa = mtrh_fasta_open(argv[optind]);
    mtrh_read *reads = (mtrh_read *)calloc(BATCH_READS, sizeof(mtrh_read));
    double t_k1 = 0, t_k2 = 0, t_chain = 0; long long queries = 0;
    int n;
    while ((n = mtrh_fasta_next_batch(fa, reads, BATCH_READS, BATCH_BASES)) > 0) {
        int64_t total = 0;
        for (int i = 0; i < n; i++) total += reads[i].len;
        uint8_t *bases = (uint8_t *)malloc((size_t)total);
        int64_t *offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
        int32_t *lens = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
        if (!bases || !offs || !lens) { fprintf(stderr, "cannot allocate the batch\n"); exit(EXIT_FAILURE); }
        int64_t o = 0;
        for (int i = 0; i < n; i++) { offs[i] = o; lens[i] = reads[i].len; memcpy(bases + o, reads[i].codes, (size_t)reads[i].len); o += reads[i].len; }
        mtr_record *recs = NULL; int32_t *counts = NULL; int64_t nrec = 0;
        st = mtr_process_batch(ctx, bases, offs, lens, n, &recs, &counts, &nrec);
        if (st != MTR_OK) { fprintf(stderr, "%s\n", mtr_last_error(ctx)); exit(EXIT_FAILURE); }
        mtr_kernel_time kt[2]; mtr_get_kernel_times(ctx, kt, 2); t_k1 += kt[0].ms * 1e-3; t_k2 += kt[1].ms * 1e-3;
        int64_t cnt[MTR_N_COUNTERS]; mtr_get_counters(ctx, cnt, MTR_N_COUNTERS); queries += cnt[8];
        const double tc = now();
        int64_t p = 0;
        for (int i = 0; i < n; i++) {
            if (counts[i] > 0) {
                int *chain = (int *)malloc(sizeof(int) * (size_t)counts[i]);
                int nc = mtrh_chain(recs + p, counts[i], chain);
                mtrh_print_chain(stdout, &reads[i], recs + p, chain, nc, print_alignment);
                free(chain);
            }
            p += counts[i];
            mtrh_read_free(&reads[i]);
        }
        t_chain += now() - tc;
        mtr_free_results(recs, counts);
        free(bases); free(offs); free(lens);
    }
    mtrh_fasta_close(fa);
    free(reads);
    mtr_destroy(ctx);
    if (print_time) {                             /* the reference's -c block (main.c:108-121) */
        fprintf(stderr, "Computation time\n");
        fprintf(stderr, "%f\tall\n", now() - t_all);
        fprintf(stderr, "%f\tallocating memory\n", 0.0);
        fprintf(stderr, "%f\tranges\n", t_k1);
        fprintf(stderr, "%f\tComputing periods\n", t_k2 + t_chain);
        fprintf(stderr, "\t%f\tInitialize the input\n", 0.0);
        fprintf(stderr, "\t%f\tcount table generation\n", 0.0);
        fprintf(stderr, "\t%f\twrap around\n", 0.0);
        fprintf(stderr, "\t%f\tchaining\n", t_chain);
        fprintf(stderr, "\t%i\tCount of queries\n", (int)queries);
    }
    return EXIT_SUCCESS;
}
