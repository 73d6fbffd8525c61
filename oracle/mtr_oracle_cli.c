/*
 * mtr_oracle_cli — TEST INFRASTRUCTURE ONLY.  Same command line and stdout as reference mTR
 * (main.c:48-123), driven by the CPU restatement in mtr_oracle.c, one read at a time under isolated
 * semantics (-B: the reference's own behaviour on a multi-read file, mtro_set_file_order).  Extra flags: -l <level> -C <capture.jsonl> write the capture points in the format of
 * oracle/ref_capture.c; -S prints the work counters (DP cells, k-mer look-ups ...) to stderr.
 */
#define _POSIX_C_SOURCE 200809L
#include "mtr_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/time.h>

int main(int argc, char **argv)
{
    int print_alignment = 0, print_time = 0, manhattan = 1, level = 1, print_stats = 0, file_order = 0;
    float min_ratio = 0.6f;
    const char *cap_path = NULL;
    int opt;
    while ((opt = getopt(argc, argv, "acm:pl:C:SB")) != -1) {
        switch (opt) {
        case 'a': print_alignment = 1; break;
        case 'c': print_time = 1; break;
        case 'm': min_ratio = (float)atof(optarg);
            if (!(0 <= min_ratio && min_ratio <= 1)) { fprintf(stderr, "The input minimum match ratio must range from 0 to 1.\n"); return EXIT_FAILURE; }
            break;
        case 'p': manhattan = 0; fprintf(stderr, "Pearson's correlation coefficient distance in place of Manhattan distance.\n"); break;
        case 'l': level = atoi(optarg); break;
        case 'C': cap_path = optarg; break;
        case 'S': print_stats = 1; break;
        case 'B': file_order = 1; break;
        default:
            fprintf(stderr, "mTR [-acp] [-m ratio] <fasta file name> \n");
            return EXIT_FAILURE;
        }
    }
    if (optind >= argc) { fprintf(stderr, "The input file name is expected argument after options\n"); return EXIT_FAILURE; }
    struct timeval s, e; gettimeofday(&s, NULL);
    char **ids; uint8_t **seqs; int *lens;
    int n = mtro_read_fasta(argv[optind], &ids, &seqs, &lens);
    mtro_ctx *c = mtro_create(manhattan, min_ratio);
    if (file_order) mtro_set_file_order(c, 1);
    FILE *cap = NULL;
    if (cap_path) { cap = fopen(cap_path, "w"); if (!cap) { perror(cap_path); return EXIT_FAILURE; } mtro_set_capture(c, cap, level); }
    for (int i = 0; i < n; i++) {
        mtro_record *recs = NULL;
        int nr = mtro_process_read(c, ids[i], seqs[i], lens[i], &recs);
        if (nr > 0) {
            int *chain = (int *)malloc(sizeof(int) * (size_t)nr);
            int nc = mtro_chain(recs, nr, chain);
            if (file_order) mtro_print_chain_of_last_read(c, stdout, ids[i], recs, chain, nc, print_alignment);
            else mtro_print_chain(stdout, ids[i], lens[i], seqs[i], recs, chain, nc, print_alignment);
            fflush(stdout);
            free(chain);
        }
        free(recs);
    }
    gettimeofday(&e, NULL);
    if (print_time) fprintf(stderr, "Computation time\n%f\tall\n", (e.tv_sec - s.tv_sec) + (e.tv_usec - s.tv_usec) * 1.0E-6);
    if (print_stats) {
        const mtro_stats *st = mtro_get_stats(c);
        fprintf(stderr, "{\"reads\":%d,\"dp_calls\":%lld,\"dp_cells\":%lld,\"dp_rows\":%lld,\"dp_max_cells\":%lld,\"revise_dp_calls\":%lld,"
                        "\"revise_dp_cells\":%lld,\"kmer_tables\":%lld,\"kmer_lookups\":%lld,\"searches_passing_maxfreq\":%lld,"
                        "\"ranges_candidate\":%lld,\"ranges_executed\":%lld,\"records\":%lld,\"di_passes\":%lld,\"di_positions\":%lld}\n",
                n, (long long)st->dp_calls, (long long)st->dp_cells, (long long)st->dp_rows, (long long)st->dp_max_cells,
                (long long)st->revise_dp_calls, (long long)st->revise_dp_cells, (long long)st->kmer_tables, (long long)st->kmer_lookups,
                (long long)st->searches_passing_maxfreq, (long long)st->ranges_candidate, (long long)st->ranges_executed,
                (long long)st->records, (long long)st->di_passes, (long long)st->di_positions);
    }
    if (cap) fclose(cap);
    mtro_destroy(c);
    return EXIT_SUCCESS;
}
