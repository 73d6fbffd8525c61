/* main.c — mTR [-acp] [-m ratio] <fasta file name>: the reference's command line (main.c:48-123) over the
 * MI355X hot path.  Reads are taken in batches, handed to libmtr_hip.so (mtr_process_batch), and every read's
 * records are chained and printed in input order. */
#define _POSIX_C_SOURCE 200809L
#include "mtr_host.h"
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/time.h>

#define BATCH_READS 16384            /* reads per device batch: two batches overlap on the GPU (two contexts) */
#define N_SLOTS 3                    /* ingest ring: one batch being parsed, one on the device, one being printed */
#define BATCH_BASES (512LL << 20)

static void usage(void)
{
    fprintf(stderr, "mTR [-acp] [-m ratio] <fasta file name> \n");
    fprintf(stderr, "-a: Output the alignment between the input sequence and predicted tandem repeat. \n");
    fprintf(stderr, "-c: Print the computation time of each step.\n");
    fprintf(stderr, "-m ratio: Give a minimum match ratio ranging from 0 to 1.\n");
    fprintf(stderr, "-p: Use Pearson's correlation coefficient distance in place of Manhattan distance.\n");
    fprintf(stderr, "-d n: (this build) use GPU n.  -B: (this build) results of a read depend on the longer reads before it in the file, as in reference mTR.\n");
}

/* ---- double-buffered ingest: a reader thread fills batches, main() consumes them in order ------------------------- */
typedef struct { mtrh_read *reads; int n; uint8_t *bases; int64_t *offs; int32_t *lens; int state; /* 0 free, 1 full */ } batch_t;
typedef struct {
    mtrh_fasta *fa; batch_t slot[N_SLOTS]; int head, tail;      /* producer fills slot[head], consumer takes slot[tail] */
    pthread_mutex_t mu; pthread_cond_t cv; pthread_t th;
} ingest_t;

static void *ingest_main(void *arg)
{
    ingest_t *g = (ingest_t *)arg;
    for (;;) {
        batch_t *b = &g->slot[g->head];
        pthread_mutex_lock(&g->mu);
        while (b->state != 0) pthread_cond_wait(&g->cv, &g->mu);
        pthread_mutex_unlock(&g->mu);
        const int n = mtrh_fasta_next_batch(g->fa, b->reads, BATCH_READS, BATCH_BASES);
        b->n = n; b->bases = NULL; b->offs = NULL; b->lens = NULL;
        if (n > 0) {
            int64_t total = 0;
            for (int i = 0; i < n; i++) total += b->reads[i].len;
            b->bases = (uint8_t *)malloc((size_t)total);
            b->offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
            b->lens = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
            if (!b->bases || !b->offs || !b->lens) { fprintf(stderr, "cannot allocate the batch\n"); exit(EXIT_FAILURE); }
            int64_t o = 0;
            for (int i = 0; i < n; i++) { b->offs[i] = o; b->lens[i] = b->reads[i].len; memcpy(b->bases + o, b->reads[i].codes, (size_t)b->reads[i].len); o += b->reads[i].len; }
        }
        pthread_mutex_lock(&g->mu);
        b->state = 1;
        pthread_cond_broadcast(&g->cv);
        pthread_mutex_unlock(&g->mu);
        g->head = (g->head + 1) % N_SLOTS;
        if (n == 0) return NULL;
    }
}
static void ingest_start(ingest_t *g, mtrh_fasta *fa)
{
    memset(g, 0, sizeof(*g));
    g->fa = fa;
    for (int k = 0; k < N_SLOTS; k++) g->slot[k].reads = (mtrh_read *)calloc(BATCH_READS, sizeof(mtrh_read));
    pthread_mutex_init(&g->mu, NULL); pthread_cond_init(&g->cv, NULL);
    if (pthread_create(&g->th, NULL, ingest_main, g) != 0) { fprintf(stderr, "cannot start the FASTA thread\n"); exit(EXIT_FAILURE); }
}
static batch_t *ingest_take(ingest_t *g)
{
    batch_t *b = &g->slot[g->tail];
    pthread_mutex_lock(&g->mu);
    while (b->state != 1) pthread_cond_wait(&g->cv, &g->mu);
    pthread_mutex_unlock(&g->mu);
    g->tail = (g->tail + 1) % N_SLOTS;
    return b;
}
static void ingest_release(ingest_t *g, batch_t *b)
{
    free(b->bases); free(b->offs); free(b->lens); b->bases = NULL; b->offs = NULL; b->lens = NULL;
    pthread_mutex_lock(&g->mu);
    b->state = 0;
    pthread_cond_broadcast(&g->cv);
    pthread_mutex_unlock(&g->mu);
}
static void ingest_stop(ingest_t *g)
{
    pthread_join(g->th, NULL);
    for (int k = 0; k < N_SLOTS; k++) free(g->slot[k].reads);
    pthread_mutex_destroy(&g->mu); pthread_cond_destroy(&g->cv);
}

typedef struct { int print_alignment; double t_wait, t_submit, t_fetch, t_kernel, t_chain; long long queries; } run_t;
static double now(void);

/* a batch whose kernels were started: wait, fetch the records, chain and print every read in input order */
static void finish_batch(run_t *run, mtr_ctx *ctx, batch_t *b)
{
    const int n = b->n;
    mtrh_read *reads = b->reads;
    mtr_record *recs = NULL; int32_t *counts = NULL; int64_t nrec = 0;
    double t_mark = now();
    mtr_status st = mtr_fetch_results(ctx, &recs, &counts, &nrec);
    if (st != MTR_OK) { fprintf(stderr, "%s\n", mtr_last_error(ctx)); exit(EXIT_FAILURE); }
    run->t_fetch += now() - t_mark;
    mtr_kernel_time kt[2]; mtr_get_kernel_times(ctx, kt, 2); run->t_kernel += (kt[0].ms + kt[1].ms) * 1e-3;
    int64_t cnt[MTR_N_COUNTERS]; mtr_get_counters(ctx, cnt, MTR_N_COUNTERS); run->queries += cnt[8];
    const double tc = now();
    if (!run->print_alignment) {
        int64_t p = 0;
        for (int i = 0; i < n; i++) {
            if (counts[i] > 0) {
                int *chain = (int *)malloc(sizeof(int) * (size_t)counts[i]);
                int nc = mtrh_chain(recs + p, counts[i], chain);
                mtrh_print_chain(stdout, &reads[i], recs + p, chain, nc, 0);
                free(chain);
            }
            p += counts[i];
            mtrh_read_free(&reads[i]);
        }
    } else {
        /* -a: chain every read first, then ONE device call aligns all reported repeats of the batch (the batch is
         * still resident in this context), then print in input order */
        int *chains = (int *)malloc(sizeof(int) * (size_t)(nrec > 0 ? nrec : 1));
        int *nchain = (int *)calloc((size_t)n, sizeof(int));
        int64_t *cfirst = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
        int64_t p = 0, ntask = 0;
        for (int i = 0; i < n; i++) {
            cfirst[i] = ntask;
            if (counts[i] > 0) { nchain[i] = mtrh_chain(recs + p, counts[i], chains + p); ntask += nchain[i]; }
            p += counts[i];
        }
        for (int i = 0; i < n; i++) (void)mtr_get_bases_after_read(ctx, i, reads[i].after);     /* zeros unless -B */
        int32_t *t_read = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ntask > 0 ? ntask : 1));
        mtr_record *t_rec = (mtr_record *)malloc(sizeof(mtr_record) * (size_t)(ntask > 0 ? ntask : 1));
        if (!chains || !nchain || !cfirst || !t_read || !t_rec) { fprintf(stderr, "cannot allocate the alignment tasks\n"); exit(EXIT_FAILURE); }
        p = 0;
        for (int i = 0; i < n; i++) {
            for (int t = 0; t < nchain[i]; t++) { t_read[cfirst[i] + t] = i; t_rec[cfirst[i] + t] = recs[p + chains[p + t]]; }
            p += counts[i];
        }
        uint8_t *ops = NULL; int64_t *ooff = NULL; int32_t *ends = NULL;
        st = mtr_alignments(ctx, (int32_t)ntask, t_read, t_rec, &ops, &ooff, &ends);
        if (st != MTR_OK) { fprintf(stderr, "%s\n", mtr_last_error(ctx)); exit(EXIT_FAILURE); }
        p = 0;
        for (int i = 0; i < n; i++) {
            if (nchain[i] > 0) mtrh_print_chain_ops(stdout, &reads[i], recs + p, chains + p, nchain[i], ops, ooff, ends, cfirst[i]);
            p += counts[i];
            mtrh_read_free(&reads[i]);
        }
        free(ops); free(ooff); free(ends); free(chains); free(nchain); free(cfirst); free(t_read); free(t_rec);
    }
    run->t_chain += now() - tc;
    mtr_free_results(recs, counts);
}

static double now(void) { struct timeval t; gettimeofday(&t, NULL); return t.tv_sec + t.tv_usec * 1.0E-6; }

int main(int argc, char **argv)
{
    int print_time = 0, print_alignment = 0, manhattan = 1, device = 0, file_order = 0;
    float min_match_ratio = 0.6f;                 /* MIN_MATCH_RATIO, mTR.h:32 */
    int opt;
    while ((opt = getopt(argc, argv, "acm:pd:B")) != -1) {
        switch (opt) {
        case 'a': print_alignment = 1; break;
        case 'c': print_time = 1; break;
        case 'm':
            min_match_ratio = (float)atof(optarg);
            if (!(0 <= min_match_ratio && min_match_ratio <= 1)) { fprintf(stderr, "The input minimum match ratio must range from 0 to 1.\n"); exit(EXIT_FAILURE); }
            break;
        case 'p': manhattan = 0; fprintf(stderr, "Pearson's correlation coefficient distance in place of Manhattan distance.\n"); break;
        case 'd': device = atoi(optarg); break;    /* extension: GPU ordinal */
        case 'B': file_order = 1; break;           /* extension: the reference's whole-file behaviour (mtr_hip.h, file-order mode) */
        default: usage(); exit(EXIT_FAILURE);
        }
    }
    if (optind >= argc) { fprintf(stderr, "The input file name is expected argument after options\n"); exit(EXIT_FAILURE); }

    const double t_all = now();
    /* Two contexts = two device batches in flight: batch b+1 is packed, uploaded and started while batch b is still
     * running; batch b is then fetched, chained and printed while b+1 runs.  Output order = input order. */
    mtr_ctx *ctxs[2] = { NULL, NULL };
    mtr_status st = MTR_OK;
    for (int k = 0; k < 2 && st == MTR_OK; k++) st = mtr_create(device, manhattan, min_match_ratio, &ctxs[k]);
    if (st != MTR_OK) { fprintf(stderr, "fatal error: no usable HIP device (mtr_create returned %d); this build has no CPU path\n", (int)st); exit(EXIT_FAILURE); }

    mtr_file_state *fstate = NULL;                /* -B: what the reads of the file leave behind for the reads after them */
    if (file_order && mtr_file_state_create(&fstate) != MTR_OK) { fprintf(stderr, "fatal error: out of memory\n"); exit(EXIT_FAILURE); }
    mtrh_fasta *fa = mtrh_fasta_open(argv[optind]);
    ingest_t ing;
    ingest_start(&ing, fa);
    const int host_timing = getenv("MTR_HOST_TIMING") != NULL;       /* development aid: phase times on stderr */
    if (host_timing) fprintf(stderr, "[host] create %.3f s\n", now() - t_all);
    run_t run; memset(&run, 0, sizeof run);
    run.print_alignment = print_alignment;
    batch_t *prev = NULL; mtr_ctx *prev_ctx = NULL;
    for (int k = 0;; k++) {
        double t_mark = now();
        batch_t *b = ingest_take(&ing);
        run.t_wait += now() - t_mark;
        if (b->n == 0) { ingest_release(&ing, b); break; }
        mtr_ctx *ctx = ctxs[k & 1];
        t_mark = now();
        st = fstate ? mtr_upload_batch_in_file(ctx, fstate, b->bases, b->offs, b->lens, b->n)     /* uploads happen in file order */
                    : mtr_upload_batch(ctx, b->bases, b->offs, b->lens, b->n);
        if (st == MTR_OK) st = mtr_run_resident_async(ctx);
        if (st != MTR_OK) {                       /* like the reference: everything before the failing batch is reported first */
            if (prev) finish_batch(&run, prev_ctx, prev);
            fflush(stdout); fprintf(stderr, "%s\n", mtr_last_error(ctx)); exit(EXIT_FAILURE);
        }
        run.t_submit += now() - t_mark;
        if (prev) { finish_batch(&run, prev_ctx, prev); ingest_release(&ing, prev); }
        prev = b; prev_ctx = ctx;
    }
    if (prev) { finish_batch(&run, prev_ctx, prev); ingest_release(&ing, prev); }
    ingest_stop(&ing);
    const double t_k1 = 0, t_k2 = run.t_kernel, t_chain = run.t_chain; const long long queries = run.queries;
    if (host_timing) fprintf(stderr, "[host] waiting for the FASTA thread %.3f s, pack+upload+launch %.3f s, waiting for the device + fetch %.3f s (kernels %.3f s), chain+print %.3f s\n",
                             run.t_wait, run.t_submit, run.t_fetch, run.t_kernel, run.t_chain);
    mtrh_fasta_close(fa);
    mtr_destroy(ctxs[0]); mtr_destroy(ctxs[1]);
    mtr_file_state_destroy(fstate);
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
