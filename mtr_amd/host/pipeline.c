/* pipeline.c — the run of one rank: plan the chunks of the input file(s), parse the chunks this rank owns on parser
 * threads, and push their batches through two device contexts in turn, so that packing + upload + launch of batch b+1
 * and fetch (+ alignments) of batch b overlap the kernels.  Replaces the per-read loop of handle_one_file.c:271-293.
 *
 * Ownership of chunks: one file -> chunk c belongs to rank c % world and round c / world (a round is what the launcher
 * gathers together); several files (test_multiple_TRs/test.sh: one read per file, 2.6-140 kb) -> longest first to the
 * least loaded rank, one round.
 */
#define _GNU_SOURCE
#include "mtr_host.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define BATCH_READS 16384
#define BATCH_BASES ((int64_t)512 << 20)
#define CHUNK_BYTES ((size_t)24 << 20)         /* ~12 000 reads of 2 kb: one device batch */
#define PARSE_AHEAD 3
#define RESULT_QUEUE 4

typedef struct { int file; size_t begin, end; int owner, round; } chunk_t;
/* what the device thread holds while it runs (see device_main) */
#define MAX_CTX 6
typedef struct { mtr_ctx *ctxs[MAX_CTX]; mtr_file_state *fs; mtrh_result *pend[MAX_CTX]; mtr_ctx *pend_ctx[MAX_CTX]; int n_pend; int *file_ended; char *dead_msg; } device_state;

struct mtrh_run {
    mtrh_opts o; mtrh_engine eng;
    int n_files; mtrh_file *files;
    int n_chunks, n_rounds; chunk_t *chunks;
    int n_list; int *list;                     /* the chunks this rank parses, ascending: its own (with -B also those before them) */
    mtrh_batch **parsed; int *pstate;          /* per list entry */
    int next_parse, consumed;
    pthread_mutex_t mu; pthread_cond_t cv_parse, cv_res;
    int n_parsers; pthread_t parsers[16], device;
    mtrh_result *queue[RESULT_QUEUE]; int q_head, q_n, device_done, stopping;
    double t_parse_wait, t_submit, t_fetch, t_kernel; long long queries;
    double t_create, t_phase[MTR_N_KERNEL_TIMES];   /* seconds: creating the device contexts; device time by phase of the chain (mtr_get_kernel_times ids) */
    int overlap;                                   /* more than one device batch: launches of the two contexts overlap */
    int parse_failed;                              /* a parser thread could not allocate: the device thread reports it */
    int *chunk_done;                               /* per chunk: its last result has been handed on */
    device_state dev;                              /* the device thread's state (well defined after a longjmp out of device_body) */
};

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
/* development aid (MTR_HOST_TIMING): a line on stderr with the time since the first stamp */
void mtrh_stamp(const char *what)
{
    static int on = -1; static double t_first;
    if (on < 0) { on = getenv("MTR_HOST_TIMING") ? 1 : 0; t_first = now_s(); }
    if (on) fprintf(stderr, "[host +%.3f s] %s\n", now_s() - t_first, what);
}

/* ---- plan ------------------------------------------------------------------------------------------------------------- */
static int cmp_size_desc(const void *a, const void *b)
{
    const chunk_t *x = *(const chunk_t *const *)a, *y = *(const chunk_t *const *)b;
    const size_t sx = x->end - x->begin, sy = y->end - y->begin;
    if (sx != sy) return sx > sy ? -1 : 1;
    return x < y ? -1 : 1;
}

static void plan(mtrh_run *r)
{
    const int world = r->o.world > 0 ? r->o.world : 1;
    const size_t cb = r->o.chunk_bytes ? r->o.chunk_bytes : CHUNK_BYTES;
    int cap = 0;
    for (int f = 0; f < r->n_files; f++) cap += (int)(r->files[f].size / cb) + world + 2;
    r->chunks = (chunk_t *)calloc((size_t)cap, sizeof(chunk_t));
    for (int f = 0; f < r->n_files; f++) {
        /* a multiple of the number of ranks, so that the last round is a full one */
        size_t want = (r->files[f].size + cb - 1) / cb;
        if (want < 1) want = 1;
        if (r->n_files == 1 && world > 1) want = (want + (size_t)world - 1) / (size_t)world * (size_t)world;
        int nc = 0;
        size_t *off = mtrh_plan_chunks(&r->files[f], (int)want, &nc);
        for (int c = 0; c < nc; c++) { chunk_t *k = &r->chunks[r->n_chunks++]; k->file = f; k->begin = off[c]; k->end = off[c + 1]; }
        free(off);
    }
    if (!r->o.lpt) {
        for (int c = 0; c < r->n_chunks; c++) { r->chunks[c].owner = c % world; r->chunks[c].round = c / world; }
        r->n_rounds = (r->n_chunks + world - 1) / world;
    } else {
        /* longest processing time first: a read's cost grows faster than its length (the DPs of a range are rows x unit) */
        chunk_t **ord = (chunk_t **)malloc(sizeof(chunk_t *) * (size_t)r->n_chunks);
        double *load = (double *)calloc((size_t)world, sizeof(double));
        for (int c = 0; c < r->n_chunks; c++) ord[c] = &r->chunks[c];
        qsort(ord, (size_t)r->n_chunks, sizeof(chunk_t *), cmp_size_desc);
        for (int c = 0; c < r->n_chunks; c++) {
            chunk_t *k = ord[c];
            int best = 0;
            for (int q = 1; q < world; q++) if (load[q] < load[best]) best = q;
            k->owner = best; k->round = 0;
            load[best] += pow((double)(k->end - k->begin), 1.5);
        }
        r->n_rounds = r->n_chunks > 0 ? 1 : 0;
        free(ord); free(load);
    }
}

/* ---- parser threads ----------------------------------------------------------------------------------------------------- */
static void *parser_main(void *arg)
{
    mtrh_run *r = (mtrh_run *)arg;
    jmp_buf oom;
    mtrh_thread_kind = MTRH_THREAD_PARSER;
    if (setjmp(oom)) {                                   /* an allocation of this thread failed (alloc.c): the device thread says so */
        mtrh_oom_target = NULL;
        pthread_mutex_lock(&r->mu);
        r->parse_failed = 1;
        pthread_cond_broadcast(&r->cv_parse);
        pthread_mutex_unlock(&r->mu);
        return NULL;
    }
    mtrh_oom_target = &oom;
    for (;;) {
        pthread_mutex_lock(&r->mu);
        while (!r->stopping && r->next_parse < r->n_list && r->next_parse >= r->consumed + PARSE_AHEAD + r->n_parsers) pthread_cond_wait(&r->cv_parse, &r->mu);
        if (r->stopping || r->next_parse >= r->n_list) { pthread_mutex_unlock(&r->mu); return NULL; }
        const int idx = r->next_parse++;
        pthread_mutex_unlock(&r->mu);
        const chunk_t *c = &r->chunks[r->list[idx]];
        /* the code array (a byte per base) is only read by -a (the rows of an alignment block) and -B (the file state): every other run parses into the image alone */
        mtrh_batch *b = (r->o.print_alignment || r->o.file_order) ? mtrh_parse_chunk(&r->files[c->file], c->begin, c->end, BATCH_READS, BATCH_BASES)
                                                                  : mtrh_parse_chunk_packed(&r->files[c->file], c->begin, c->end, BATCH_READS, BATCH_BASES);
        pthread_mutex_lock(&r->mu);
        r->parsed[idx] = b; r->pstate[idx] = 1;
        pthread_cond_broadcast(&r->cv_parse);
        pthread_mutex_unlock(&r->mu);
    }
}

/* ---- results ---------------------------------------------------------------------------------------------------------- */
void mtrh_result_free(mtrh_result *x)
{
    if (!x) return;
    mtrh_batch_free(x->batch);
    free(x->counts); free(x->wire); free(x->chain_len); free(x->chain_idx); free(x->ops); free(x->ops_off); free(x->ends); free(x->after); free(x->fatal_msg);
    free(x);
}

static void push_result(mtrh_run *r, mtrh_result *x)
{
    pthread_mutex_lock(&r->mu);
    if (x->last_of_chunk && x->chunk >= 0 && x->chunk < r->n_chunks) r->chunk_done[x->chunk] = 1;
    while (r->q_n == RESULT_QUEUE && !r->stopping) pthread_cond_wait(&r->cv_res, &r->mu);
    if (r->stopping) { pthread_mutex_unlock(&r->mu); mtrh_result_free(x); return; }
    r->queue[(r->q_head + r->q_n) % RESULT_QUEUE] = x; r->q_n++;
    pthread_cond_broadcast(&r->cv_res);
    pthread_mutex_unlock(&r->mu);
}

mtrh_result *mtrh_run_next(mtrh_run *r)
{
    pthread_mutex_lock(&r->mu);
    while (r->q_n == 0 && !r->device_done) pthread_cond_wait(&r->cv_res, &r->mu);
    mtrh_result *x = NULL;
    if (r->q_n > 0) { x = r->queue[r->q_head]; r->q_head = (r->q_head + 1) % RESULT_QUEUE; r->q_n--; pthread_cond_broadcast(&r->cv_res); }
    pthread_mutex_unlock(&r->mu);
    return x;
}

static mtrh_result *result_new(int chunk, int file_idx, mtrh_batch *b)
{
    mtrh_result *x = (mtrh_result *)calloc(1, sizeof *x);
    if (!x) { fprintf(stderr, "cannot allocate a result\n"); exit(EXIT_FAILURE); }
    x->chunk = chunk; x->file_idx = file_idx; x->batch = b; x->n_report = b ? b->n : 0; x->ticket = -1;
    return x;
}

static void result_fail(mtrh_result *x, int n_report, const char *msg)
{
    x->fatal = 1; x->n_report = n_report;
    free(x->fatal_msg); x->fatal_msg = strdup(msg ? msg : "device error");
}

static void *xmalloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "cannot allocate %zu bytes\n", n); exit(EXIT_FAILURE); }
    return p;
}

/* -a: chain every read where the batch is resident, then ONE device call aligns all reported repeats of the batch */
static void add_alignments(mtrh_run *r, mtr_ctx *ctx, mtrh_result *x)
{
    const mtrh_batch *b = x->batch;
    const int n = x->n_report;
    x->with_alignments = 1;
    x->chain_len = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    x->after = (uint8_t *)calloc((size_t)b->n * 2 + 2, 1);
    int64_t total = 0;
    for (int i = 0; i < n; i++) total += x->counts[i];
    x->chain_idx = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)(total + 1));
    mtrh_rec *recs = (mtrh_rec *)xmalloc(sizeof(mtrh_rec) * (size_t)(total + 1));
    int *tmp = (int *)xmalloc(sizeof(int) * (size_t)(total + 1));
    const uint8_t *p = x->wire, *end = x->wire + x->wire_bytes;
    int64_t nk = 0, base = 0;
    int32_t *t_read = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)(total + 1));
    const mtrh_rec **t_rec = (const mtrh_rec **)xmalloc(sizeof(mtrh_rec *) * (size_t)(total + 1));
    for (int i = 0; i < n; i++) {
        const int c = x->counts[i];
        for (int t = 0; t < c; t++) if (!mtrh_rec_next(&p, end, &recs[base + t])) { result_fail(x, 0, "internal error: malformed record table"); goto done; }
        const int nc = c > 0 ? mtrh_chain(recs + base, c, tmp) : 0;
        x->chain_len[i] = nc;
        for (int t = 0; t < nc; t++) { x->chain_idx[nk] = tmp[t]; t_read[nk] = i; t_rec[nk] = &recs[base + tmp[t]]; nk++; }
        base += c;
    }
    x->n_chain = nk;
    for (int i = 0; i < b->n; i++) (void)r->eng.bases_after(ctx, i, x->after + 2 * (size_t)i);      /* zeros unless -B */
    {
        mtr_record *full = (mtr_record *)xmalloc(sizeof(mtr_record) * (size_t)(nk + 1));     /* the ABI takes mtr_record: header + unit are read */
        for (int64_t k = 0; k < nk; k++) {
            memcpy(&full[k], t_rec[k]->h, MTR_WIRE_HEADER_BYTES);
            const int per = full[k].rep_period;
            memcpy(full[k].unit, t_rec[k]->unit, (size_t)per); full[k].unit[per] = 0;
        }
        uint8_t *ops = NULL; int64_t *off = NULL; int32_t *ends = NULL;
        const mtr_status st = r->eng.alignments(ctx, (int32_t)nk, t_read, full, &ops, &off, &ends);
        free(full);
        if (st != MTR_OK) { result_fail(x, 0, r->eng.last_error(ctx)); free(ops); free(off); free(ends); goto done; }
        x->ops = ops; x->ops_off = off; x->ends = ends;
    }
done:
    free(recs); free(tmp); free(t_read); free((void *)t_rec);
}

/* a batch whose kernels were started: wait, fetch the records in wire form (+ the alignments), hand the result on */
static void finish_batch(mtrh_run *r, mtr_ctx *ctx, mtrh_result *x)
{
    const double t0 = now_s();
    const mtrh_batch *b = x->batch;
    mtr_status st = r->eng.wait(ctx);
    { static int first = 1; if (__atomic_exchange_n(&first, 0, __ATOMIC_RELAXED)) mtrh_stamp("first batch: device done"); }
    int n_report = b->n;
    if (st == MTR_ERR_DP_TOO_LARGE) {
        /* like the reference (wrap_around_DP.c:96-99): everything before the failing read is reported, then the message */
        int32_t ff = -1;
        (void)r->eng.first_failed(ctx, &ff);
        result_fail(x, ff > 0 ? ff : 0, r->eng.last_error(ctx));
        n_report = x->n_report;
    } else if (st != MTR_OK) { result_fail(x, 0, r->eng.last_error(ctx)); n_report = 0; }
    x->counts = (int32_t *)calloc((size_t)n_report + 1, sizeof(int32_t));
    x->ticket = -1;
    if (n_report > 0 && r->o.gather && !x->fatal && !r->o.print_alignment) {
        /* several GPUs: the table stays on this GPU, compacted to the wire form, until the round's exchange takes it to the first GPU (multi.c) */
        int64_t bytes = 0, total = 0; int32_t ticket = -1;
        st = r->eng.gather_stage(r->o.gather, r->o.rank, ctx, x->counts, &total, &bytes, &ticket);
        if (st == MTR_OK) { x->ticket = ticket; x->wire_bytes = bytes; }
        else if (st != MTR_ERR_OVERFLOW) { result_fail(x, 0, r->eng.last_error(ctx)); n_report = 0; }
    }
    /* no gather - or every staging slot of this GPU holds a table that waits for its round's exchange (several files: ONE round, so a GPU with more than
     * the pool's batches staged them all): the table comes to the host with this thread's own fetch, as on one GPU */
    if (n_report > 0 && x->ticket < 0 && !(x->fatal && n_report == 0)) {
        if (r->o.gather && !x->fatal && !r->o.print_alignment) x->fetched_by_run = 1;
        const uint8_t *blob = NULL; const int32_t *counts = NULL; int64_t bytes = 0, total = 0;
        st = r->eng.fetch_packed(ctx, x->fatal ? n_report : -1, &blob, &bytes, &counts, &total);
        if (st != MTR_OK) { result_fail(x, 0, r->eng.last_error(ctx)); n_report = 0; }
        else {
            memcpy(x->counts, counts, sizeof(int32_t) * (size_t)n_report);
            x->wire = (uint8_t *)xmalloc((size_t)bytes + 8); x->wire_bytes = bytes;
            if (bytes) memcpy(x->wire, blob, (size_t)bytes);
        }
    }
    x->n_report = n_report;
    mtr_kernel_time kt[MTR_N_KERNEL_TIMES];
    memset(kt, 0, sizeof kt);
    if (r->eng.kernel_times(ctx, kt, MTR_N_KERNEL_TIMES) == MTR_OK) {
        x->t_kernel_ms = (double)kt[0].ms + (double)kt[1].ms;
        for (int i = 0; i < MTR_N_KERNEL_TIMES; i++) x->t_phase_ms[i] = (double)kt[i].ms;
    }
    int64_t cnt[MTR_N_COUNTERS];
    if (r->eng.counters(ctx, cnt, MTR_N_COUNTERS) == MTR_OK) x->queries = cnt[8];
    if (r->o.print_alignment && n_report > 0) add_alignments(r, ctx, x);
    pthread_mutex_lock(&r->mu);
    r->t_fetch += now_s() - t0; r->t_kernel += x->t_kernel_ms * 1e-3; r->queries += x->queries;
    for (int i = 0; i < MTR_N_KERNEL_TIMES; i++) r->t_phase[i] += x->t_phase_ms[i] * 1e-3;
    pthread_mutex_unlock(&r->mu);
    push_result(r, x);
}

/* after a failed allocation (this thread's or a parser's): every chunk this rank reports and has not finished gets a result that
 * carries the message - what is printed before it stays printed, the run ends with status 1, and no rank waits for a chunk forever */
static void report_failure(mtrh_run *r, const char *msg)
{
    for (int c = 0; c < r->n_chunks; c++) {
        if (r->chunks[c].owner != r->o.rank) continue;
        pthread_mutex_lock(&r->mu);
        const int done = r->chunk_done[c], stopping = r->stopping;
        pthread_mutex_unlock(&r->mu);
        if (done) continue;
        if (stopping) break;
        mtrh_result *x = (mtrh_result *)calloc(1, sizeof *x);
        x->chunk = c; x->file_idx = r->chunks[c].file; x->batch = (mtrh_batch *)calloc(1, sizeof(mtrh_batch)); x->last_of_chunk = 1; x->ticket = -1;
        x->counts = (int32_t *)calloc(1, sizeof(int32_t));
        x->fatal = 1; x->fatal_msg = strdup(msg);
        push_result(r, x);
    }
}

/* what the device thread holds while it runs: kept outside device_body so that a failed allocation (a longjmp out of it) still
 * finds the batch in flight and the contexts - they are waited for and destroyed, not left to the process's end */
static void device_body(mtrh_run *r, device_state *d);
static void *device_main(void *arg)
{
    mtrh_run *r = (mtrh_run *)arg;
    jmp_buf oom;
    /* the state lives in the run (heap memory), not in this frame: an automatic variable changed between setjmp and longjmp is
     * indeterminate afterwards (C11 7.13.2.1), and the clean-up below frees pointers and destroys contexts out of it */
    device_state *d = &r->dev;
    memset(d, 0, sizeof *d);
    mtrh_thread_kind = MTRH_THREAD_DEVICE;
    if (setjmp(oom)) {                                   /* an allocation of this thread failed (alloc.c) */
        mtrh_oom_target = NULL;
        d = &r->dev;
        for (int t = 0; t < d->n_pend; t++) { if (d->pend_ctx[t]) (void)r->eng.wait(d->pend_ctx[t]); mtrh_result_free(d->pend[t]); }
        if (d->fs) r->eng.fs_destroy(d->fs);
        for (int t = 0; t < MAX_CTX; t++) if (d->ctxs[t]) { (void)r->eng.wait(d->ctxs[t]); r->eng.destroy(d->ctxs[t]); }
        free(d->file_ended); free(d->dead_msg);
        memset(d, 0, sizeof *d);
        report_failure(r, "fatal error: cannot allocate memory");
    } else {
        mtrh_oom_target = &oom;
        device_body(r, d);
        mtrh_oom_target = NULL;
    }
    pthread_mutex_lock(&r->mu);
    r->device_done = 1;
    pthread_cond_broadcast(&r->cv_res);
    pthread_mutex_unlock(&r->mu);
    return NULL;
}

/* the batches in flight, oldest first: everything before a failure is reported first, like the reference */
static void finish_oldest(mtrh_run *r, device_state *d)
{
    mtrh_result *x = d->pend[0]; mtr_ctx *c = d->pend_ctx[0];
    for (int t = 1; t < d->n_pend; t++) { d->pend[t - 1] = d->pend[t]; d->pend_ctx[t - 1] = d->pend_ctx[t]; }
    d->n_pend--;
    d->pend[d->n_pend] = NULL; d->pend_ctx[d->n_pend] = NULL;
    finish_batch(r, c, x);
}
static void finish_all(mtrh_run *r, device_state *d) { while (d->n_pend > 0) finish_oldest(r, d); }

#define ctxs (d->ctxs)
#define fs (d->fs)
#define file_ended (d->file_ended)
#define dead_msg (d->dead_msg)
static void device_body(mtrh_run *r, device_state *d)
{
    /* nctx contexts = nctx device batches in flight (a context is created when a batch needs it).  Two keep the chip busy on reads of a few kb:
     * the launches are bound by instruction issue.  Launches of LONG reads (config 3's 42 kb) are bound by their longest work items - a 40 000-row
     * alignment is one wavefront's serial chain - and leave most of the chip idle behind them: more batches in flight fill it ([measured, MI355X,
     * round 5] 100 reads of 42 kb per batch: 78.5 ms a step with two contexts, 53.8 with three; round 6: below). */
    int fs_file = -1;
    int k = 0, dead = 0, nctx = r->o.contexts > 0 ? (r->o.contexts > MAX_CTX ? MAX_CTX : r->o.contexts) : 0;
    file_ended = (int *)calloc((size_t)r->n_files + 1, sizeof(int));
    for (int idx = 0; idx < r->n_list; idx++) {
        double t0 = now_s();
        pthread_mutex_lock(&r->mu);
        while (!r->pstate[idx] && !r->stopping && !r->parse_failed) pthread_cond_wait(&r->cv_parse, &r->mu);
        if (!r->pstate[idx] && r->parse_failed && !r->stopping) {
            pthread_mutex_unlock(&r->mu);
            finish_all(r, d);                                    /* what is on the device is reported first */
            mtrh_oom(0);
        }
        mtrh_batch *b = r->parsed[idx]; r->parsed[idx] = NULL;
        r->consumed = idx + 1;
        pthread_cond_broadcast(&r->cv_parse);
        const int stopping = r->stopping;
        r->t_parse_wait += now_s() - t0;
        pthread_mutex_unlock(&r->mu);
        if (stopping) { mtrh_batch_free(b); break; }
        const int cid = r->list[idx];
        const chunk_t *c = &r->chunks[cid];
        const int mine = c->owner == r->o.rank;
        if (r->o.file_order && fs_file != c->file) {             /* the state is per file */
            if (fs) r->eng.fs_destroy(fs);
            fs = NULL; fs_file = c->file;
            if (r->eng.fs_create(&fs) != MTR_OK) { dead = 1; dead_msg = strdup("fatal error: out of memory"); }
        }
        while (b) {
            mtrh_batch *nx = b->next; b->next = NULL;
            const int last = nx == NULL;
            if (!mine) {                                         /* -B: reads another rank processes still shape the state */
                if (fs && b->n > 0 && !file_ended[c->file]) (void)r->eng.fs_skip(fs, b->codes, b->offs, b->lens, b->n);
                if (b->end != MTRH_END_NONE) file_ended[c->file] = 1;
                mtrh_batch_free(b); b = nx;
                continue;
            }
            mtrh_result *x = result_new(cid, c->file, b);
            x->last_of_chunk = last;
            if (dead) { x->n_report = 0; x->counts = (int32_t *)calloc(1, sizeof(int32_t)); result_fail(x, 0, dead_msg); }
            if (dead || file_ended[c->file] || b->n == 0) {
                /* nothing to run: an empty chunk (its status still travels), or input the reference never reads */
                if (file_ended[c->file] && !dead) { x->n_report = 0; b->end = MTRH_END_NONE; }
                if (!x->counts) x->counts = (int32_t *)calloc((size_t)x->n_report + 1, sizeof(int32_t));
                if (b->n == 0) x->n_report = 0;
                finish_all(r, d);
                if (b->end != MTRH_END_NONE) file_ended[c->file] = 1;
                push_result(r, x);
                b = nx;
                continue;
            }
            t0 = now_s();
            if (nctx == 0) {                                     /* decided by the first batch that runs */
                int64_t bases = 0;
                for (int i = 0; i < b->n; i++) bases += b->lens[i];
                /* round 6 [measured, MI355X, 100 reads of 42 kb per batch]: 46.6 / 38.9 / 36.4 / 32.8 / 31.6 / 34.6 ms a step with 3 / 4 / 5 / 6 / 8 / 10 batches in flight
                 * (a lone launch is 102 ms whatever the number: tails): six for long reads - contexts come into being as batches need them, so a job of two
                 * batches makes two, and one that cannot be created leaves the job with the ones it has */
                nctx = bases / (b->n > 0 ? b->n : 1) >= 8000 ? 6 : 2;
                if (nctx == 6) {
                    /* six contexts of long reads are ~100 GB of scratch; a process that takes that much right after another gave it back waits seconds for the
                     * driver ([measured] the same 1 200-read job 0.8 s, and 3.9 s as the second of two in a row): only where the job has the batches to earn it */
                    int my_chunks = 0;
                    for (int q = 0; q < r->n_chunks; q++) if (r->chunks[q].owner == r->o.rank) my_chunks++;
                    if (my_chunks < 8) nctx = 3;
                }
                /* a long job of short reads: a third batch in flight is worth 1.4 % ([measured] 10 000 reads of 2 kb per batch: 36.64 / 36.13 / 36.02 ms a step
                 * with two / three / four) and costs a third context's memory (16 GB) and creation (~40 ms): from ~32 batches on */
                if (nctx == 2 && bases > 0) {
                    int64_t my_bytes = 0;                    /* FASTA bytes of this rank's chunks: about its bases (headers and line ends are a per cent) */
                    for (int q = 0; q < r->n_chunks; q++) if (r->chunks[q].owner == r->o.rank) my_bytes += (int64_t)(r->chunks[q].end - r->chunks[q].begin);
                    if (my_bytes / bases >= 32) nctx = 3;
                }
            }
            mtr_ctx **pc = &ctxs[k % nctx];
            mtr_status st = MTR_OK;
            if (!*pc) {
                const double tc = now_s();
                st = r->eng.create(r->o.device, r->o.manhattan, r->o.min_match_ratio, pc);
                pthread_mutex_lock(&r->mu); r->t_create += now_s() - tc; pthread_mutex_unlock(&r->mu);
                mtrh_stamp(k == 0 ? "first device context created" : k == 1 ? "second device context created" : "third device context created");
                if (st != MTR_OK && k % nctx >= 1) {
                    /* a second or third context that cannot be had (a card shared with other ranks, a smaller one): the job goes on with the contexts it has -
                     * every batch in flight is finished first, so that the context the next batch takes is idle */
                    *pc = NULL;
                    finish_all(r, d);
                    nctx = k % nctx; k = 0;
                    pc = &ctxs[0]; st = MTR_OK;
                    mtrh_stamp("a further device context could not be created: going on with fewer batches in flight");
                }
                if (st != MTR_OK) {
                    char m[256];
                    snprintf(m, sizeof m, "fatal error: no usable HIP device (mtr_create returned %d); this build has no CPU path", (int)st);
                    dead = 1; dead_msg = strdup(m);
                }
            }
            if (!dead) {
                st = fs ? r->eng.upload_in_file(*pc, fs, b->codes, b->offs, b->lens, b->n)       /* uploads happen in file order */
                        : r->eng.upload_packed(*pc, b->packed, b->n_words, b->woff, b->lens, b->n);
                if (st == MTR_OK) st = r->eng.run_async(*pc);
                if (st == MTR_ERR_OOM && !fs && (k % nctx) >= 1) {
                    /* a further context that exists but cannot get its device buffers (six contexts of long reads take 17 GB of scratch each; a card shared with
                     * other runs): like one that cannot be created - it goes, the batches in flight are finished, and the batch runs on the first context.
                     * (Not with -B: the file state has taken the batch's reads already.) */
                    r->eng.destroy(*pc); *pc = NULL;
                    finish_all(r, d);
                    nctx = k % nctx; k = 0;
                    pc = &ctxs[0];
                    mtrh_stamp("a further device context ran out of memory: going on with fewer batches in flight");
                    st = r->eng.upload_packed(*pc, b->packed, b->n_words, b->woff, b->lens, b->n);
                    if (st == MTR_OK) st = r->eng.run_async(*pc);
                }
                if (st != MTR_OK) { dead = 1; dead_msg = strdup(r->eng.last_error(*pc)); }
            }
            if (!r->o.print_alignment && !r->o.file_order) { free(b->codes); b->codes = NULL; }   /* only -a rows and -B need the byte codes from here on */
            free(b->packed); b->packed = NULL;
            pthread_mutex_lock(&r->mu); r->t_submit += now_s() - t0; pthread_mutex_unlock(&r->mu);
            if (k < 2) mtrh_stamp(k == 0 ? "first batch uploaded and launched" : "second batch uploaded and launched");
            if (dead) {
                finish_all(r, d);                                /* like the reference: everything before a failure is reported first */
                x->counts = (int32_t *)calloc(1, sizeof(int32_t));
                result_fail(x, 0, dead_msg);
                push_result(r, x);
            } else {
                if (b->end != MTRH_END_NONE) file_ended[c->file] = 1;
                d->pend[d->n_pend] = x; d->pend_ctx[d->n_pend] = *pc; d->n_pend++; k++;
                while (d->n_pend > nctx - 1) finish_oldest(r, d);       /* the batch just launched stays in flight (with three contexts: the one before it too) */
            }
            b = nx;
        }
    }
    finish_all(r, d);
    mtrh_stamp("last batch fetched");
    if (fs) r->eng.fs_destroy(fs);
    for (int t = 0; t < MAX_CTX; t++) if (ctxs[t]) r->eng.destroy(ctxs[t]);
    mtrh_stamp("device contexts destroyed");
    free(file_ended); free(dead_msg);
    memset(d, 0, sizeof *d);
}
#undef ctxs
#undef fs
#undef file_ended
#undef dead_msg

/* ---- start / stop ------------------------------------------------------------------------------------------------------- */
mtrh_run *mtrh_run_start(const mtrh_opts *o, const char *const *paths, int n_paths)
{
    mtrh_run *r = (mtrh_run *)calloc(1, sizeof *r);
    if (!r) return NULL;
    r->o = *o;
    if (r->o.world < 1) { r->o.world = 1; r->o.rank = 0; }
    char err[512];
    mtrh_stamp("start");
    if (mtrh_engine_load(&r->eng, o->engine_lib, err, sizeof err) != 0) { fprintf(stderr, "fatal error: %s\n", err); free(r); return NULL; }
    r->n_files = n_paths; r->files = (mtrh_file *)calloc((size_t)n_paths, sizeof(mtrh_file));
    for (int f = 0; f < n_paths; f++)
        if (mtrh_file_open(&r->files[f], paths[f]) != 0) { for (int g = 0; g < f; g++) mtrh_file_close(&r->files[g]); free(r->files); free(r); return NULL; }
    mtrh_stamp("engine library loaded, files mapped");
    plan(r);
    /* what this rank parses: its own chunks; with -B every chunk up to its last one (the state needs the reads before) */
    r->list = (int *)malloc(sizeof(int) * ((size_t)r->n_chunks + 1));
    int last_owned = -1;
    for (int c = 0; c < r->n_chunks; c++) if (r->chunks[c].owner == r->o.rank) last_owned = c;
    for (int c = 0; c <= last_owned; c++) if (r->chunks[c].owner == r->o.rank || r->o.file_order) r->list[r->n_list++] = c;
    r->parsed = (mtrh_batch **)calloc((size_t)r->n_list + 1, sizeof(mtrh_batch *));
    r->pstate = (int *)calloc((size_t)r->n_list + 1, sizeof(int));
    r->chunk_done = (int *)calloc((size_t)r->n_chunks + 1, sizeof(int));
    pthread_mutex_init(&r->mu, NULL); pthread_cond_init(&r->cv_parse, NULL); pthread_cond_init(&r->cv_res, NULL);
    int np = o->parse_threads;
    if (np <= 0) { long nc = sysconf(_SC_NPROCESSORS_ONLN); np = nc >= 8 ? 4 : (nc >= 4 ? 2 : 1); }
    if (np > 16) np = 16;
    if (np > r->n_list) np = r->n_list > 0 ? r->n_list : 1;
    r->n_parsers = np;
    for (int t = 0; t < np; t++) pthread_create(&r->parsers[t], NULL, parser_main, r);
    pthread_create(&r->device, NULL, device_main, r);
    return r;
}

int mtrh_run_rank(const mtrh_run *r) { return r->o.rank; }
int mtrh_run_n_chunks(const mtrh_run *r) { return r->n_chunks; }
int mtrh_run_n_rounds(const mtrh_run *r) { return r->n_rounds; }
int mtrh_run_owner(const mtrh_run *r, int chunk) { return chunk >= 0 && chunk < r->n_chunks ? r->chunks[chunk].owner : -1; }
int mtrh_run_round_of(const mtrh_run *r, int chunk) { return chunk >= 0 && chunk < r->n_chunks ? r->chunks[chunk].round : -1; }

const char *mtrh_run_engine_path(const mtrh_run *r) { return r->eng.path; }
void mtrh_run_phase_times(const mtrh_run *r, double *t_create, double *t_phase, int n_phase)
{
    if (t_create) *t_create = r->t_create;
    for (int i = 0; t_phase && i < n_phase; i++) t_phase[i] = i < MTR_N_KERNEL_TIMES ? r->t_phase[i] : 0.0;
}
void mtrh_run_timing(const mtrh_run *r, double *t_parse_wait, double *t_submit, double *t_fetch, double *t_kernel, long long *queries)
{
    if (t_parse_wait) *t_parse_wait = r->t_parse_wait;
    if (t_submit) *t_submit = r->t_submit;
    if (t_fetch) *t_fetch = r->t_fetch;
    if (t_kernel) *t_kernel = r->t_kernel;
    if (queries) *queries = r->queries;
}

void mtrh_run_stop(mtrh_run *r)
{
    if (!r) return;
    pthread_mutex_lock(&r->mu);
    r->stopping = 1;
    pthread_cond_broadcast(&r->cv_parse); pthread_cond_broadcast(&r->cv_res);
    pthread_mutex_unlock(&r->mu);
    pthread_join(r->device, NULL);
    for (int t = 0; t < r->n_parsers; t++) pthread_join(r->parsers[t], NULL);
    for (int i = 0; i < r->q_n; i++) mtrh_result_free(r->queue[(r->q_head + i) % RESULT_QUEUE]);
    for (int i = 0; i < r->n_list; i++) mtrh_batch_free(r->parsed[i]);
    for (int f = 0; f < r->n_files; f++) mtrh_file_close(&r->files[f]);
    mtrh_engine_unload(&r->eng);
    pthread_mutex_destroy(&r->mu); pthread_cond_destroy(&r->cv_parse); pthread_cond_destroy(&r->cv_res);
    free(r->files); free(r->chunks); free(r->list); free(r->parsed); free(r->pstate); free(r->chunk_done);
    free(r);
}
