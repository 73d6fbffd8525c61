/* print.c — report lines (chaining.cpp:127-143) and, with -a, the alignment block (chaining.cpp:164-167 + the printing
 * half of pretty_print_alignment, wrap_around_DP.c:187-212), formatted into memory so that a pool of threads can format
 * the reads of a batch side by side while one writer emits the pieces in input order.
 *
 * The alignment itself is not computed here: the device returns the path of every reported repeat (mtr_alignments:
 * one byte per column, last column first), and this file turns it into the three text rows.
 */
#define _GNU_SOURCE
#include "mtr_host.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

enum { T_MATCH = 1, T_MISMATCH = 2, T_DEL = 3, T_INS = 4 };
static const char BASE[4] = { 'A', 'C', 'G', 'T' };

int mtrh_rec_next(const uint8_t **p, const uint8_t *end, mtrh_rec *out)
{
    const uint8_t *q = *p;
    if (q + MTR_WIRE_HEADER_BYTES > end) return 0;
    out->h = (const int32_t *)q;
    const int per = out->h[MTRH_PERIOD];
    if (per < 0 || per > MTR_MAX_PERIOD) return 0;
    const int64_t need = mtr_wire_record_bytes(per);
    if (q + need > end) return 0;
    out->unit = (const char *)q + MTR_WIRE_HEADER_BYTES;
    out->score = (const int32_t *)(q + MTR_WIRE_HEADER_BYTES + ((per + 3) & ~3));
    *p = q + need;
    return 1;
}

/* ---- a growing text buffer ---------------------------------------------------------------------------------------- */
typedef struct { char *s; size_t n, cap; } tbuf;
static void tb_room(tbuf *b, size_t more)
{
    if (b->n + more <= b->cap) return;
    size_t c = b->cap ? b->cap * 2 : 1 << 16;
    while (c < b->n + more) c *= 2;
    b->s = (char *)realloc(b->s, c);
    if (!b->s) { fprintf(stderr, "cannot allocate the output buffer\n"); exit(EXIT_FAILURE); }
    b->cap = c;
}
static inline void tb_put(tbuf *b, const char *s, size_t n) { tb_room(b, n); memcpy(b->s + b->n, s, n); b->n += n; }
static inline void tb_ch(tbuf *b, char c) { tb_room(b, 1); b->s[b->n++] = c; }
static void tb_int(tbuf *b, int v)
{   /* "%d" */
    char t[16]; int k = 0; unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { t[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    tb_room(b, (size_t)k + 1);
    if (v < 0) b->s[b->n++] = '-';
    while (k) b->s[b->n++] = t[--k];
}

/* chaining.cpp:127-143: ID L start+1 end+1 repeat_len period copies matches ratio mismatches insertions deletions unit */
static void report_line(tbuf *b, const char *id, int id_len, int L, const mtrh_rec *r)
{
    tb_put(b, id, (size_t)id_len); tb_ch(b, '\t');
    tb_int(b, L); tb_ch(b, '\t');
    tb_int(b, r->h[MTRH_REP_START] + 1); tb_ch(b, '\t');
    tb_int(b, r->h[MTRH_REP_END] + 1); tb_ch(b, '\t');
    tb_int(b, r->h[MTRH_REPEAT_LEN]); tb_ch(b, '\t');
    tb_int(b, r->h[MTRH_PERIOD]); tb_ch(b, '\t');
    tb_int(b, r->h[MTRH_COPIES]); tb_ch(b, '\t');
    tb_int(b, r->h[MTRH_MATCHES]); tb_ch(b, '\t');
    tb_room(b, 64);
    b->n += (size_t)snprintf(b->s + b->n, 64, "%f", (float)r->h[MTRH_MATCHES] / r->h[MTRH_REPEAT_LEN]);    /* a float division, printed as a double */
    tb_ch(b, '\t');
    tb_int(b, r->h[MTRH_MISMATCHES]); tb_ch(b, '\t');
    tb_int(b, r->h[MTRH_INSERTIONS]); tb_ch(b, '\t');
    tb_int(b, r->h[MTRH_DELETIONS]); tb_ch(b, '\t');
    int per = r->h[MTRH_PERIOD]; if (per < 0) per = 0;
    tb_put(b, r->unit, strnlen(r->unit, (size_t)per)); tb_ch(b, '\n');
}

/* the block after a report line with -a: ops[] = one byte per column, last column first; end_pos = read position of the
 * last aligned base, end_col = 1-origin unit column it is aligned to */
static void alignment_block(tbuf *b, const uint8_t *codes, int L, const uint8_t after[2], const mtrh_rec *r,
                            const uint8_t *ops, int64_t n_ops, int end_pos, int end_col)
{
    const int U = r->h[MTRH_PERIOD];
    tb_ch(b, '\n');
    tb_put(b, "match gain = ", 13); tb_int(b, r->h[MTRH_GAIN]);
    tb_put(b, ", mismatch penalty = ", 21); tb_int(b, r->h[MTRH_MISMATCH_PEN]);
    tb_put(b, ", indel penalty = ", 18); tb_int(b, r->h[MTRH_INDEL_PEN]);
    tb_put(b, "\n\n", 2);
    if (U <= 0 || n_ops <= 0) return;
    char *a_in = (char *)malloc((size_t)n_ops * 3), *a_sym = a_in + n_ops, *a_rep = a_sym + n_ops;
    if (!a_in) { fprintf(stderr, "cannot allocate the alignment rows\n"); exit(EXIT_FAILURE); }
    int p = end_pos, j = end_col;
    for (int64_t q = 0; q < n_ops; q++) {
        const int c = ops[q];
        const int code = (p >= 0 && p < L) ? codes[p] : ((p >= L && p < L + 2) ? after[p - L] : 0);   /* one past the read: 'A' under isolated semantics */
        const char xb = BASE[code & 3], ub = r->unit[j - 1];
        if (c == T_MATCH) { a_in[q] = xb; a_sym[q] = '|'; a_rep[q] = ub; p--; j--; }
        else if (c == T_MISMATCH) { a_in[q] = xb; a_sym[q] = ' '; a_rep[q] = ub; p--; j--; }
        else if (c == T_DEL) { a_in[q] = '-'; a_sym[q] = ' '; a_rep[q] = ub; j--; }
        else { a_in[q] = xb; a_sym[q] = ' '; a_rep[q] = '-'; p--; }
        if (j == 0) j = U;
    }
    tb_room(b, (size_t)n_ops * 3 + (size_t)(n_ops / MTRH_ALIGN_WIDTH + 1) * 4);
    for (long s = (long)n_ops - 1; 0 <= s; s -= MTRH_ALIGN_WIDTH) {
        const long e = (-1 <= s - MTRH_ALIGN_WIDTH) ? s - MTRH_ALIGN_WIDTH : -1;
        for (long q = s; e < q; q--) b->s[b->n++] = a_in[q];
        b->s[b->n++] = '\n';
        for (long q = s; e < q; q--) b->s[b->n++] = a_sym[q];
        b->s[b->n++] = '\n';
        for (long q = s; e < q; q--) b->s[b->n++] = a_rep[q];
        b->s[b->n++] = '\n'; b->s[b->n++] = '\n';
    }
    free(a_in);
}

/* per-read start of its records in the wire blob (n_report + 1 entries), or NULL if the blob is malformed */
static const uint8_t **read_starts(const mtrh_result *r)
{
    const int n = r->n_report;
    const uint8_t **st = (const uint8_t **)malloc(sizeof(uint8_t *) * ((size_t)n + 1));
    const uint8_t *p = r->wire, *end = r->wire + r->wire_bytes;
    for (int i = 0; i < n; i++) {
        st[i] = p;
        for (int t = 0; t < r->counts[i]; t++) { mtrh_rec x; if (!mtrh_rec_next(&p, end, &x)) { free(st); return NULL; } }
    }
    st[n] = p;
    return st;
}

typedef struct { const mtrh_result *r; const uint8_t **starts; const int64_t *chain_first; } fmt_ctx;

static void format_reads(const fmt_ctx *f, int first, int last, tbuf *b)
{
    const mtrh_result *r = f->r; const mtrh_batch *bt = r->batch;
    mtrh_rec stack_recs[64]; int stack_chain[64];
    for (int i = first; i < last; i++) {
        const int c = r->counts[i];
        if (c <= 0) continue;
        mtrh_rec *recs = c <= 64 ? stack_recs : (mtrh_rec *)malloc(sizeof(mtrh_rec) * (size_t)c);
        int *chain = c <= 64 ? stack_chain : (int *)malloc(sizeof(int) * (size_t)c);
        const uint8_t *p = f->starts[i], *end = f->starts[i + 1];
        for (int t = 0; t < c; t++) (void)mtrh_rec_next(&p, end, &recs[t]);
        if (!r->with_alignments) {
            const int nc = mtrh_chain(recs, c, chain);
            for (int t = 0; t < nc; t++) report_line(b, bt->ids[i], bt->id_lens[i], bt->lens[i], &recs[chain[t]]);
        } else {
            /* the chain was made where the batch was resident (its records were aligned there) */
            const int64_t k0 = f->chain_first[i];
            for (int t = 0; t < r->chain_len[i]; t++) {
                const int64_t k = k0 + t;
                const mtrh_rec *x = &recs[r->chain_idx[k]];
                report_line(b, bt->ids[i], bt->id_lens[i], bt->lens[i], x);
                alignment_block(b, bt->codes + bt->offs[i], bt->lens[i], r->after + 2 * (size_t)i, x,
                                r->ops + r->ops_off[k], r->ops_off[k + 1] - r->ops_off[k], r->ends[2 * k], r->ends[2 * k + 1]);
            }
        }
        if (c > 64) { free(recs); free(chain); }
    }
}

char *mtrh_format_result(const mtrh_result *r, int first_read, int last_read, size_t *out_len)
{
    fmt_ctx f; f.r = r; f.starts = read_starts(r); f.chain_first = NULL;
    *out_len = 0;
    if (!f.starts) return NULL;
    int64_t *cf = NULL;
    if (r->with_alignments) {
        cf = (int64_t *)malloc(sizeof(int64_t) * ((size_t)r->n_report + 1));
        cf[0] = 0;
        for (int i = 0; i < r->n_report; i++) cf[i + 1] = cf[i] + r->chain_len[i];
        f.chain_first = cf;
    }
    tbuf b = { NULL, 0, 0 };
    tb_room(&b, 1);
    if (last_read > r->n_report) last_read = r->n_report;
    format_reads(&f, first_read, last_read, &b);
    free((void *)f.starts); free(cf);
    *out_len = b.n;
    return b.s;
}

/* ---- the printer: one manager thread takes results in order; a pool formats slices of a result; the manager writes ---- */
#define MAX_PRINT_THREADS 32
typedef struct qnode { mtrh_result *r; struct qnode *next; } qnode;
struct mtrh_printer {
    FILE *out; int threads;
    pthread_t manager, pool[MAX_PRINT_THREADS];
    pthread_mutex_t mu; pthread_cond_t cv_q, cv_job, cv_done;
    qnode *head, *tail; int closing; int queued;
    /* the result being formatted */
    fmt_ctx job; int n_slices, next_slice, done_slices; int slice_first[MAX_PRINT_THREADS * 4 + 1]; tbuf slice_buf[MAX_PRINT_THREADS * 4];
    int pool_exit;
    int oom;                                           /* a thread of the pool could not allocate while it formatted a slice */
    int status, ended, cur_file; double t_chain;
};

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static void *pool_main(void *arg)
{
    mtrh_printer *p = (mtrh_printer *)arg;
    mtrh_thread_kind = MTRH_THREAD_PRINTER;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (!p->pool_exit && p->next_slice >= p->n_slices) pthread_cond_wait(&p->cv_job, &p->mu);
        if (p->pool_exit) break;
        const int s = p->next_slice++;
        pthread_mutex_unlock(&p->mu);
        jmp_buf oom;                                   /* (alloc.c: a failed allocation inside format_reads comes back here) */
        const int failed = setjmp(oom);
        if (!failed) { mtrh_oom_target = &oom; format_reads(&p->job, p->slice_first[s], p->slice_first[s + 1], &p->slice_buf[s]); }
        mtrh_oom_target = NULL;
        pthread_mutex_lock(&p->mu);
        if (failed) p->oom = 1;
        if (++p->done_slices == p->n_slices) pthread_cond_broadcast(&p->cv_done);
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

/* what the reference prints when its reader stops (handle_one_file.c:185, :243-246) */
static void end_message(const mtrh_batch *bt)
{
    if (bt->end == MTRH_END_BADCHAR) fprintf(stderr, "Invalid character: %c \n", bt->bad_char);
    else if (bt->end == MTRH_END_TOOLONG)
        fprintf(stderr, "fatal error: The length %d is tentatively at most %i.\nread ID = %.*s\nSet MAX_INPUT_LENGTH to a larger value",
                MTR_MAX_INPUT_LENGTH, MTR_MAX_INPUT_LENGTH, (int)bt->end_id_len, bt->end_id ? bt->end_id : "");
}

static void print_one(mtrh_printer *p, mtrh_result *r)
{
    const double t0 = now_s();
    const mtrh_batch *bt = r->batch;
    const int n = r->n_report;
    fmt_ctx f; f.r = r; f.starts = read_starts(r); f.chain_first = NULL;
    int64_t *cf = NULL;
    if (!f.starts) { fprintf(stderr, "internal error: malformed record table\n"); p->status = 1; p->ended = 1; return; }
    if (r->with_alignments) {
        cf = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
        cf[0] = 0;
        for (int i = 0; i < n; i++) cf[i + 1] = cf[i] + r->chain_len[i];
        f.chain_first = cf;
    }
    int slices = p->threads > 1 && n >= 256 ? p->threads * 4 : 1;
    if (slices == 1) {
        tbuf b = { NULL, 0, 0 };
        format_reads(&f, 0, n, &b);
        if (b.n) fwrite(b.s, 1, b.n, p->out);
        free(b.s);
    } else {
        pthread_mutex_lock(&p->mu);
        p->job = f; p->done_slices = 0; p->oom = 0;            /* (a failure belongs to the result whose slices it hit, not to every later one) */
        for (int s = 0; s <= slices; s++) p->slice_first[s] = (int)((int64_t)n * s / slices);
        for (int s = 0; s < slices; s++) p->slice_buf[s].n = 0;
        p->n_slices = slices; p->next_slice = 0;
        pthread_cond_broadcast(&p->cv_job);
        while (p->done_slices < slices) pthread_cond_wait(&p->cv_done, &p->mu);
        const int pool_failed = p->oom;
        pthread_mutex_unlock(&p->mu);
        if (pool_failed) { free((void *)f.starts); free(cf); mtrh_oom(0); }       /* (reported by manager_main like its own failure) */
        for (int s = 0; s < slices; s++) if (p->slice_buf[s].n) fwrite(p->slice_buf[s].s, 1, p->slice_buf[s].n, p->out);
    }
    free((void *)f.starts); free(cf);
    p->t_chain += now_s() - t0;
    if (r->fatal) {                                     /* a device-side error: like the reference, everything before it is out first */
        fflush(p->out);
        fprintf(stderr, "%s\n", r->fatal_msg ? r->fatal_msg : "device error");
        p->status = 1; p->ended = 1;
    } else if (bt->end != MTRH_END_NONE) {
        fflush(p->out);
        end_message(bt);
        p->ended = 1;
        if (bt->end != MTRH_END_EMPTY) p->status = 1;
    }
}

static void *manager_main(void *arg)
{
    mtrh_printer *p = (mtrh_printer *)arg;
    mtrh_thread_kind = MTRH_THREAD_PRINTER;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (!p->head && !p->closing) pthread_cond_wait(&p->cv_q, &p->mu);
        qnode *q = p->head;
        if (!q) { pthread_mutex_unlock(&p->mu); break; }
        p->head = q->next; if (!p->head) p->tail = NULL;
        p->queued--;
        pthread_cond_broadcast(&p->cv_q);
        pthread_mutex_unlock(&p->mu);
        if (q->r->file_idx != p->cur_file) { p->cur_file = q->r->file_idx; p->ended = 0; }    /* every file is a run of its own */
        if (!p->ended) {
            jmp_buf oom;                               /* (alloc.c) the printer could not allocate: what is out stays out, the run ends with status 1 */
            if (setjmp(oom) == 0) { mtrh_oom_target = &oom; print_one(p, q->r); }
            else { fflush(p->out); fprintf(stderr, "fatal error: cannot allocate memory\n"); p->status = 1; p->ended = 1; }
            mtrh_oom_target = NULL;
        }
        mtrh_result_free(q->r);
        free(q);
    }
    fflush(p->out);
    return NULL;
}

mtrh_printer *mtrh_printer_start(FILE *out, int threads)
{
    mtrh_printer *p = (mtrh_printer *)calloc(1, sizeof *p);
    if (!p) return NULL;
    if (threads < 1) threads = 1;
    if (threads > MAX_PRINT_THREADS) threads = MAX_PRINT_THREADS;
    p->out = out; p->threads = threads; p->cur_file = -1;
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->cv_q, NULL); pthread_cond_init(&p->cv_job, NULL); pthread_cond_init(&p->cv_done, NULL);
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_create(&p->pool[t], NULL, pool_main, p);
    pthread_create(&p->manager, NULL, manager_main, p);
    return p;
}

mtrh_printer *mtrh_printer_start_stdout(int threads) { return mtrh_printer_start(stdout, threads); }
/* the report on a file descriptor of the caller's: the launcher keeps the process's fd 1 pointed at stderr, because communication
 * libraries write banners there whenever they first connect (RCCL: at the first collective, not at the group's creation) */
mtrh_printer *mtrh_printer_start_fd(int fd, int threads)
{
    FILE *f = fdopen(fd, "w");
    return f ? mtrh_printer_start(f, threads) : NULL;
}

void mtrh_printer_push(mtrh_printer *p, mtrh_result *r)
{
    qnode *q = (qnode *)calloc(1, sizeof *q);
    q->r = r;
    pthread_mutex_lock(&p->mu);
    while (p->queued >= 4) pthread_cond_wait(&p->cv_q, &p->mu);        /* bounded: results hold the reads' bases */
    if (p->tail) p->tail->next = q; else p->head = q;
    p->tail = q; p->queued++;
    pthread_cond_broadcast(&p->cv_q);
    pthread_mutex_unlock(&p->mu);
}

int mtrh_printer_finish(mtrh_printer *p, double *t_chain)
{
    pthread_mutex_lock(&p->mu);
    p->closing = 1;
    pthread_cond_broadcast(&p->cv_q);
    pthread_mutex_unlock(&p->mu);
    pthread_join(p->manager, NULL);
    pthread_mutex_lock(&p->mu);
    p->pool_exit = 1;
    pthread_cond_broadcast(&p->cv_job);
    pthread_mutex_unlock(&p->mu);
    if (p->threads > 1) for (int t = 0; t < p->threads; t++) pthread_join(p->pool[t], NULL);
    for (int s = 0; s < MAX_PRINT_THREADS * 4; s++) free(p->slice_buf[s].s);
    const int st = p->status;
    if (t_chain) *t_chain = p->t_chain;
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_q); pthread_cond_destroy(&p->cv_job); pthread_cond_destroy(&p->cv_done);
    free(p);
    return st;
}
