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
#include <time.h>
#include <unistd.h>

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

/* printf("%f", (float)matches / repeat_len) (chaining.cpp:136-137: the float quotient, promoted to double) without printf: a float is M x 2^e with a 24-bit M,
 * so M x 10^6 is an exact 44-bit integer and the six decimals are its quotient by 2^-e, rounded half to even on the exact remainder - what glibc's
 * correctly rounded conversion prints.  snprintf was a third of the formatting time of a report line.  Anything unusual (negative, >= 2^24, inf, nan:
 * a zero repeat_len) goes to snprintf.  out must hold 64 bytes; returns the number of characters.  tests/test_host_ceiling.py compares the two. */
int mtrh_format_ratio(int matches, int repeat_len, char *out)
{
    const float f = (float)matches / repeat_len;
    uint32_t bits; memcpy(&bits, &f, 4);
    const int ex = (int)((bits >> 23) & 0xffu);
    if ((bits >> 31) || ex == 255 || ex >= 150 + 24) return snprintf(out, 64, "%f", f);
    uint64_t q;                                        /* the value x 10^6, rounded */
    if (ex == 0) q = 0;                                /* zero and subnormals: below 10^-37 */
    else {
        const uint64_t N = (uint64_t)((bits & 0x7fffffu) | 0x800000u) * 1000000ull;
        const int e = ex - 150;                        /* value = M x 2^e */
        if (e >= 0) q = N << e;                        /* (M < 2^24, e < 24: an integer below 2^48, times 10^6 below 2^64 only for e <= 20 - larger: snprintf) */
        else if (-e > 62) q = 0;
        else {
            const int s = -e;
            const uint64_t rem = N & ((1ull << s) - 1ull), half = 1ull << (s - 1);
            q = N >> s;
            if (rem > half || (rem == half && (q & 1ull))) q++;
        }
        if (e > 20) return snprintf(out, 64, "%f", f);
    }
    uint64_t ip = q / 1000000ull; unsigned fr = (unsigned)(q % 1000000ull);
    char t[24]; int k = 0, n = 0;
    do { t[k++] = (char)('0' + ip % 10); ip /= 10; } while (ip);
    while (k) out[n++] = t[--k];
    out[n++] = '.';
    for (int d = 5; d >= 0; d--) { out[n + d] = (char)('0' + fr % 10); fr /= 10; }
    n += 6;
    out[n] = 0;
    return n;
}

/* chaining.cpp:127-143: ID L start+1 end+1 repeat_len period copies matches ratio mismatches insertions deletions unit */
static inline char *put_int(char *o, int v)
{   /* "%d" */
    char t[16]; int k = 0; unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { t[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *o++ = '-';
    while (k) *o++ = t[--k];
    return o;
}
static void report_line(tbuf *b, const char *id, int id_len, int L, const mtrh_rec *r)
{
    int per = r->h[MTRH_PERIOD]; if (per < 0) per = 0;
    tb_room(b, (size_t)id_len + (size_t)per + 256);       /* ONE check per line: ten integers of at most 11 characters, the ratio (at most 64), twelve tabs */
    char *o = b->s + b->n;
    memcpy(o, id, (size_t)id_len); o += id_len; *o++ = '\t';
    o = put_int(o, L); *o++ = '\t';
    o = put_int(o, r->h[MTRH_REP_START] + 1); *o++ = '\t';
    o = put_int(o, r->h[MTRH_REP_END] + 1); *o++ = '\t';
    o = put_int(o, r->h[MTRH_REPEAT_LEN]); *o++ = '\t';
    o = put_int(o, r->h[MTRH_PERIOD]); *o++ = '\t';
    o = put_int(o, r->h[MTRH_COPIES]); *o++ = '\t';
    o = put_int(o, r->h[MTRH_MATCHES]); *o++ = '\t';
    o += mtrh_format_ratio(r->h[MTRH_MATCHES], r->h[MTRH_REPEAT_LEN], o);             /* a float division, printed as a double */
    *o++ = '\t';
    o = put_int(o, r->h[MTRH_MISMATCHES]); *o++ = '\t';
    o = put_int(o, r->h[MTRH_INSERTIONS]); *o++ = '\t';
    o = put_int(o, r->h[MTRH_DELETIONS]); *o++ = '\t';
    const size_t ul = strnlen(r->unit, (size_t)per);
    memcpy(o, r->unit, ul); o += ul; *o++ = '\n';
    b->n = (size_t)(o - b->s);
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

/* ---- the printer: results in output order through three stages - prepared and formatted by a pool, written by one thread ----
 * Round 6: up to `depth` results are in the printer at once.  A result is a job: one pool thread walks its wire form for the reads' starts
 * (read_starts: a serial walk over the records' headers), then the pool formats its slices - of THIS job and of the jobs behind it, whatever
 * is claimable -, and the writer thread emits finished jobs in order.  Before (one manager thread: format a result's slices, wait, write,
 * next result) the pool idled while the manager wrote and the writer idled while the pool formatted: 1.1-1.2 s per million reads whatever
 * the number of cores, the ceiling of `mTR -g 8` behind eight GPUs (measured with tests/null_engine.c).  What ends a file's output
 * (a device-side error, a fatal character, an empty record, a failed allocation) is decided by the writer, in order, as before. */
#define MAX_PRINT_THREADS 32
#define MAX_SLICES (MAX_PRINT_THREADS * 4)
typedef struct pjob {
    mtrh_result *r; fmt_ctx f; int64_t *cf;
    int prep_claimed, prepared, malformed, oom;
    int n_slices, next_slice, done_slices; int slice_first[MAX_SLICES + 1]; tbuf slice_buf[MAX_SLICES];
    struct pjob *next;
} pjob;
struct mtrh_printer {
    FILE *out; int threads, depth;
    pthread_t writer, pool[MAX_PRINT_THREADS];
    pthread_mutex_t mu; pthread_cond_t cv_q, cv_job, cv_done;
    pjob *head, *tail; int closing, queued, pool_exit;
    int status, ended, cur_file; double t_chain;
};

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* the reads' starts in the wire form and the slices of the job (any thread, outside the lock) */
static void job_prepare(pjob *j, int threads)
{
    mtrh_result *r = j->r; const int n = r->n_report;
    j->f.r = r; j->f.chain_first = NULL;
    j->f.starts = read_starts(r);
    if (!j->f.starts) { j->malformed = 1; j->n_slices = 0; return; }
    if (r->with_alignments) {
        j->cf = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
        j->cf[0] = 0;
        for (int i = 0; i < n; i++) j->cf[i + 1] = j->cf[i] + r->chain_len[i];
        j->f.chain_first = j->cf;
    }
    const int slices = threads > 1 && n >= 256 ? threads * 4 : 1;
    for (int s = 0; s <= slices; s++) j->slice_first[s] = (int)((int64_t)n * s / slices);
    j->n_slices = slices;
}

static void job_free(pjob *j)
{
    for (int s = 0; s < MAX_SLICES; s++) free(j->slice_buf[s].s);
    free((void *)j->f.starts); free(j->cf);
    mtrh_result_free(j->r);
    free(j);
}

/* a unit of work of the pool: the preparation of a job, or one slice of a prepared job - the oldest job first (the lock is held) */
static __attribute__((noinline)) pjob *claim_unit(mtrh_printer *p, int *unit)
{
    for (pjob *q = p->head; q; q = q->next) {
        if (!q->prep_claimed) { q->prep_claimed = 1; *unit = -1; return q; }
        if (q->prepared && q->next_slice < q->n_slices) { *unit = q->next_slice++; return q; }
    }
    return NULL;
}
static void *pool_main(void *arg)
{
    mtrh_printer *p = (mtrh_printer *)arg;
    mtrh_thread_kind = MTRH_THREAD_PRINTER;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        int unit = -2;                                 /* -1: prepare, >= 0: slice */
        pjob *const j = claim_unit(p, &unit);
        if (unit == -2) {
            if (p->pool_exit) break;
            pthread_cond_wait(&p->cv_job, &p->mu);
            continue;
        }
        pthread_mutex_unlock(&p->mu);
        jmp_buf oom;                                   /* (alloc.c: a failed allocation inside the unit comes back here) */
        const int failed = setjmp(oom);
        if (!failed) {
            mtrh_oom_target = &oom;
            if (unit < 0) job_prepare(j, p->threads);
            else format_reads(&j->f, j->slice_first[unit], j->slice_first[unit + 1], &j->slice_buf[unit]);
        }
        mtrh_oom_target = NULL;
        pthread_mutex_lock(&p->mu);
        if (failed) j->oom = 1;                        /* (a failure belongs to the result whose unit it hit, not to every later one) */
        if (unit < 0) {
            if (failed) j->n_slices = 0;
            j->prepared = 1;
            pthread_cond_broadcast(&p->cv_job);        /* its slices can be claimed */
            if (j->n_slices == 0) pthread_cond_broadcast(&p->cv_done);
        } else if (++j->done_slices == j->n_slices) pthread_cond_broadcast(&p->cv_done);
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

/* the writer's part of a finished job: its text, then whatever ends the file's output */
static void write_job(mtrh_printer *p, pjob *j)
{
    mtrh_result *r = j->r;
    const mtrh_batch *bt = r->batch;
    if (j->oom) { fflush(p->out); fprintf(stderr, "fatal error: cannot allocate memory\n"); p->status = 1; p->ended = 1; return; }
    if (j->malformed) { fprintf(stderr, "internal error: malformed record table\n"); p->status = 1; p->ended = 1; return; }
    for (int s = 0; s < j->n_slices; s++) if (j->slice_buf[s].n) fwrite(j->slice_buf[s].s, 1, j->slice_buf[s].n, p->out);
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

static void *writer_main(void *arg)
{
    mtrh_printer *p = (mtrh_printer *)arg;
    mtrh_thread_kind = MTRH_THREAD_PRINTER;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (!p->head && !p->closing) pthread_cond_wait(&p->cv_done, &p->mu);
        pjob *j = p->head;
        if (!j) { pthread_mutex_unlock(&p->mu); break; }
        const double t0 = now_s();
        if (p->threads <= 1) {                         /* no pool: this thread prepares and formats the job itself */
            pthread_mutex_unlock(&p->mu);
            jmp_buf oom;
            if (setjmp(oom) == 0) {
                mtrh_oom_target = &oom;
                job_prepare(j, 1);
                if (!j->malformed) format_reads(&j->f, 0, j->r->n_report, &j->slice_buf[0]);
            } else { j->oom = 1; j->n_slices = 0; }
            mtrh_oom_target = NULL;
            pthread_mutex_lock(&p->mu);
        } else while (!(j->prepared && j->done_slices == j->n_slices)) pthread_cond_wait(&p->cv_done, &p->mu);
        p->head = j->next; if (!p->head) p->tail = NULL;
        p->queued--;
        pthread_cond_broadcast(&p->cv_q);
        pthread_mutex_unlock(&p->mu);
        if (j->r->file_idx != p->cur_file) { p->cur_file = j->r->file_idx; p->ended = 0; }    /* every file is a run of its own */
        if (!p->ended) write_job(p, j);
        p->t_chain += now_s() - t0;
        job_free(j);
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
    p->depth = threads > 1 ? 4 + threads / 4 : 4;      /* results in the printer at once (bounded: they hold the reads' bases) */
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->cv_q, NULL); pthread_cond_init(&p->cv_job, NULL); pthread_cond_init(&p->cv_done, NULL);
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_create(&p->pool[t], NULL, pool_main, p);
    pthread_create(&p->writer, NULL, writer_main, p);
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
/* threads for a printer by the cores of the machine: a quarter of them, 2 .. 16 (1 on a single core) */
int mtrh_printer_default_threads(void)
{
    const long nc = sysconf(_SC_NPROCESSORS_ONLN);
    if (nc < 2) return 1;
    if (nc < 8) return 2;
    return nc / 2 > 16 ? 16 : (int)(nc / 2);
}

void mtrh_printer_push(mtrh_printer *p, mtrh_result *r)
{
    pjob *j = (pjob *)calloc(1, sizeof *j);
    j->r = r;
    pthread_mutex_lock(&p->mu);
    while (p->queued >= p->depth) pthread_cond_wait(&p->cv_q, &p->mu);
    if (p->tail) p->tail->next = j; else p->head = j;
    p->tail = j; p->queued++;
    pthread_cond_broadcast(&p->cv_job);
    pthread_cond_broadcast(&p->cv_done);
    pthread_mutex_unlock(&p->mu);
}

int mtrh_printer_finish(mtrh_printer *p, double *t_chain)
{
    pthread_mutex_lock(&p->mu);
    p->closing = 1;
    pthread_cond_broadcast(&p->cv_done);
    pthread_mutex_unlock(&p->mu);
    pthread_join(p->writer, NULL);
    pthread_mutex_lock(&p->mu);
    p->pool_exit = 1;
    pthread_cond_broadcast(&p->cv_job);
    pthread_mutex_unlock(&p->mu);
    if (p->threads > 1) for (int t = 0; t < p->threads; t++) pthread_join(p->pool[t], NULL);
    const int st = p->status | (ferror(p->out) ? 1 : 0);      /* a report that could not be written is a failed run */
    if (t_chain) *t_chain = p->t_chain;
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_q); pthread_cond_destroy(&p->cv_job); pthread_cond_destroy(&p->cv_done);
    free(p);
    return st;
}
