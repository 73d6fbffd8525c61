/* shard.c — the results of one rank on their way to rank 0 ("gather of per-read record lists to rank 0,
 * rank 0 chains + prints in input order").  A result becomes one self-contained byte string: read IDs and lengths, the
 * per-read record counts, the record table in wire form (include/mtr_hip.h) and, with -a, the chains and alignment paths
 * made where the batch was resident plus the reads' bases at 2 bit/base (the printer shows them above the unit).  The
 * launcher (mtr_amd/run.py) moves the bytes with torch.distributed (RCCL on the GPU box, gloo in CPU tests) and does
 * nothing else with them.
 */
#define _GNU_SOURCE
#include "mtr_host.h"
#include <stdlib.h>
#include <string.h>

#define SHARD_MAGIC 0x3252544d52485300LL        /* "\0SHRMTR2" */
enum { H_MAGIC = 0, H_BYTES, H_CHUNK, H_FILE, H_LAST, H_N, H_NREPORT, H_ALIGN, H_FATAL, H_END, H_BADCHAR, H_WIRE, H_IDS, H_WORDS,
       H_NCHAIN, H_OPS, H_MSG, H_ENDID, H_KERNEL_US, H_QUERIES, H_N_FIELDS = 24 };

static size_t pad8(size_t n) { return (n + 7) & ~(size_t)7; }

static void *xmalloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "cannot allocate %zu bytes\n", n); exit(EXIT_FAILURE); }
    return p;
}

uint8_t *mtrh_result_serialize(const mtrh_result *r, size_t *out_bytes)
{
    const mtrh_batch *b = r->batch;
    const int n = b->n, nr = r->n_report;
    const int al = r->with_alignments && nr > 0;
    int64_t ids = 0, words = 0;
    for (int i = 0; i < n; i++) ids += b->id_lens[i];
    if (al) for (int i = 0; i < n; i++) words += mtr_packed_words(b->lens[i]);
    const int64_t nk = al ? r->n_chain : 0, ops = al ? r->ops_off[nk] : 0;
    const size_t msg = r->fatal_msg ? strlen(r->fatal_msg) : 0, endid = b->end_id ? (size_t)b->end_id_len : 0;
    size_t total = sizeof(int64_t) * H_N_FIELDS;
    total += pad8(4 * (size_t)n) * 2 + pad8(4 * (size_t)nr) + pad8((size_t)ids) + pad8((size_t)r->wire_bytes);
    if (al) total += pad8(4 * (size_t)words) + pad8(4 * (size_t)nr) + pad8(4 * (size_t)nk) + 8 * ((size_t)nk + 1) + pad8((size_t)ops) + pad8(8 * (size_t)nk) + pad8(2 * (size_t)n);
    total += pad8(msg) + pad8(endid);
    uint8_t *out = (uint8_t *)xmalloc(total);
    memset(out, 0, total);
    int64_t *h = (int64_t *)out;
    h[H_MAGIC] = SHARD_MAGIC; h[H_BYTES] = (int64_t)total; h[H_CHUNK] = r->chunk; h[H_FILE] = r->file_idx; h[H_LAST] = r->last_of_chunk;
    h[H_N] = n; h[H_NREPORT] = nr; h[H_ALIGN] = al; h[H_FATAL] = r->fatal; h[H_END] = b->end; h[H_BADCHAR] = (unsigned char)b->bad_char;
    h[H_WIRE] = r->wire_bytes; h[H_IDS] = ids; h[H_WORDS] = words; h[H_NCHAIN] = nk; h[H_OPS] = ops; h[H_MSG] = (int64_t)msg; h[H_ENDID] = (int64_t)endid;
    h[H_KERNEL_US] = (int64_t)(r->t_kernel_ms * 1000.0); h[H_QUERIES] = r->queries;
    uint8_t *p = out + sizeof(int64_t) * H_N_FIELDS;
#define PUT(src, bytes) do { if ((bytes) > 0) memcpy(p, (src), (size_t)(bytes)); p += pad8((size_t)(bytes)); } while (0)
    PUT(b->lens, 4 * (size_t)n);
    PUT(b->id_lens, 4 * (size_t)n);
    PUT(r->counts, 4 * (size_t)nr);
    { uint8_t *q = p; for (int i = 0; i < n; i++) { memcpy(q, b->ids[i], (size_t)b->id_lens[i]); q += b->id_lens[i]; } p += pad8((size_t)ids); }
    PUT(r->wire, r->wire_bytes);
    if (al) {
        uint32_t *w = (uint32_t *)p;
        for (int i = 0; i < n; i++) { (void)mtr_pack_read(b->codes + b->offs[i], b->lens[i], w); w += mtr_packed_words(b->lens[i]); }
        p += pad8(4 * (size_t)words);
        PUT(r->chain_len, 4 * (size_t)nr);
        PUT(r->chain_idx, 4 * (size_t)nk);
        PUT(r->ops_off, 8 * ((size_t)nk + 1));
        PUT(r->ops, ops);
        PUT(r->ends, 8 * (size_t)nk);
        PUT(r->after, 2 * (size_t)n);
    }
    PUT(r->fatal_msg, msg);
    PUT(b->end_id, endid);
#undef PUT
    *out_bytes = total;
    return out;
}

mtrh_result *mtrh_result_deserialize(const uint8_t *blob, size_t bytes, size_t *used)
{
    if (bytes < sizeof(int64_t) * H_N_FIELDS) return NULL;
    const int64_t *h = (const int64_t *)blob;
    if (h[H_MAGIC] != SHARD_MAGIC || h[H_BYTES] < (int64_t)(sizeof(int64_t) * H_N_FIELDS) || (size_t)h[H_BYTES] > bytes) return NULL;
    const int n = (int)h[H_N], nr = (int)h[H_NREPORT], al = (int)h[H_ALIGN];
    const int64_t ids = h[H_IDS], words = h[H_WORDS], nk = h[H_NCHAIN], ops = h[H_OPS];
    if (n < 0 || nr < 0 || nr > n || ids < 0 || words < 0 || nk < 0 || ops < 0 || h[H_WIRE] < 0 || h[H_MSG] < 0 || h[H_ENDID] < 0) return NULL;
    {   /* the sections the header announces must add up to the blob BEFORE anything is copied out of it */
        size_t need = sizeof(int64_t) * H_N_FIELDS + 2 * pad8(4 * (size_t)n) + pad8(4 * (size_t)nr) + pad8((size_t)ids) + pad8((size_t)h[H_WIRE])
                      + pad8((size_t)h[H_MSG]) + pad8((size_t)h[H_ENDID]);
        if (al) need += pad8(4 * (size_t)words) + pad8(4 * (size_t)nr) + pad8(4 * (size_t)nk) + pad8(8 * ((size_t)nk + 1)) + pad8((size_t)ops)
                        + pad8(4 * 2 * (size_t)nk) + pad8(2 * (size_t)n);
        if (need != (size_t)h[H_BYTES]) return NULL;
    }
    mtrh_result *r = (mtrh_result *)calloc(1, sizeof *r);
    r->ticket = -1;
    mtrh_batch *b = (mtrh_batch *)calloc(1, sizeof *b);
    r->batch = b;
    r->chunk = (int32_t)h[H_CHUNK]; r->file_idx = (int32_t)h[H_FILE]; r->last_of_chunk = (int32_t)h[H_LAST];
    r->n_report = nr; r->with_alignments = al; r->fatal = (int)h[H_FATAL];
    r->t_kernel_ms = (double)h[H_KERNEL_US] / 1000.0; r->queries = h[H_QUERIES];
    b->n = n; b->end = (int)h[H_END]; b->bad_char = (char)h[H_BADCHAR];
    const uint8_t *p = blob + sizeof(int64_t) * H_N_FIELDS;
#define GET(dst, type, count) do { (dst) = (type *)xmalloc(sizeof(type) * ((size_t)(count) + 1)); memcpy((dst), p, sizeof(type) * (size_t)(count)); p += pad8(sizeof(type) * (size_t)(count)); } while (0)
    GET(b->lens, int32_t, n);
    GET(b->id_lens, int32_t, n);
    GET(r->counts, int32_t, nr);
    b->id_store = (char *)xmalloc((size_t)ids + 1); memcpy(b->id_store, p, (size_t)ids); p += pad8((size_t)ids);
    b->ids = (const char **)xmalloc(sizeof(char *) * ((size_t)n + 1));
    { const char *q = b->id_store; for (int i = 0; i < n; i++) { b->ids[i] = q; q += b->id_lens[i]; } }
    r->wire_bytes = h[H_WIRE]; r->wire = (uint8_t *)xmalloc((size_t)r->wire_bytes + 8); memcpy(r->wire, p, (size_t)r->wire_bytes); p += pad8((size_t)r->wire_bytes);
    if (al) {
        /* the reads' bases come at 2 bit/base; the printer wants codes */
        int64_t total = 0;
        b->offs = (int64_t *)xmalloc(sizeof(int64_t) * ((size_t)n + 1));
        int64_t wsum = 0;
        for (int i = 0; i < n; i++) { b->offs[i] = total; if (b->lens[i] < 0) { wsum = -1; break; } total += b->lens[i]; wsum += mtr_packed_words(b->lens[i]); }
        if (wsum != words) { mtrh_result_free(r); return NULL; }        /* the lengths do not describe the packed image that follows */
        b->codes = (uint8_t *)xmalloc((size_t)total + 1);
        const uint32_t *w = (const uint32_t *)p;
        for (int i = 0; i < n; i++) {
            uint8_t *d = b->codes + b->offs[i];
            for (int q = 0; q < b->lens[i]; q++) d[q] = (uint8_t)((w[q >> 4] >> (30 - 2 * (q & 15))) & 3u);
            w += mtr_packed_words(b->lens[i]);
        }
        p += pad8(4 * (size_t)words);
        GET(r->chain_len, int32_t, nr);
        GET(r->chain_idx, int32_t, nk); r->n_chain = nk;
        GET(r->ops_off, int64_t, nk + 1);
        GET(r->ops, uint8_t, ops);
        GET(r->ends, int32_t, 2 * nk);
        GET(r->after, uint8_t, 2 * (size_t)n);
    }
    if (h[H_MSG] > 0 || r->fatal) { r->fatal_msg = (char *)xmalloc((size_t)h[H_MSG] + 1); memcpy(r->fatal_msg, p, (size_t)h[H_MSG]); r->fatal_msg[h[H_MSG]] = 0; }
    p += pad8((size_t)h[H_MSG]);
    if (h[H_ENDID] > 0) {
        /* keep the ID behind the read IDs' store so that one free releases both */
        b->id_store = (char *)realloc(b->id_store, (size_t)ids + (size_t)h[H_ENDID] + 1);
        { const char *q = b->id_store; for (int i = 0; i < n; i++) { b->ids[i] = q; q += b->id_lens[i]; } }
        memcpy(b->id_store + ids, p, (size_t)h[H_ENDID]);
        b->end_id = b->id_store + ids; b->end_id_len = (int32_t)h[H_ENDID];
    }
    p += pad8((size_t)h[H_ENDID]);
#undef GET
    if ((size_t)(p - blob) != (size_t)h[H_BYTES]) { mtrh_result_free(r); return NULL; }
    if (used) *used = (size_t)h[H_BYTES];
    return r;
}

/* ---- a round: what one gather carries ------------------------------------------------------------------------------------ */
/* all results this rank produces for `round`, serialised one after the other (malloc'ed; *bytes = 0 if it owns nothing) */
uint8_t *mtrh_run_round_blob(mtrh_run *run, int round, size_t *bytes)
{
    int owned = 0;
    const int nc = mtrh_run_n_chunks(run);
    for (int c = 0; c < nc; c++) if (mtrh_run_round_of(run, c) == round && mtrh_run_owner(run, c) == mtrh_run_rank(run)) owned++;
    uint8_t *out = NULL; size_t n = 0, cap = 0;
    while (owned > 0) {
        mtrh_result *x = mtrh_run_next(run);
        if (!x) break;
        size_t sz = 0;
        uint8_t *s = mtrh_result_serialize(x, &sz);
        if (n + sz > cap) { cap = cap ? cap * 2 : sz + (1 << 16); while (cap < n + sz) cap *= 2; out = (uint8_t *)realloc(out, cap); if (!out) { fprintf(stderr, "cannot allocate the round\n"); exit(EXIT_FAILURE); } }
        memcpy(out + n, s, sz); n += sz;
        free(s);
        if (x->last_of_chunk) owned--;
        mtrh_result_free(x);
    }
    *bytes = n;
    return out ? out : (uint8_t *)xmalloc(1);
}

static int by_chunk(const void *a, const void *b)
{
    const mtrh_result *x = *(const mtrh_result *const *)a, *y = *(const mtrh_result *const *)b;
    if (x->chunk != y->chunk) return x->chunk < y->chunk ? -1 : 1;
    return x->file_idx - y->file_idx;     /* equal chunks keep their order below (sequence numbers) */
}

/* rank 0: the blobs of every rank for one round -> results in output order -> the printer.  Returns the number of
 * results, or -1 if a blob is malformed. */
int mtrh_print_round(mtrh_printer *pr, const uint8_t *const *blobs, const size_t *sizes, int n_blobs)
{
    mtrh_result **all = NULL; int n = 0, cap = 0;
    for (int k = 0; k < n_blobs; k++) {
        size_t o = 0;
        while (o < sizes[k]) {
            size_t used = 0;
            mtrh_result *x = mtrh_result_deserialize(blobs[k] + o, sizes[k] - o, &used);
            if (!x) { for (int i = 0; i < n; i++) mtrh_result_free(all[i]); free(all); return -1; }
            if (n == cap) { cap = cap ? cap * 2 : 64; all = (mtrh_result **)realloc(all, sizeof(*all) * (size_t)cap); }
            all[n++] = x;
            o += used;
        }
    }
    /* stable by chunk: the results of one chunk come from one rank, already in order (insertion sort: n is small) */
    for (int a = 1; a < n; a++) { mtrh_result *t = all[a]; int b = a; while (b > 0 && by_chunk(&all[b - 1], &t) > 0) { all[b] = all[b - 1]; b--; } all[b] = t; }
    for (int i = 0; i < n; i++) mtrh_printer_push(pr, all[i]);
    free(all);
    return n;
}
