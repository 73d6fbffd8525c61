/* null_engine.c — TEST / MEASUREMENT INFRASTRUCTURE, not product code.  A stand-in for libmtr_hip.so that costs the host nothing:
 * every batch is "finished" the moment it is launched, and every read is answered with the same K records (MTR_NULL_RECORDS, default 2) of a
 * unit of P bases (MTR_NULL_PERIOD, default 100) out of one table built once per process - no per-base work, no hashing, no copy (what the
 * real engine's DMA engines do, costs a host thread nothing either).  Behind `mTR -g N` it measures the ceiling of the HOST pipeline alone:
 * cutting the file at "\n>", parsing + 2-bit packing (fasta.c), the batches' way through the runs (pipeline.c), the round's exchange
 * (multi.c), wire unpack + chaining + formatting (chain.c, print.c) and the single writer - the part of handle_one_file.c:201-293 and
 * chaining.cpp:243-363 that stays on the CPU however fast the GPUs are.  bench.py: `host_ceiling`; tests/test_host_ceiling.py.
 * Nothing under mtr_amd/ refers to it; it is handed to the host through MTR_LIB like the replay engine.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mtr_hip.h"

#define NULL_MAX_READS (1 << 16)
struct mtr_ctx { char err[128]; int32_t n; int ran; };
struct mtr_file_state { int dummy; };

static pthread_once_t once = PTHREAD_ONCE_INIT;
static uint8_t *g_blob; static int32_t *g_counts; static int64_t g_per_read; static int g_k, g_p;

static void build_table(void)
{
    g_k = getenv("MTR_NULL_RECORDS") ? atoi(getenv("MTR_NULL_RECORDS")) : 2;
    g_p = getenv("MTR_NULL_PERIOD") ? atoi(getenv("MTR_NULL_PERIOD")) : 100;
    if (g_k < 0) g_k = 0;
    if (g_k > 8) g_k = 8;
    if (g_p < 2) g_p = 2;
    if (g_p > MTR_MAX_PERIOD) g_p = MTR_MAX_PERIOD;
    const int64_t rb = mtr_wire_record_bytes(g_p);
    g_per_read = rb * g_k;
    g_blob = (uint8_t *)calloc((size_t)(g_per_read * NULL_MAX_READS) + 8, 1);
    g_counts = (int32_t *)malloc(sizeof(int32_t) * NULL_MAX_READS);
    uint8_t *one = (uint8_t *)calloc((size_t)g_per_read + 8, 1);
    for (int t = 0; t < g_k; t++) {
        /* K repeats side by side (the chain keeps them all): 10 copies of the unit each, the counts of a 90 % match */
        const int32_t start = 50 + t * (12 * g_p), len = 10 * g_p;
        const int32_t h[14] = { start, start + len - 1, len, g_p, 10, len * 9 / 10, len / 50, len / 20, len / 33, 7, 1, 1, 3, 0 };
        uint8_t *p = one + rb * t;
        memcpy(p, h, sizeof h);
        for (int j = 0; j < g_p; j++) p[MTR_WIRE_HEADER_BYTES + j] = (uint8_t)"ACGT"[(j * 7 + j / 3) & 3];
        int32_t *sc = (int32_t *)(p + MTR_WIRE_HEADER_BYTES + ((g_p + 3) & ~3));
        for (int j = 0; j < g_p; j++) sc[j] = 9 + (j & 1);
    }
    for (int i = 0; i < NULL_MAX_READS; i++) { memcpy(g_blob + g_per_read * i, one, (size_t)g_per_read); g_counts[i] = g_k; }
    free(one);
}

int mtr_abi_version(void) { return MTR_ABI_VERSION; }
const char *mtr_last_error(const mtr_ctx *c) { return c ? c->err : "no context"; }
mtr_status mtr_create(int device, int manhattan, float r, mtr_ctx **out)
{
    (void)device; (void)manhattan; (void)r;
    if (!out) return MTR_ERR_BAD_ARG;
    pthread_once(&once, build_table);
    *out = (mtr_ctx *)calloc(1, sizeof(mtr_ctx));
    return *out ? MTR_OK : MTR_ERR_OOM;
}
void mtr_destroy(mtr_ctx *c) { free(c); }
mtr_status mtr_upload_batch_packed(mtr_ctx *c, const uint32_t *packed, int64_t n_words, const int64_t *woff, const int32_t *lens, int32_t n)
{
    if (!c || !packed || !woff || !lens || n <= 0 || n > NULL_MAX_READS || n_words <= 0) return MTR_ERR_BAD_ARG;
    c->n = n; c->ran = 0;
    return MTR_OK;
}
mtr_status mtr_upload_batch_in_file(mtr_ctx *c, mtr_file_state *fs, const uint8_t *bases, const int64_t *offsets, const int32_t *lens, int32_t n)
{
    if (!c || !fs || !bases || !offsets || !lens || n <= 0 || n > NULL_MAX_READS) return MTR_ERR_BAD_ARG;
    c->n = n; c->ran = 0;
    return MTR_OK;
}
mtr_status mtr_run_resident_async(mtr_ctx *c) { if (!c || c->n <= 0) return MTR_ERR_BAD_ARG; c->ran = 1; return MTR_OK; }
mtr_status mtr_wait(mtr_ctx *c) { return c ? MTR_OK : MTR_ERR_BAD_ARG; }
mtr_status mtr_get_first_failed_read(const mtr_ctx *c, int32_t *out) { if (!c || !out) return MTR_ERR_BAD_ARG; *out = -1; return MTR_OK; }
mtr_status mtr_fetch_results_packed(mtr_ctx *c, int32_t limit, const uint8_t **out_blob, int64_t *out_bytes, const int32_t **out_counts, int64_t *out_total)
{
    if (!c || !c->ran) return MTR_ERR_BAD_ARG;
    int n = c->n; if (limit >= 0 && limit < n) n = limit;
    *out_blob = g_blob; *out_bytes = g_per_read * n; *out_counts = g_counts; if (out_total) *out_total = (int64_t)g_k * n;
    return MTR_OK;
}
mtr_status mtr_get_bases_after_read(const mtr_ctx *c, int32_t i, uint8_t out[2]) { (void)c; (void)i; out[0] = out[1] = 0; return MTR_OK; }
mtr_status mtr_get_kernel_times(const mtr_ctx *c, mtr_kernel_time *out, int32_t n) { (void)c; for (int i = 0; i < n; i++) { out[i].ms = 0; out[i].launches = 0; } return MTR_OK; }
mtr_status mtr_get_counters(const mtr_ctx *c, int64_t *out, int32_t n) { (void)c; for (int i = 0; i < n; i++) out[i] = 0; return MTR_OK; }
mtr_status mtr_file_state_create(mtr_file_state **out) { *out = (mtr_file_state *)calloc(1, sizeof(mtr_file_state)); return MTR_OK; }
void mtr_file_state_destroy(mtr_file_state *fs) { free(fs); }
mtr_status mtr_file_state_skip(mtr_file_state *fs, const uint8_t *b, const int64_t *o, const int32_t *l, int32_t n) { (void)fs; (void)b; (void)o; (void)l; (void)n; return MTR_OK; }
mtr_status mtr_alignments(mtr_ctx *c, int32_t n, const int32_t *read_idx, const mtr_record *records, uint8_t **out_ops, int64_t **out_off, int32_t **out_end)
{   /* (-a is not what this engine is for: empty paths) */
    (void)c; (void)read_idx; (void)records;
    *out_ops = (uint8_t *)malloc(1); *out_off = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t)); *out_end = (int32_t *)calloc((size_t)n * 2 + 2, sizeof(int32_t));
    return MTR_OK;
}

/* the gather of ABI 5: a ticket is the number of reads of the staged batch; the exchange hands out the shared table (no copy: the product's tables arrive by DMA) */
struct mtr_gather { int n; long long exchanges, bytes; };
mtr_status mtr_device_count(int32_t *out) { if (!out) return MTR_ERR_BAD_ARG; const char *e = getenv("MTR_REPLAY_DEVICES"); *out = e ? atoi(e) : 8; return *out > 0 ? MTR_OK : MTR_ERR_NO_DEVICE; }
const char *mtr_gather_last_error(const mtr_gather *g) { (void)g; return ""; }
mtr_status mtr_gather_create(int32_t n, const int32_t *devices, mtr_gather **out)
{
    if (!out || n <= 0 || !devices) return MTR_ERR_BAD_ARG;
    *out = (mtr_gather *)calloc(1, sizeof(mtr_gather)); (*out)->n = n;
    return MTR_OK;
}
mtr_status mtr_gather_wait_ready(mtr_gather *g) { return g ? MTR_OK : MTR_ERR_BAD_ARG; }
mtr_status mtr_gather_get_stats(const mtr_gather *g, int64_t *out, int32_t n)
{
    if (!g || !out) return MTR_ERR_BAD_ARG;
    const int64_t v[6] = { g->exchanges, 0, g->bytes, 0, 0, 1 };
    for (int i = 0; i < n && i < 6; i++) out[i] = v[i];
    return MTR_OK;
}
void mtr_gather_destroy(mtr_gather *g) { free(g); }
mtr_status mtr_gather_stage(mtr_gather *g, int32_t rank, mtr_ctx *c, int32_t *counts_host, int64_t *out_total, int64_t *out_bytes, int32_t *out_ticket)
{
    if (!g || !c || !counts_host || !out_total || !out_bytes || !out_ticket || rank < 0 || rank >= g->n) return MTR_ERR_BAD_ARG;
    for (int i = 0; i < c->n; i++) counts_host[i] = g_k;
    *out_total = (int64_t)g_k * c->n; *out_bytes = g_per_read * c->n; *out_ticket = c->n;
    return MTR_OK;
}
mtr_status mtr_gather_exchange(mtr_gather *g, int32_t n, const int32_t *tickets, const uint8_t **out_ptrs, int64_t *out_bytes)
{
    if (!g || n < 0 || (n > 0 && (!tickets || !out_ptrs || !out_bytes))) return MTR_ERR_BAD_ARG;
    for (int i = 0; i < n; i++) { out_ptrs[i] = g_blob; out_bytes[i] = g_per_read * tickets[i]; __atomic_fetch_add(&g->bytes, out_bytes[i], __ATOMIC_RELAXED); }
    __atomic_fetch_add(&g->exchanges, 1, __ATOMIC_RELAXED);
    return MTR_OK;
}
