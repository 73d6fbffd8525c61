/* replay_engine.c — TEST INFRASTRUCTURE, not product code.  A stand-in for libmtr_hip.so that implements the part of
 * include/mtr_hip.h the host side (mtr_amd/host/) calls, and answers every read with the records the REFERENCE produced
 * for it (tests/golden G4 captures, converted to a table by tests/replay_table.py).  It lets the CPU suite drive the
 * real host pipeline — FASTA cutting and parsing, batching, the wire form, serialisation, the gather of the multi-GPU
 * launcher, chaining and printing — on machines without a GPU.  Nothing under mtr_amd/ refers to it; it is handed to the
 * host through the engine path (MTR_LIB / --engine-lib).  A read that is not in the table is an error (status 4).
 *
 * Its mtr_alignments is a plain CPU wrap-around DP (the recurrence of wrap_around_DP.c:77-185 restated), there only so
 * that the -a printing path can be exercised here; on the GPU box the device computes the paths.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mtr_hip.h"

typedef struct { int32_t len, n_records; uint64_t hash; int64_t wire_bytes; const uint8_t *wire; } entry;
struct mtr_ctx {
    char err[256];
    entry *tab; int64_t n_tab; uint8_t *raw;
    int32_t n; int32_t *lens; uint8_t **codes; const entry **hit;
    int32_t *counts; uint8_t *blob; int64_t blob_bytes;
    int ran, fail_after;
    int oom;                            /* MTR_REPLAY_OOM_CTX=n (tests): the n-th context created in the process (1 = the first) cannot launch: mtr_run_resident_async returns MTR_ERR_OOM */
};
struct mtr_file_state { int dummy; };

static uint64_t hash_codes(const uint8_t *c, int32_t n)
{   /* sum of (code + 1) * P^i mod 2^64 (tests/replay_table.py computes the same with numpy) */
    uint64_t h = 0, p = 1;
    for (int32_t i = 0; i < n; i++) { h += (uint64_t)(c[i] + 1) * p; p *= 1099511628211ULL; }
    return h;
}

int mtr_abi_version(void) { return MTR_ABI_VERSION; }
const char *mtr_last_error(const mtr_ctx *ctx) { return ctx ? ctx->err : "no context"; }

mtr_status mtr_create(int device, int manhattan, float min_match_ratio, mtr_ctx **out)
{
    (void)device; (void)manhattan; (void)min_match_ratio;
    const char *path = getenv("MTR_REPLAY_TABLE");
    if (!out || !path) return MTR_ERR_NO_DEVICE;
    FILE *fp = fopen(path, "rb");
    if (!fp) return MTR_ERR_NO_DEVICE;
    fseek(fp, 0, SEEK_END); long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    mtr_ctx *c = (mtr_ctx *)calloc(1, sizeof *c);
    c->raw = (uint8_t *)malloc((size_t)sz + 8);
    if (fread(c->raw, 1, (size_t)sz, fp) != (size_t)sz || sz < 16 || memcmp(c->raw, "MTRREPLY", 8) != 0) { fclose(fp); free(c->raw); free(c); return MTR_ERR_NO_DEVICE; }
    fclose(fp);
    memcpy(&c->n_tab, c->raw + 8, 8);
    c->tab = (entry *)calloc((size_t)c->n_tab + 1, sizeof(entry));
    const uint8_t *p = c->raw + 16;
    for (int64_t i = 0; i < c->n_tab; i++) {
        memcpy(&c->tab[i].len, p, 4); memcpy(&c->tab[i].n_records, p + 4, 4); memcpy(&c->tab[i].hash, p + 8, 8); memcpy(&c->tab[i].wire_bytes, p + 16, 8);
        c->tab[i].wire = p + 24; p += 24 + c->tab[i].wire_bytes;
    }
    c->fail_after = getenv("MTR_REPLAY_FAIL_AT") ? atoi(getenv("MTR_REPLAY_FAIL_AT")) : -1;
    { static int created; const int nth = __atomic_add_fetch(&created, 1, __ATOMIC_RELAXED); const char *e = getenv("MTR_REPLAY_OOM_CTX"); c->oom = e && atoi(e) == nth; }
    *out = c;
    return MTR_OK;
}

static void drop_batch(mtr_ctx *c)
{
    for (int i = 0; i < c->n; i++) free(c->codes[i]);
    free(c->lens); free(c->codes); free((void *)c->hit); free(c->counts); free(c->blob);
    c->lens = NULL; c->codes = NULL; c->hit = NULL; c->counts = NULL; c->blob = NULL; c->n = 0; c->ran = 0;
}

void mtr_destroy(mtr_ctx *c) { if (!c) return; drop_batch(c); free(c->tab); free(c->raw); free(c); }

static mtr_status take(mtr_ctx *c, int32_t n, const int32_t *lens)
{
    drop_batch(c);
    c->n = n; c->lens = (int32_t *)malloc(sizeof(int32_t) * (size_t)n); memcpy(c->lens, lens, sizeof(int32_t) * (size_t)n);
    c->codes = (uint8_t **)calloc((size_t)n, sizeof(uint8_t *)); c->hit = (const entry **)calloc((size_t)n, sizeof(entry *));
    return MTR_OK;
}
static mtr_status look_up(mtr_ctx *c)
{
    for (int i = 0; i < c->n; i++) {
        const uint64_t h = hash_codes(c->codes[i], c->lens[i]);
        c->hit[i] = NULL;
        for (int64_t t = 0; t < c->n_tab; t++) if (c->tab[t].len == c->lens[i] && c->tab[t].hash == h) { c->hit[i] = &c->tab[t]; break; }
        if (!c->hit[i]) { snprintf(c->err, sizeof c->err, "replay: read %d (length %d) is not in the table", i, c->lens[i]); return MTR_ERR_HIP; }
    }
    return MTR_OK;
}

mtr_status mtr_upload_batch_packed(mtr_ctx *c, const uint32_t *packed, int64_t n_words, const int64_t *woff, const int32_t *lens, int32_t n)
{
    if (!c || !packed || !woff || !lens || n <= 0) return MTR_ERR_BAD_ARG;
    take(c, n, lens);
    for (int i = 0; i < n; i++) {
        if (woff[i] + mtr_packed_words(lens[i]) > n_words) return MTR_ERR_BAD_ARG;
        const uint32_t *w = packed + woff[i];
        c->codes[i] = (uint8_t *)malloc((size_t)lens[i] + 2);
        for (int p = 0; p < lens[i]; p++) c->codes[i][p] = (uint8_t)((w[p >> 4] >> (30 - 2 * (p & 15))) & 3u);
        /* the image must be zero behind the read (isolated semantics) */
        for (int p = lens[i]; p < (lens[i] / 16 + 4) * 16; p++) if ((w[p >> 4] >> (30 - 2 * (p & 15))) & 3u) { snprintf(c->err, sizeof c->err, "replay: read %d is not zero-padded", i); return MTR_ERR_BAD_ARG; }
    }
    return look_up(c);
}
mtr_status mtr_upload_batch_in_file(mtr_ctx *c, mtr_file_state *fs, const uint8_t *bases, const int64_t *offsets, const int32_t *lens, int32_t n)
{
    if (!c || !fs || !bases || !offsets || !lens || n <= 0) return MTR_ERR_BAD_ARG;
    take(c, n, lens);
    for (int i = 0; i < n; i++) { c->codes[i] = (uint8_t *)malloc((size_t)lens[i] + 2); memcpy(c->codes[i], bases + offsets[i], (size_t)lens[i]); }
    return look_up(c);
}
mtr_status mtr_run_resident_async(mtr_ctx *c)
{
    if (!c || c->n <= 0) return MTR_ERR_BAD_ARG;
    if (c->oom) { snprintf(c->err, sizeof c->err, "replayed failure: scratch allocation failed"); return MTR_ERR_OOM; }
    c->ran = 1;
    return MTR_OK;
}

mtr_status mtr_wait(mtr_ctx *c)
{
    if (!c) return MTR_ERR_BAD_ARG;
    if (c->ran && c->fail_after >= 0 && c->fail_after < c->n) { snprintf(c->err, sizeof c->err, "You need to increse the value of WrapDPsize. (replayed failure)"); return MTR_ERR_DP_TOO_LARGE; }
    return MTR_OK;
}
mtr_status mtr_get_first_failed_read(const mtr_ctx *c, int32_t *out) { if (!c || !out) return MTR_ERR_BAD_ARG; *out = (c->fail_after >= 0 && c->fail_after < c->n) ? c->fail_after : -1; return MTR_OK; }

mtr_status mtr_fetch_results_packed(mtr_ctx *c, int32_t limit, const uint8_t **out_blob, int64_t *out_bytes, const int32_t **out_counts, int64_t *out_total)
{
    if (!c || !c->ran) return MTR_ERR_BAD_ARG;
    int n = c->n; if (limit >= 0 && limit < n) n = limit;
    free(c->counts); free(c->blob);
    c->counts = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    int64_t bytes = 0, total = 0;
    for (int i = 0; i < n; i++) { bytes += c->hit[i]->wire_bytes; c->counts[i] = c->hit[i]->n_records; total += c->counts[i]; }
    c->blob = (uint8_t *)malloc((size_t)bytes + 8);
    int64_t o = 0;
    for (int i = 0; i < n; i++) { memcpy(c->blob + o, c->hit[i]->wire, (size_t)c->hit[i]->wire_bytes); o += c->hit[i]->wire_bytes; }
    *out_blob = c->blob; *out_bytes = bytes; *out_counts = c->counts; if (out_total) *out_total = total;
    return MTR_OK;
}

mtr_status mtr_get_bases_after_read(const mtr_ctx *c, int32_t i, uint8_t out[2]) { (void)c; (void)i; out[0] = out[1] = 0; return MTR_OK; }
mtr_status mtr_get_kernel_times(const mtr_ctx *c, mtr_kernel_time *out, int32_t n) { (void)c; for (int i = 0; i < n; i++) { out[i].ms = 0; out[i].launches = 0; } return MTR_OK; }
mtr_status mtr_get_counters(const mtr_ctx *c, int64_t *out, int32_t n) { (void)c; for (int i = 0; i < n; i++) out[i] = 0; return MTR_OK; }
mtr_status mtr_file_state_create(mtr_file_state **out) { *out = (mtr_file_state *)calloc(1, sizeof(mtr_file_state)); return MTR_OK; }
void mtr_file_state_destroy(mtr_file_state *fs) { free(fs); }
mtr_status mtr_file_state_skip(mtr_file_state *fs, const uint8_t *b, const int64_t *o, const int32_t *l, int32_t n) { (void)fs; (void)b; (void)o; (void)l; (void)n; return MTR_OK; }

/* wrap-around alignment of org[rep_start-1+i], i = 1..rep_end-rep_start+1, against the unit: one traceback code per cell */
mtr_status mtr_alignments(mtr_ctx *c, int32_t n, const int32_t *read_idx, const mtr_record *records, uint8_t **out_ops, int64_t **out_off, int32_t **out_end)
{
    enum { T_STOP = 0, T_MATCH, T_MISMATCH, T_DEL, T_INS };
    int64_t *off = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    int32_t *ends = (int32_t *)calloc((size_t)n * 2 + 2, sizeof(int32_t));
    uint8_t *ops = NULL; size_t cap = 0, used = 0;
    for (int t = 0; t < n; t++) {
        const mtr_record *r = &records[t];
        const uint8_t *codes = c->codes[read_idx[t]]; const int L = c->lens[read_idx[t]];
        const int U = r->rep_period, G = r->match_gain, MM = r->mismatch_penalty, D = r->indel_penalty;
        const int rows = r->rep_end - r->rep_start + 1, base = r->rep_start - 1;
        int *unit = (int *)malloc(sizeof(int) * (size_t)(U + 1));
        for (int j = 1; j <= U; j++) { const char ch = r->unit[j - 1]; unit[j] = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3; }
        int *prev = (int *)calloc((size_t)U + 1, sizeof(int)), *cur = (int *)calloc((size_t)U + 1, sizeof(int));
        uint8_t *tb = (uint8_t *)malloc((size_t)rows * (size_t)U);
        int best = 0, bi = 0, bj = 0;
        for (int i = 1; i <= rows; i++) {
            const int p = base + i;
            const int x = (p >= 0 && p < L) ? codes[p] : 0;
            uint8_t *row = tb + (size_t)(i - 1) * (size_t)U;
            for (int j = 1; j <= U; j++) {
                int v, k;
                if (x == unit[j]) { v = prev[j - 1] + G; k = T_MATCH; }
                else {
                    const int sub = prev[j - 1] - MM, ins = prev[j] - D;
                    v = sub > ins ? sub : ins;
                    int del = -1;
                    if (j > 1) { del = cur[j - 1] - D; if (del > v) v = del; }
                    if (v <= 0) { v = 0; k = T_STOP; }
                    else if (v == sub) k = T_MISMATCH;
                    else if (j > 1 && v == del) k = T_DEL;
                    else k = T_INS;
                }
                cur[j] = v; row[j - 1] = (uint8_t)k;
                if (best < v) { best = v; bi = i; bj = j; }
            }
            cur[0] = cur[U];
            if (row[0] == T_INS && cur[1] == cur[0] - D) row[0] = T_DEL;
            int *sw = prev; prev = cur; cur = sw;
        }
        int i = bi, j = bj;
        ends[2 * t] = base + bi; ends[2 * t + 1] = bj;
        while (i > 0) {
            const int k = tb[(size_t)(i - 1) * (size_t)U + (size_t)(j - 1)];
            if (k == T_STOP) break;
            if (used + 1 > cap) { cap = cap ? cap * 2 : 4096; ops = (uint8_t *)realloc(ops, cap); }
            ops[used++] = (uint8_t)k;
            if (k == T_MATCH || k == T_MISMATCH) { i--; j--; } else if (k == T_DEL) j--; else i--;
            if (j == 0) j = U;
        }
        off[t + 1] = (int64_t)used;
        free(unit); free(prev); free(cur); free(tb);
    }
    if (!ops) ops = (uint8_t *)malloc(1);
    *out_ops = ops; *out_off = off; *out_end = ends;
    return MTR_OK;
}

/* ---- the multi-GPU gather of include/mtr_hip.h (ABI 5), replayed in host memory: same tickets, same lifetimes, no device.  MTR_REPLAY_DEVICES = how many
 * "GPUs" the stand-in reports (default 8); MTR_REPLAY_GATHER_FAIL=1 makes mtr_gather_create fail as it does where RCCL cannot be used, so that the host's
 * fall-back to mtr_fetch_results_packed is exercised as well. */
#include <pthread.h>
#define RG_SLOTS 1024
typedef struct { uint8_t *p; int64_t bytes; int busy; } rg_slot;
struct mtr_gather { int n; rg_slot *slots; pthread_mutex_t mu; uint8_t *host; size_t host_cap; char err[128]; long long exchanges, bytes; int usable; };

mtr_status mtr_device_count(int32_t *out) { if (!out) return MTR_ERR_BAD_ARG; const char *e = getenv("MTR_REPLAY_DEVICES"); *out = e ? atoi(e) : 8; return *out > 0 ? MTR_OK : MTR_ERR_NO_DEVICE; }
const char *mtr_gather_last_error(const mtr_gather *g) { return g ? g->err : "no gather"; }
mtr_status mtr_gather_create(int32_t n, const int32_t *devices, mtr_gather **out)
{   /* like the product: the object always comes into being; whether "RCCL" is usable shows in mtr_gather_wait_ready / mtr_gather_get_stats */
    if (!out || n <= 0 || !devices) return MTR_ERR_BAD_ARG;
    mtr_gather *g = (mtr_gather *)calloc(1, sizeof *g);
    *out = g;
    g->usable = 1;
    if (getenv("MTR_REPLAY_GATHER_FAIL")) { snprintf(g->err, sizeof g->err, "replayed failure: no RCCL here"); g->usable = 0; }
    for (int r = 0; r < n; r++) for (int q = 0; q < r; q++) if (devices[q] == devices[r]) { snprintf(g->err, sizeof g->err, "device %d is given to two ranks", devices[r]); g->usable = 0; }
    g->n = n; g->slots = (rg_slot *)calloc((size_t)n * RG_SLOTS, sizeof(rg_slot));
    pthread_mutex_init(&g->mu, NULL);
    return MTR_OK;
}
mtr_status mtr_gather_wait_ready(mtr_gather *g) { return !g ? MTR_ERR_BAD_ARG : g->usable ? MTR_OK : MTR_ERR_NO_DEVICE; }
mtr_status mtr_gather_get_stats(const mtr_gather *g, int64_t *out, int32_t n)
{
    if (!g || !out) return MTR_ERR_BAD_ARG;
    const int64_t v[6] = { g->usable ? g->exchanges : 0, g->usable ? 0 : g->exchanges, g->usable ? g->bytes : 0, g->usable ? 0 : g->bytes, g->usable ? 1 : 0, g->usable ? 1 : -1 };
    for (int i = 0; i < n && i < 6; i++) out[i] = v[i];
    return MTR_OK;
}
void mtr_gather_destroy(mtr_gather *g)
{
    if (!g) return;
    if (g->slots) { for (int i = 0; i < g->n * RG_SLOTS; i++) free(g->slots[i].p); free(g->slots); pthread_mutex_destroy(&g->mu); }
    free(g->host); free(g);
}
mtr_status mtr_gather_stage(mtr_gather *g, int32_t rank, mtr_ctx *c, int32_t *counts_host, int64_t *out_total, int64_t *out_bytes, int32_t *out_ticket)
{
    if (!g || !c || !counts_host || !out_total || !out_bytes || !out_ticket || rank < 0 || rank >= g->n) return MTR_ERR_BAD_ARG;
    { mtr_status w = mtr_wait(c); if (w != MTR_OK) return w; }
    const uint8_t *blob; int64_t bytes, total; const int32_t *counts;
    mtr_status st = mtr_fetch_results_packed(c, -1, &blob, &bytes, &counts, &total);
    if (st != MTR_OK) return st;
    memcpy(counts_host, counts, sizeof(int32_t) * (size_t)c->n);
    pthread_mutex_lock(&g->mu);
    int slot = -1;
    for (int i = 0; i < RG_SLOTS && slot < 0; i++) if (!g->slots[rank * RG_SLOTS + i].busy) slot = i;
    if (slot < 0) { pthread_mutex_unlock(&g->mu); return MTR_ERR_OVERFLOW; }
    rg_slot *s = &g->slots[rank * RG_SLOTS + slot];
    s->busy = 1; s->bytes = bytes; free(s->p); s->p = (uint8_t *)malloc((size_t)bytes + 8); memcpy(s->p, blob, (size_t)bytes);
    pthread_mutex_unlock(&g->mu);
    *out_total = total; *out_bytes = bytes; *out_ticket = rank * RG_SLOTS + slot;
    return MTR_OK;
}
mtr_status mtr_gather_exchange(mtr_gather *g, int32_t n, const int32_t *tickets, const uint8_t **out_ptrs, int64_t *out_bytes)
{
    if (!g || n < 0 || (n > 0 && (!tickets || !out_ptrs || !out_bytes))) return MTR_ERR_BAD_ARG;
    size_t total = 0;
    pthread_mutex_lock(&g->mu);
    for (int i = 0; i < n; i++) {
        if (tickets[i] < 0 || tickets[i] >= g->n * RG_SLOTS || !g->slots[tickets[i]].busy) { pthread_mutex_unlock(&g->mu); snprintf(g->err, sizeof g->err, "ticket %d names no staged table", tickets[i]); return MTR_ERR_BAD_ARG; }
        total += ((size_t)g->slots[tickets[i]].bytes + 255) & ~(size_t)255;
    }
    if (g->host_cap < total + 8) { free(g->host); g->host_cap = total + total / 4 + 4096; g->host = (uint8_t *)malloc(g->host_cap); }
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        rg_slot *s = &g->slots[tickets[i]];
        memcpy(g->host + off, s->p, (size_t)s->bytes);
        out_ptrs[i] = g->host + off; out_bytes[i] = s->bytes;
        off += ((size_t)s->bytes + 255) & ~(size_t)255;
        s->busy = 0;
        g->bytes += s->bytes;
    }
    g->exchanges++;
    pthread_mutex_unlock(&g->mu);
    return MTR_OK;
}
