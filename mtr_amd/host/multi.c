/* multi.c — mTR -g N: the N GPUs of one node in ONE process (host code stays C; replaces the per-read loop of handle_one_file.c:271-293 across
 * GPUs, and the Python launcher for everything but jobs torchrun starts).
 *
 * N runs (pipeline.c), one per GPU: every run maps the file(s), makes the same plan and takes the chunks its rank owns (one file: chunk c ->
 * GPU c % N, round c / N; several files: longest first to the least loaded GPU, one round), with its own parser threads, device thread and
 * contexts.  Reads shard (isolated semantics); the path has ONE exchange, and this file runs it: per round, the wire-form record tables every GPU
 * left staged on its own device (mtr_gather_stage, in the GPU's device thread) are gathered to the first GPU over RCCL / xGMI and copied to the host
 * in one piece (mtr_gather_exchange: ncclCommInitAll at start, one ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd per round); then the calling
 * thread hands the round's results to the printer in output order, where they are chained and printed as on one GPU (chain.c, print.c).  While
 * round t is gathered and printed the GPUs run round t + 1.
 *
 * RCCL comes up in the background (loading librccl.so + ncclCommInitAll: 2.2 s measured on an MI355X box, three times the whole 100 000-read job):
 * a round that finds it up goes over RCCL, an earlier one copies the staged tables straight to the host inside the same call (mtr_hip.h), and a job
 * that ends before RCCL is up leaves without waiting for it.  No gather at all - every GPU's device thread fetches its own tables
 * (mtr_fetch_results_packed) - with MTR_GATHER=host, -a (the chains are made where the batch is resident, which needs the records on the host there),
 * one GPU (nothing to gather) and more ranks than GPUs (a rehearsal of N runs on one card).  MTR_GATHER=rccl waits for RCCL before the first batch and
 * fails if it cannot be had.  The output is the same byte for byte in every case.
 */
#define _GNU_SOURCE
#include "mtr_host.h"
#include <pthread.h>
#include <unistd.h>

struct mtrh_multi {
    int n; mtrh_run **runs; mtrh_engine eng; mtr_gather *gather;
    char note[300];
    long long bytes;
};

void mtrh_multi_gather_line(const mtrh_multi *m, char *buf, size_t n)
{
    if (!m->gather) { snprintf(buf, n, "%d GPUs\tgather host, 0 exchange(s), %lld bytes of record tables%s%s%s", m->n, m->bytes, m->note[0] ? " (" : "", m->note, m->note[0] ? ")" : ""); return; }
    int64_t st[6] = { 0, 0, 0, 0, 0, 0 };
    (void)m->eng.gather_get_stats(m->gather, st, 6);
    char up[96] = "";
    if (st[5] == 1) snprintf(up, sizeof up, "RCCL up after %lld ms", (long long)st[4]);
    else snprintf(up, sizeof up, st[5] == 0 ? "RCCL still coming up when the job ended" : "RCCL not usable");
    snprintf(buf, n, "%d GPUs\tgather rccl, %lld exchange(s) over RCCL + %lld straight to the host, %lld bytes of record tables (%s%s%s)", m->n, (long long)st[0], (long long)st[1],
             (long long)(st[2] + st[3]), up, m->note[0] ? "; " : "", m->note);
}
mtrh_run *mtrh_multi_run(const mtrh_multi *m, int gpu) { return gpu >= 0 && gpu < m->n ? m->runs[gpu] : NULL; }
int mtrh_multi_n(const mtrh_multi *m) { return m->n; }

mtrh_multi *mtrh_multi_start(const mtrh_opts *o, int n_gpus, const char *const *paths, int n_paths)
{
    if (n_gpus < 1 || n_gpus > 64) { fprintf(stderr, "fatal error: -g takes 1 .. 64 GPUs\n"); return NULL; }
    mtrh_multi *m = (mtrh_multi *)calloc(1, sizeof *m);
    char err[512];
    if (mtrh_engine_load(&m->eng, o->engine_lib, err, sizeof err) != 0) { fprintf(stderr, "fatal error: %s\n", err); free(m); return NULL; }
    int32_t ndev = 0;
    if (m->eng.device_count(&ndev) != MTR_OK || ndev <= 0) {
        fprintf(stderr, "fatal error: no usable HIP device; this build has no CPU path\n");
        free(m); return NULL;
    }
    /* The files are looked at BEFORE the gather exists: mtr_gather_create starts a thread that loads librccl.so and runs ncclCommInitAll (~2 s), and a caller
     * that leaves through exit() after a plain user error (`mTR -g 2 typo.fa`) would tear the runtime down under that thread.  A file that cannot
     * be opened is reported here, with the reader's own message (fasta.c, handle_one_file.c:192-196), while nothing runs yet. */
    for (int f = 0; f < n_paths; f++) {
        mtrh_file probe;
        if (mtrh_file_open(&probe, paths[f]) != 0) { free(m); return NULL; }
        mtrh_file_close(&probe);
    }
    m->n = n_gpus;
    int32_t *dev = (int32_t *)calloc((size_t)n_gpus, sizeof(int32_t));
    for (int r = 0; r < n_gpus; r++) dev[r] = (o->device + r) % ndev;             /* -d k: the first GPU; more ranks than GPUs share (a rehearsal) */
    /* the gather: RCCL (in the background) unless something rules it out */
    const char *want = getenv("MTR_GATHER");
    const int want_rccl = want && strcmp(want, "rccl") == 0;
    if (want && strcmp(want, "host") == 0) snprintf(m->note, sizeof m->note, "MTR_GATHER=host");
    else if (o->print_alignment) snprintf(m->note, sizeof m->note, "-a: the chains are made where the batch is resident, the records go to the host there");
    else if (n_gpus > ndev) snprintf(m->note, sizeof m->note, "%d ranks on %d GPU(s): RCCL takes a device once per communicator", n_gpus, (int)ndev);
    else if (n_gpus == 1 && !want_rccl) snprintf(m->note, sizeof m->note, "one GPU: nothing to gather");
    else {
        mtr_status st = m->eng.gather_create(n_gpus, dev, &m->gather);
        if (st == MTR_OK && want_rccl) st = m->eng.gather_wait_ready(m->gather);          /* asked for by name: every round on RCCL, or not at all */
        if (st != MTR_OK) {
            snprintf(m->note, sizeof m->note, "%s", m->gather ? m->eng.gather_last_error(m->gather) : "the gather could not be created");
            if (m->gather) m->eng.gather_destroy(m->gather);
            m->gather = NULL;
            if (want_rccl) { fprintf(stderr, "fatal error: MTR_GATHER=rccl, but %s\n", m->note); free(dev); free(m); return NULL; }
        }
    }
    if (want_rccl && !m->gather) { fprintf(stderr, "fatal error: MTR_GATHER=rccl, but %s\n", m->note); free(dev); free(m); return NULL; }
    m->runs = (mtrh_run **)calloc((size_t)n_gpus, sizeof(mtrh_run *));
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    for (int r = 0; r < n_gpus; r++) {
        mtrh_opts q = *o;
        q.rank = r; q.world = n_gpus; q.device = dev[r]; q.gather = m->gather;
        q.lpt = n_paths > 1;
        if (q.parse_threads <= 0) { long per = (ncpu > 2 ? ncpu - 2 : 1) / n_gpus; q.parse_threads = per >= 4 ? 4 : (per >= 1 ? (int)per : 1); }
        m->runs[r] = mtrh_run_start(&q, paths, n_paths);
        if (!m->runs[r]) {
            for (int t = 0; t < r; t++) mtrh_run_stop(m->runs[t]);
            /* (a run that cannot start although its files opened a moment ago: the caller will exit(), so RCCL's start-up thread is waited for first) */
            if (m->gather) { (void)m->eng.gather_wait_ready(m->gather); m->eng.gather_destroy(m->gather); }
            free(m->runs); free(dev); free(m);
            return NULL;
        }
    }
    free(dev);
    return m;
}

/* the results of one round in output order, and the round's exchange */
int mtrh_multi_drain(mtrh_multi *m, mtrh_printer *p)
{
    mtrh_run *r0 = m->runs[0];
    const int n_chunks = mtrh_run_n_chunks(r0), n_rounds = mtrh_run_n_rounds(r0);
    int cap = 64, failed = 0;
    mtrh_result **res = (mtrh_result **)malloc(sizeof(mtrh_result *) * (size_t)cap);
    int32_t *tickets = NULL; const uint8_t **ptrs = NULL; int64_t *sizes = NULL; int *owner_of = NULL; int tcap = 0;
    for (int t = 0; t < n_rounds; t++) {
        int n = 0, nt = 0;
        for (int c = 0; c < n_chunks; c++) {
            if (mtrh_run_round_of(r0, c) != t) continue;
            mtrh_run *run = m->runs[mtrh_run_owner(r0, c)];
            for (;;) {                                         /* a chunk's results, up to its last one */
                mtrh_result *x = mtrh_run_next(run);
                if (!x) break;                                 /* the run ended early (stopped): nothing more from it */
                if (n == cap) { cap *= 2; res = (mtrh_result **)realloc(res, sizeof(mtrh_result *) * (size_t)cap); }
                res[n++] = x;
                if (x->ticket >= 0) nt++;
                if (x->last_of_chunk) break;
            }
        }
        if (nt > 0 && !failed) {
            if (nt > tcap) {
                tcap = nt + 16;
                tickets = (int32_t *)realloc(tickets, sizeof(int32_t) * (size_t)tcap); ptrs = (const uint8_t **)realloc((void *)ptrs, sizeof(uint8_t *) * (size_t)tcap);
                sizes = (int64_t *)realloc(sizes, sizeof(int64_t) * (size_t)tcap); owner_of = (int *)realloc(owner_of, sizeof(int) * (size_t)tcap);
            }
            int k = 0;
            for (int i = 0; i < n; i++) if (res[i]->ticket >= 0) { tickets[k] = res[i]->ticket; owner_of[k] = i; k++; }
            const mtr_status st = m->eng.gather_exchange(m->gather, k, tickets, ptrs, sizes);
            if (st != MTR_OK) {
                fprintf(stderr, "fatal error: the gather of the record tables failed: %s\n", m->eng.gather_last_error(m->gather));
                failed = 1;
            } else {
                for (int j = 0; j < k; j++) {
                    mtrh_result *x = res[owner_of[j]];
                    x->wire = (uint8_t *)malloc((size_t)sizes[j] + 8); x->wire_bytes = sizes[j];
                    if (sizes[j]) memcpy(x->wire, ptrs[j], (size_t)sizes[j]);
                    x->ticket = -1;
                    m->bytes += sizes[j];
                }
            }
        }
        for (int i = 0; i < n; i++) {
            if (failed && res[i]->ticket >= 0) { mtrh_result_free(res[i]); continue; }      /* its table never arrived */
            if (!m->gather || res[i]->fetched_by_run) m->bytes += res[i]->wire_bytes;
            mtrh_printer_push(p, res[i]);
        }
        if (failed) break;
    }
    free(res); free(tickets); free((void *)ptrs); free(sizes); free(owner_of);
    return failed ? -1 : 0;
}

int mtrh_multi_stop(mtrh_multi *m)
{
    if (!m) return 0;
    int pending = 0;
    for (int r = 0; r < m->n; r++) mtrh_run_stop(m->runs[r]);
    if (m->gather) {
        int64_t st[6] = { 0, 0, 0, 0, 0, 0 };
        if (m->eng.gather_get_stats(m->gather, st, 6) == MTR_OK && st[5] == 0) pending = 1;      /* RCCL still coming up: nobody waits for it */
        m->eng.gather_destroy(m->gather);
    }
    free(m->runs);
    free(m);
    return pending;
}
