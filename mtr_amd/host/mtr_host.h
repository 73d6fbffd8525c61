/* mtr_host.h — host side of the MI355X mTR: same command line, FASTA input and report/alignment output as reference
 * mTR (main.c, handle_one_file.c, chaining.cpp, pretty_print_alignment), with the per-read hot path delegated to
 * libmtr_hip.so through the C-ABI of include/mtr_hip.h.  Plain C, like the reference's host side.
 *
 * One pipeline serves the command line (mtr_amd/host/mTR, one GPU) and the multi-GPU launcher (python -m mtr_amd.run,
 * one process per GPU, libmtr_host.so through ctypes; torch.distributed only moves the bytes):
 *
 *   FASTA file(s), mmap'ed ── cut at record boundaries ──> chunks ── chunk c belongs to rank owner(c) ──┐
 *   parser threads (the reference reader's rules, 2-bit packing) ──> batches ──> two device contexts in turn
 *   (mtr_upload_batch_packed, mtr_run_resident_async; the batch before is fetched in wire form meanwhile)
 *   ──> results ── [multi-GPU: serialised, gathered to rank 0 over RCCL] ──> chaining + printing in input order
 *   (a pool formats, one writer emits).
 */
#ifndef MTR_HOST_H
#define MTR_HOST_H
#include <setjmp.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mtr_hip.h"

/* ---- allocation (alloc.c): never returns NULL; a worker thread that cannot allocate jumps to its entry function ------------- */
enum { MTRH_THREAD_MAIN = 0, MTRH_THREAD_PARSER, MTRH_THREAD_DEVICE, MTRH_THREAD_PRINTER };
extern __thread jmp_buf *mtrh_oom_target;      /* where mtrh_oom() jumps to in this thread (NULL: print and exit, like the reference) */
extern __thread int mtrh_thread_kind;
void  mtrh_oom(size_t bytes) __attribute__((noreturn));
void *mtrh_xmalloc(size_t n);
void *mtrh_xcalloc(size_t n, size_t m);
void *mtrh_xrealloc(void *p, size_t n);
char *mtrh_xstrdup(const char *s);
#ifndef MTRH_NO_ALLOC_WRAP
#define malloc(n) mtrh_xmalloc(n)
#define calloc(n, m) mtrh_xcalloc(n, m)
#define realloc(p, n) mtrh_xrealloc(p, n)
#define strdup(s) mtrh_xstrdup(s)
#endif

#define MTRH_BLK 4096                 /* fgets chunk of the reference reader (mTR.h:57) */
#define MTRH_OVERLAP 10               /* MAX_LEN_overlapping (mTR.h:39) */
#define MTRH_ALIGN_WIDTH 50           /* ALIGNMENT_WIDTH_PRINTING (mTR.h:38) */

/* ---- the engine: include/mtr_hip.h as a table of entry points (types taken from the header) ---------------------- */
typedef struct mtrh_engine {
    void *dl;
    __typeof__(mtr_create) *create;
    __typeof__(mtr_destroy) *destroy;
    __typeof__(mtr_last_error) *last_error;
    __typeof__(mtr_upload_batch_packed) *upload_packed;
    __typeof__(mtr_upload_batch_in_file) *upload_in_file;
    __typeof__(mtr_run_resident_async) *run_async;
    __typeof__(mtr_wait) *wait;
    __typeof__(mtr_fetch_results_packed) *fetch_packed;
    __typeof__(mtr_get_first_failed_read) *first_failed;
    __typeof__(mtr_alignments) *alignments;
    __typeof__(mtr_get_bases_after_read) *bases_after;
    __typeof__(mtr_get_kernel_times) *kernel_times;
    __typeof__(mtr_get_counters) *counters;
    __typeof__(mtr_file_state_create) *fs_create;
    __typeof__(mtr_file_state_destroy) *fs_destroy;
    __typeof__(mtr_file_state_skip) *fs_skip;
    /* several GPUs in one process (ABI 5) */
    __typeof__(mtr_device_count) *device_count;
    __typeof__(mtr_gather_create) *gather_create;
    __typeof__(mtr_gather_destroy) *gather_destroy;
    __typeof__(mtr_gather_last_error) *gather_last_error;
    __typeof__(mtr_gather_stage) *gather_stage;
    __typeof__(mtr_gather_exchange) *gather_exchange;
    __typeof__(mtr_gather_wait_ready) *gather_wait_ready;
    __typeof__(mtr_gather_get_stats) *gather_get_stats;
    char path[4096];                    /* the library that was bound, resolved (reported by mTR -c) */
} mtrh_engine;
/* dlopen a library that implements include/mtr_hip.h; lib_path NULL = $MTR_LIB, else libmtr_hip.so next to this code */
int  mtrh_engine_load(mtrh_engine *e, const char *lib_path, char *err, size_t errlen);
void mtrh_engine_unload(mtrh_engine *e);

/* ---- a record of the wire form (include/mtr_hip.h), viewed in place ------------------------------------------------ */
typedef struct { const int32_t *h; const char *unit; const int32_t *score; } mtrh_rec;   /* h[0..13] = rep_start .. reserved */
enum { MTRH_REP_START = 0, MTRH_REP_END, MTRH_REPEAT_LEN, MTRH_PERIOD, MTRH_COPIES, MTRH_MATCHES, MTRH_MISMATCHES,
       MTRH_INSERTIONS, MTRH_DELETIONS, MTRH_KMER, MTRH_GAIN, MTRH_MISMATCH_PEN, MTRH_INDEL_PEN };
/* the record at *p; advances *p; returns 0 when the record does not fit before end */
int  mtrh_rec_next(const uint8_t **p, const uint8_t *end, mtrh_rec *out);

/* ---- FASTA: the reference reader's rules (handle_one_file.c:169-269) on a memory-mapped file ----------------------- */
typedef struct { const char *path; const char *map; size_t size; int fd; } mtrh_file;
int  mtrh_file_open(mtrh_file *f, const char *path);           /* 0 = ok; prints the reference's message otherwise */
void mtrh_file_close(mtrh_file *f);
/* cuts the file into at most n_target pieces of about equal size: offsets[0..n] with offsets[0] = 0, offsets[n] = size;
 * every inner offset is the '>' of a header at a line start */
size_t *mtrh_plan_chunks(const mtrh_file *f, int n_target, int *n_chunks);

enum { MTRH_END_NONE = 0,      /* more input may follow */
       MTRH_END_EMPTY,         /* an empty record: the reference stops reading here (handle_one_file.c:283) */
       MTRH_END_BADCHAR,       /* "Invalid character" (fatal in the reference, after the reads before it were reported) */
       MTRH_END_TOOLONG };     /* a read reached MAX_INPUT_LENGTH (fatal likewise) */

typedef struct mtrh_batch {
    int32_t  n;                            /* reads */
    int32_t *lens;
    int64_t *offs;    uint8_t  *codes;     /* base codes 0..3, one byte per base (kept for -a rows and -B) */
    int64_t *woff;    uint32_t *packed; int64_t n_words;   /* the device image (mtr_upload_batch_packed) */
    const char **ids; int32_t *id_lens;    /* header after '>' (not NUL-terminated; points into the file map or an owned copy) */
    char    *id_store;                     /* owned copy of the IDs when the batch was deserialised */
    int      end;     char bad_char;       /* MTRH_END_*: what followed the last read of this batch */
    const char *end_id; int32_t end_id_len;/* MTRH_END_TOOLONG: the ID the reference prints */
    struct mtrh_batch *next;               /* a chunk may yield several batches */
} mtrh_batch;
/* parse [begin, end) of the file (a piece of mtrh_plan_chunks) into batches of at most max_reads reads / max_bases bases */
mtrh_batch *mtrh_parse_chunk(const mtrh_file *f, size_t begin, size_t end, int max_reads, int64_t max_bases);
/* the same without the code array (codes = NULL, offs = running base counts): the bases go straight into the 2-bit image, 16 characters at a time (fasta.c) */
mtrh_batch *mtrh_parse_chunk_packed(const mtrh_file *f, size_t begin, size_t end, int max_reads, int64_t max_bases);
void mtrh_batch_free(mtrh_batch *b);       /* the whole list */

/* ---- chaining.cpp:243-363 with alignments taken in insertion order ------------------------------------------------ */
/* fills chain[] (capacity n) with the indices of the records of the maximum-score chain in print order; returns its length */
int  mtrh_chain(const mtrh_rec *recs, int n, int *chain);

/* ---- results of one batch: everything chaining + printing need ------------------------------------------------------ */
typedef struct mtrh_result {
    int32_t  chunk, file_idx, last_of_chunk;  /* chunk of the plan (output order), its file, 1 = the chunk's last result */
    mtrh_batch *batch;                     /* ids, lens, (codes) */
    int32_t  n_report;                     /* reads to report: batch->n, fewer after a device-side failure */
    int32_t *counts;  uint8_t *wire; int64_t wire_bytes;     /* records per read, wire form */
    int32_t  ticket;                       /* >= 0: the wire form is still on the GPU, staged for the gather (mtr_gather_stage); wire is filled by the exchange */
    int32_t  fetched_by_run;               /* a gather exists, but this table came to the host with the run's own fetch (every staging slot of its GPU was taken) */
    /* -a only: the chains (made where the batch was resident) and the alignment paths of their records */
    int32_t  with_alignments;
    int32_t *chain_len; int32_t *chain_idx; int64_t n_chain;  /* per read: length; concatenated record indices (within the read) */
    uint8_t *ops; int64_t *ops_off; int32_t *ends; uint8_t *after;   /* mtr_alignments' results per chained record; 2 bytes per read */
    int      fatal;  char *fatal_msg;      /* a device-side error: printed on stderr after n_report reads, exit status 1 */
    double   t_kernel_ms; int64_t queries;
    double   t_phase_ms[MTR_N_KERNEL_TIMES];  /* device time by phase (mtr_get_kernel_times ids); not serialised */
} mtrh_result;
void   mtrh_result_free(mtrh_result *r);
/* one self-contained byte string (malloc'ed) and back: what the multi-GPU launcher gathers to rank 0 */
uint8_t *mtrh_result_serialize(const mtrh_result *r, size_t *out_bytes);
mtrh_result *mtrh_result_deserialize(const uint8_t *blob, size_t bytes, size_t *used);

/* ---- the run: parse + device pipeline of the chunks one rank owns ---------------------------------------------------- */
typedef struct mtrh_opts {
    int   print_alignment, manhattan, file_order, device;
    float min_match_ratio;
    int   rank, world;                     /* chunk c of the plan belongs to rank owner[c] (see mtrh_run_plan) */
    int   lpt;                             /* 1 = chunks go to ranks longest-first (several files); 0 = round-robin */
    size_t chunk_bytes;                    /* 0 = default */
    int   parse_threads, print_threads;    /* 0 = default */
    const char *engine_lib;                /* NULL = default (mtrh_engine_load) */
    mtr_gather *gather;                    /* several GPUs in one process: batches leave their tables staged on the GPU for the RCCL gather (NULL: fetched to the host) */
    int   contexts;                        /* device batches in flight per GPU: 0 = 2 (3 on a long job), or 6 where the reads are long (mean >= 8 kb: launches bound by their longest item) */
} mtrh_opts;
typedef struct mtrh_run mtrh_run;
/* opens the files, plans the chunks, starts the parser threads and the device thread; NULL + message on stderr on failure */
mtrh_run *mtrh_run_start(const mtrh_opts *o, const char *const *paths, int n_paths);
int   mtrh_run_rank(const mtrh_run *r);
int   mtrh_run_n_chunks(const mtrh_run *r);
int   mtrh_run_n_rounds(const mtrh_run *r);                    /* a round = the chunks gathered together */
int   mtrh_run_owner(const mtrh_run *r, int chunk);
int   mtrh_run_round_of(const mtrh_run *r, int chunk);
/* next finished result of this rank in output order (blocks); NULL at the end */
mtrh_result *mtrh_run_next(mtrh_run *r);
void  mtrh_run_timing(const mtrh_run *r, double *t_parse_wait, double *t_submit, double *t_fetch, double *t_kernel, long long *queries);
const char *mtrh_run_engine_path(const mtrh_run *r);           /* the engine library the run bound (resolved path) */
/* seconds spent creating the device contexts, and device time by phase of the chain (ids of mtr_get_kernel_times) summed over the batches: what mTR -c prints */
void  mtrh_run_phase_times(const mtrh_run *r, double *t_create, double *t_phase, int n_phase);
void  mtrh_stamp(const char *what);           /* development aid: with MTR_HOST_TIMING set, a line on stderr with the time since the first stamp */
void  mtrh_run_stop(mtrh_run *r);
/* all results this rank produces for `round`, serialised one after the other (malloc'ed; free() it) */
uint8_t *mtrh_run_round_blob(mtrh_run *r, int round, size_t *bytes);

/* ---- several GPUs in ONE process: mTR -g N (multi.c) ------------------------------------------------------------------------
 * N runs, one per GPU (its own parser threads, device thread and contexts; chunk c of the plan belongs to GPU c % N, or longest-first
 * over the GPUs for several files), merged in output order by the calling thread.  The path's one exchange: per round the staged
 * wire-form tables of every GPU are gathered to the first GPU over RCCL (mtr_gather_exchange) and chained + printed from there.
 * RCCL comes up in the background (2 s on an MI355X box: more than a 100 000-read job takes) and is used from the round that finds it up; until
 * then - and where it cannot be used at all - the staged tables are copied straight to the host by the same call.  No gather object at all (every
 * GPU's device thread fetches its tables itself): MTR_GATHER=host, -a (the chains are made where the batch is resident), one GPU (nothing to
 * gather), more ranks than GPUs (a rehearsal on a shared card).  MTR_GATHER=rccl waits for RCCL before the first batch.  Same output always. */
typedef struct mtrh_multi mtrh_multi;
typedef struct mtrh_printer mtrh_printer;
mtrh_multi *mtrh_multi_start(const mtrh_opts *o, int n_gpus, const char *const *paths, int n_paths);
/* every result to the printer in output order (blocks until the runs are through); 0, or -1 after an exchange failed */
int   mtrh_multi_drain(mtrh_multi *m, mtrh_printer *p);
/* one line for mTR -c: how the tables reached the printer ("<N> GPUs\tgather rccl, ..." / "... gather host, ...") */
void  mtrh_multi_gather_line(const mtrh_multi *m, char *buf, size_t n);
mtrh_run *mtrh_multi_run(const mtrh_multi *m, int gpu);
int   mtrh_multi_n(const mtrh_multi *m);
/* returns 1 when RCCL was still coming up in the background (a job shorter than the library's start-up): the caller leaves through _exit */
int   mtrh_multi_stop(mtrh_multi *m);

/* ---- printing: chaining.cpp:125-171 (+ the alignment block of wrap_around_DP.c:187-212 with -a) --------------------- */
typedef struct mtrh_printer mtrh_printer;
mtrh_printer *mtrh_printer_start(FILE *out, int threads);
mtrh_printer *mtrh_printer_start_stdout(int threads);          /* for callers without a FILE* (ctypes) */
mtrh_printer *mtrh_printer_start_fd(int fd, int threads);       /* the report on a descriptor of the caller's (fdopen; flushed by mtrh_printer_finish) */
/* takes ownership of r; results must arrive in output order.  Once a file's input ended (an empty record, a fatal
 * character, a device-side error: the message has been printed) the later results of that file are dropped. */
void  mtrh_printer_push(mtrh_printer *p, mtrh_result *r);
/* waits for everything pushed; returns the exit status (0, or 1 after a fatal input / device error) */
int   mtrh_printer_finish(mtrh_printer *p, double *t_chain);
/* rank 0: the blobs of every rank for one round -> results in output order -> the printer; -1 = a malformed blob */
int   mtrh_print_round(mtrh_printer *p, const uint8_t *const *blobs, const size_t *sizes, int n_blobs);
/* pool threads for a printer by the cores of the machine (half of them, 2 .. 16) */
int   mtrh_printer_default_threads(void);
/* "%f" of (float)matches / repeat_len without printf (print.c); out holds 64 bytes; returns the length */
int   mtrh_format_ratio(int matches, int repeat_len, char *out);
/* formats one result into a malloc'ed buffer (used by the printer's pool; exposed for tests) */
char *mtrh_format_result(const mtrh_result *r, int first_read, int last_read, size_t *out_len);
#endif
