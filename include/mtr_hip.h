/*
 * mtr_hip.h — C-ABI of libmtr_hip.so, the MI355X (gfx950) implementation of reference mTR's
 * per-read hot path.
 *
 * The reference has no FFI; the seam this library stands behind is the function boundary of its
 * per-read layer:
 *
 *   upper edge  void handle_one_read(char *readID, int inputLen, int read_cnt, int print_alignment)
 *               (reference mTR.h:127, called from handle_one_file.c:286), whose inputs also arrive
 *               through the globals orgInputString (mTR.h:65), Manhattan_Distance (mTR.h:61, -p sets
 *               it to 0) and min_match_ratio (mTR.h:62, -m);
 *   lower edge  insert_an_alignment_into_set(...17 arguments...) (mTR.h:151-168), called once per
 *               qualified repeat in candidate order (handle_one_read.c:156-176, :239-243).
 *
 * mtr_process_batch() is the batch form of that edge: it takes N reads as integer base codes
 * (what handle_one_file.c:284-285 copies into orgInputString) and returns, per read and in the
 * reference's insertion order, exactly the 17 arguments of insert_an_alignment_into_set (readID and
 * inputLen are the caller's own).  Chaining + printing (chaining.cpp) stay on the host side of the
 * boundary (mtr_amd/host/).
 *
 * Semantics = the reference run one read per process ("isolated semantics"): the
 * results do not depend on which other reads share the batch.
 *
 * Conventions: every entry point returns an mtr_status; nothing calls exit(); the context owns all
 * device memory; one context per GPU / host thread; not re-entrant on one context.
 * There is NO CPU fallback: without a HIP device every entry point fails with MTR_ERR_NO_DEVICE.
 */
#ifndef MTR_HIP_H
#define MTR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTR_MAX_PERIOD 500          /* reference mTR.h:35 MAX_PERIOD */
#define MTR_MAX_INPUT_LENGTH 1000000 /* reference mTR.h:31: the reader's limit (a read that reaches it is fatal) */
/* Longest read the hot path takes: the reference's buffers hold L + 2r entries with r = L/10 random flank bases
 * (handle_one_read.c:194-204), so beyond L + 2r = MAX_INPUT_LENGTH it writes out of bounds; uploads refuse such reads. */
#define MTR_MAX_READ_LENGTH 833333
#define MTR_ABI_VERSION 5

typedef enum {
    MTR_OK = 0,
    MTR_ERR_NO_DEVICE = 1,      /* no HIP device / HIP runtime error at create */
    MTR_ERR_BAD_ARG = 2,        /* null pointer, bad length (<=0 or > MTR_MAX_READ_LENGTH), bad code (>3) */
    MTR_ERR_OOM = 3,            /* host or device allocation failed */
    MTR_ERR_HIP = 4,            /* a HIP call failed; mtr_last_error() has the text */
    MTR_ERR_OVERFLOW = 5,       /* a read produced more candidate ranges than L/2+64, or a caller-owned destination is too small */
    MTR_ERR_DP_TOO_LARGE = 6    /* a DP exceeded the reference's WrapDPsize (mTR.h:51): the reference exits */
} mtr_status;

/* One qualified repeat = arguments 3..17 of insert_an_alignment_into_set (mTR.h:151-168). */
typedef struct mtr_record {
    int32_t rep_start;          /* 0-origin, inclusive */
    int32_t rep_end;            /* 0-origin, inclusive */
    int32_t repeat_len;
    int32_t rep_period;
    int32_t num_freq_unit;
    int32_t num_matches;
    int32_t num_mismatches;
    int32_t num_insertions;
    int32_t num_deletions;
    int32_t kmer;
    int32_t match_gain;
    int32_t mismatch_penalty;
    int32_t indel_penalty;
    int32_t reserved;
    char    unit[MTR_MAX_PERIOD + 4];        /* "string", NUL-terminated ACGT */
    int32_t unit_score[MTR_MAX_PERIOD];      /* "string_score", first rep_period entries valid */
} mtr_record;

typedef struct mtr_ctx mtr_ctx;

/* device: HIP device ordinal.  manhattan: 1 = Manhattan DI (default), 0 = Pearson (-p).
 * min_match_ratio: the -m value (reference default 0.6, MIN_MATCH_RATIO mTR.h:32). */
mtr_status mtr_create(int device, int manhattan, float min_match_ratio, mtr_ctx **out);
void       mtr_destroy(mtr_ctx *ctx);
const char *mtr_last_error(const mtr_ctx *ctx);
int        mtr_abi_version(void);

/* Replaces the per-read loop body of handle_one_file.c:281-287 for n_reads reads at once.
 *   bases    concatenated base codes, one byte per base, 0..3 = A C G T (handle_one_file.c:169-188)
 *   offsets  n_reads start offsets into bases
 *   lens     n_reads lengths (1..MTR_MAX_READ_LENGTH)
 * On success *out_records is a malloc'ed array of all records, read after read, each read's records
 * in insertion order; (*out_counts)[i] is the number of records of read i.  Free both with
 * mtr_free_results(). */
mtr_status mtr_process_batch(mtr_ctx *ctx, const uint8_t *bases, const int64_t *offsets, const int32_t *lens,
                             int32_t n_reads, mtr_record **out_records, int32_t **out_counts, int64_t *out_total);
void       mtr_free_results(mtr_record *records, int32_t *counts);

/* The same path split so that a caller (bench.py) can keep inputs resident in HBM and time only the
 * device work: upload packs the reads to 2 bit/base and copies them to the device; run launches the
 * kernels on the context's stream and returns after they finish; fetch copies the records back. */
mtr_status mtr_upload_batch(mtr_ctx *ctx, const uint8_t *bases, const int64_t *offsets, const int32_t *lens, int32_t n_reads);
mtr_status mtr_run_resident(mtr_ctx *ctx);
/* The run split in two: _async enqueues the kernels on the context's stream and returns; mtr_wait blocks until
 * they finish and collects status, kernel times and counters.  Two contexts on one GPU can so overlap the tail of
 * one batch with the start of the next (bench.py pipelines its steps this way). */
mtr_status mtr_run_resident_async(mtr_ctx *ctx);
mtr_status mtr_wait(mtr_ctx *ctx);
mtr_status mtr_fetch_results(mtr_ctx *ctx, mtr_record **out_records, int32_t **out_counts, int64_t *out_total);

/* A failed run (mtr_wait returned an error) is remembered: fetch / export / alignments of that batch return the same
 * status instead of partial records.  After MTR_ERR_DP_TOO_LARGE the reads BEFORE the failing one (input order) are
 * still valid — the reference has printed them when it exits (wrap_around_DP.c:96-99) — and can be fetched with
 * mtr_fetch_results_packed(); *out_first_failed is the index of the first read whose DP exceeded WrapDPsize, or -1. */
mtr_status mtr_get_first_failed_read(const mtr_ctx *ctx, int32_t *out_first_failed);

/* ---- the host's own packing --------------------------------------------------------------------------------------
 * mtr_upload_batch packs the reads to the device layout on the calling thread.  A host that parses FASTA on several
 * threads packs there instead and hands over the finished image:
 *   read i occupies mtr_packed_words(lens[i]) = lens[i]/16 + 4 consecutive 32-bit words starting at word woff[i];
 *   base p of the read sits in word p>>4 at bits 31-2(p&15) .. 30-2(p&15) (first base in the top bits); every other
 *   bit of the read's words is 0 (org[L], org[L+1] read as 'A': isolated semantics).
 * mtr_pack_read writes one read's words (inline: a host needs no library for it); MTR_ERR_BAD_ARG for a code > 3. */
static inline int64_t mtr_packed_words(int32_t len) { return (int64_t)(len / 16) + 4; }
static inline mtr_status mtr_pack_read(const uint8_t *codes, int32_t len, uint32_t *dst_words)
{
    if (!codes || !dst_words || len <= 0) return MTR_ERR_BAD_ARG;
    const int32_t full = len >> 4;
    uint32_t seen = 0;
    for (int32_t q = 0; q < full; q++) {
        const uint8_t *c = codes + ((int64_t)q << 4);
        uint32_t v = 0, o = 0;
        for (int t = 0; t < 16; t++) { v = (v << 2) | c[t]; o |= c[t]; }
        dst_words[q] = v; seen |= o;
    }
    uint32_t v = 0;
    for (int32_t p = full << 4; p < len; p++) { v |= (uint32_t)codes[p] << (30 - 2 * (p & 15)); seen |= codes[p]; }
    dst_words[full] = v; dst_words[full + 1] = 0; dst_words[full + 2] = 0; dst_words[full + 3] = 0;
    return seen > 3 ? MTR_ERR_BAD_ARG : MTR_OK;
}
mtr_status mtr_upload_batch_packed(mtr_ctx *ctx, const uint32_t *packed, int64_t n_words, const int64_t *woff,
                                   const int32_t *lens, int32_t n_reads);

/* ---- wire form of the record table -------------------------------------------------------------------------------
 * A mtr_record is 2560 bytes because unit[] and unit_score[] are sized for MAX_PERIOD; a typical record uses 600.  The
 * wire form keeps what insert_an_alignment_into_set receives and nothing else, record after record:
 *   14 int32 (rep_start .. reserved, as in mtr_record) | rep_period unit bytes 'A','C','G','T', zero-padded to a
 *   multiple of 4 | rep_period int32 unit scores.
 * mtr_fetch_results_packed: the records of the last run in wire form, compacted on the device and copied into PINNED
 * host memory owned by the context (valid until the next upload / fetch on this context; do not free).  Reads
 * first_read .. first_read+n-1 only when n_reads_limit >= 0 (after MTR_ERR_DP_TOO_LARGE: the reads before the failing
 * one); pass -1 for all.  mtr_export_packed_device: the same into caller-owned DEVICE memory (for RCCL).
 * mtr_unpack_records / mtr_pack_records convert on the host (no device needed). */
#define MTR_WIRE_HEADER_BYTES 56
static inline int64_t mtr_wire_record_bytes(int32_t rep_period)
{
    const int64_t p = rep_period < 0 ? 0 : (rep_period > MTR_MAX_PERIOD ? MTR_MAX_PERIOD : rep_period);
    return MTR_WIRE_HEADER_BYTES + ((p + 3) & ~(int64_t)3) + 4 * p;
}
mtr_status mtr_fetch_results_packed(mtr_ctx *ctx, int32_t n_reads_limit, const uint8_t **out_blob, int64_t *out_bytes,
                                    const int32_t **out_counts, int64_t *out_total_records);
mtr_status mtr_export_packed_device(mtr_ctx *ctx, void *d_dst, int64_t capacity_bytes, int32_t *counts_host,
                                    int64_t *out_total_records, int64_t *out_bytes);
/* out must hold n_records entries; only the fields a record carries are written (unit is NUL-terminated) */
mtr_status mtr_unpack_records(const uint8_t *blob, int64_t bytes, int64_t n_records, mtr_record *out);
/* returns the number of bytes written, or -1 if capacity is too small */
int64_t    mtr_pack_records(const mtr_record *records, int64_t n_records, uint8_t *out, int64_t capacity);

/* ---- several GPUs in ONE process: the one exchange of the path (ABI 5) --------------------------------------------------
 * Reads shard over the GPUs of a node (isolated semantics make every read an independent unit); what is left
 * of handle_one_file.c:281-287's loop across GPUs is ONE exchange: the record tables travel to the process that chains and
 * prints (chaining.cpp).  A mtr_gather owns an RCCL communicator per GPU (ncclCommInitAll: one process, N devices) and
 * moves the wire form device to device over xGMI to the first GPU, from there in one copy to pinned host memory:
 *   mtr_gather_stage     a GPU's finished batch (after mtr_wait) is compacted to the wire form into a staging buffer on ITS OWN
 *                        device; returns a ticket.  Thread-safe: every GPU's host thread calls it for its own batches.
 *   mtr_gather_exchange  the staged tables named by tickets[0..n) - any number per GPU - go to the first GPU:
 *                        ncclGroupStart; per ticket ncclSend on the owner's communicator + ncclRecv on the first GPU's;
 *                        ncclGroupEnd; then ONE device-to-host copy.  out_ptrs[i] / out_bytes[i] = ticket i's table in pinned
 *                        host memory owned by the gather, valid until the next exchange; the tickets are released.  One
 *                        caller at a time.  Tables of the first GPU itself skip the collective (MTR_GATHER_SELF=1 sends
 *                        them through ncclSend/ncclRecv to itself as well: the RCCL path on a one-GPU box).
 * librccl.so is bound at run time by mtr_gather_create and only there, so a single-GPU process never loads it; RCCL needs
 * devices[] distinct (it takes a device once per communicator) - a gather over repeated devices works through the host copies. */
typedef struct mtr_gather mtr_gather;
mtr_status mtr_device_count(int32_t *out_count);
mtr_status mtr_gather_create(int32_t n_ranks, const int32_t *devices, mtr_gather **out);
void       mtr_gather_destroy(mtr_gather *g);
const char *mtr_gather_last_error(const mtr_gather *g);
mtr_status mtr_gather_stage(mtr_gather *g, int32_t rank, mtr_ctx *ctx, int32_t *counts_host, int64_t *out_total_records,
                            int64_t *out_bytes, int32_t *out_ticket);
mtr_status mtr_gather_exchange(mtr_gather *g, int32_t n_tickets, const int32_t *tickets, const uint8_t **out_ptrs, int64_t *out_bytes);
/* RCCL comes up in the background (loading librccl.so + ncclCommInitAll: ~2 s on an MI355X box, more than a 100 000-read job takes): mtr_gather_create
 * returns at once, and an exchange that finds RCCL not up yet - or not usable: librccl missing, a device given twice - copies its tables from every GPU's
 * staging buffer straight into the pinned host buffer instead; the results are the same.  mtr_gather_wait_ready blocks until RCCL is up (MTR_OK) or known
 * to be unusable (MTR_ERR_NO_DEVICE, reason in mtr_gather_last_error).  mtr_gather_get_stats: out[0] exchanges over RCCL, [1] exchanges straight to the host,
 * [2] / [3] their bytes, [4] ms RCCL took to come up, [5] 1 = up, 0 = still coming up, -1 = not usable.  mtr_gather_destroy does not wait for a library
 * that is still coming up: it leaves the object to the process's end, and such a process should leave through _exit. */
mtr_status mtr_gather_wait_ready(mtr_gather *g);
mtr_status mtr_gather_get_stats(const mtr_gather *g, int64_t *out, int32_t n);

/* File-order mode = the reference's own behaviour on a multi-read file (leak A, and H2) instead of
 * isolated semantics.  The reference's inputString_w_rand and orgInputString live for the whole file
 * (handle_one_file.c:85, mTR.h:65-67): the window look-ahead of a read (fill_directional_index.c:232) and the one-past
 * reads of its DPs (wrap_around_DP.c:243-245) see what the most recent LONGER read left beyond the part the current
 * read rewrites.  A mtr_file_state is the host shadow of that state for ONE file; give it the batches of the file in
 * file order (any context, any batch size — the state carries over) through mtr_upload_batch_in_file instead of
 * mtr_upload_batch, then run / fetch as usual.  Results then equal the reference run on the whole file at the lower
 * edge (the arguments of insert_an_alignment_into_set, read after read); the printed chain can still differ where two
 * chains tie, because the reference breaks those ties by heap address (chaining.cpp:201).  Reads no longer read
 * preceded take the normal path; the others run their range phase with plain 1024-bin window histograms (slower).
 * Shards of a file given to different GPUs need the state of the reads before the shard: feed those lengths/bases
 * through mtr_file_state_skip. */
typedef struct mtr_file_state mtr_file_state;
mtr_status mtr_file_state_create(mtr_file_state **out);
void       mtr_file_state_destroy(mtr_file_state *fs);
mtr_status mtr_upload_batch_in_file(mtr_ctx *ctx, mtr_file_state *fs, const uint8_t *bases, const int64_t *offsets,
                                    const int32_t *lens, int32_t n_reads);
/* advance the state over reads that another context / GPU processes (same arguments as an upload, no device work) */
mtr_status mtr_file_state_skip(mtr_file_state *fs, const uint8_t *bases, const int64_t *offsets, const int32_t *lens, int32_t n_reads);

/* orgInputString[L] and [L+1] as read i of the resident batch found them: 0 under isolated semantics, in file-order mode
 * the bases an earlier, longer read left there.  A repeat can end on them (wrap_around_DP.c:243-245), and a printer of
 * the -a alignments (mtr_alignments) needs them for its top row. */
mtr_status mtr_get_bases_after_read(const mtr_ctx *ctx, int32_t read_idx, uint8_t out[2]);

/* The -a alignments (replaces pretty_print_alignment, wrap_around_DP.c:57-213, for the repeats the caller chose to
 * report, i.e. after chaining): for n records of reads of the RESIDENT batch (mtr_process_batch / mtr_upload_batch
 * leaves it on the device) the wrap-around alignment of org[rep_start .. rep_end] against the record's unit with the
 * record's own (match_gain, mismatch_penalty, indel_penalty).  Result: one byte per alignment column in TRACEBACK order
 * (last column first, as the reference builds its three rows): 1 = match, 2 = mismatch, 3 = gap in the read
 * (deletion), 4 = gap in the unit (insertion).  (*out_off)[i] .. (*out_off)[i+1] is record i's slice of *out_ops;
 * (*out_end)[2i] is the 0-origin read position of the last aligned base and (*out_end)[2i+1] the 1-origin unit
 * column it is aligned to (where the walk back through the columns starts).  The three arrays are malloc'ed;
 * free() them. */
mtr_status mtr_alignments(mtr_ctx *ctx, int32_t n, const int32_t *read_idx, const mtr_record *records,
                          uint8_t **out_ops, int64_t **out_off, int32_t **out_end);

/* Device time of the last mtr_run_resident()/mtr_process_batch(), measured with HIP events on the stream the kernels were
 * launched on.  Ids: 0 = the range kernel when it runs alone (test entry points), 1 = the whole launch; and the phases of the
 * staged chain (launches = 0 for a batch the per-read kernel ran): 2 = candidate ranges, 3 = unit search (k-mer tables, seeds, walks),
 * 4 = two-parameter alignments, 5 = selection, 6 = revisions, 7 = comparison over k + replay of the sequential range loop. */
#define MTR_N_KERNEL_TIMES 10   /* 8, 9 (ABI 5): the two dominant kernels of the chain by themselves, launches of both passes summed -
                                 * 8 = mtr_k_revise_quads (four revisions per wavefront), 9 = mtr_k_dp2_quads (four alignments per wavefront);
                                 * launches = 0 when the batch ran one DP per wavefront (small batches) or in the per-read kernel */
typedef struct mtr_kernel_time { float ms; int32_t launches; } mtr_kernel_time;
mtr_status mtr_get_kernel_times(const mtr_ctx *ctx, mtr_kernel_time *out, int32_t n_kernels);

/* Work counters of the last run, accumulated on the device (used for the roofline figures):
 * [0] wrap-around DP calls, [1] DP cells, [2] DP rows, [3] revision DP calls, [4] revision DP cells,
 * [5] k-mer tables built, [6] k-mer look-ups, [7] candidate ranges, [8] ranges executed, [9] records,
 * [10] DI passes, [11] DI positions, [12] traceback steps, [13] undefined-behaviour guards hit. */
#define MTR_N_COUNTERS 56   /* [16..31]: shader-clock cycles per phase summed over wavefronts (total, DP forward, DP
                             * traceback, table build, seed list, walks, polish, revision votes, slot copies,
                             * revision DP forward / traceback, range-finder phases);
                             * [32] wrap_around_DP calls answered from the per-range memo (same window, same unit),
                             * [33] DP cells those calls would have filled, [34] k-mer tables proven unnecessary;
                             * [48..51] four-per-wavefront passes: cell bytes written / cells of the DPs served, alignments then revisions;
                             * [52] revisions answered by an identical revision of the same range, [53] reads the chain sent back to the per-read kernel,
                             * [54] candidate ranges the chain searched ([8]: the ranges the reference's loop reaches) */
mtr_status mtr_get_counters(const mtr_ctx *ctx, int64_t *out, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* MTR_HIP_H */
