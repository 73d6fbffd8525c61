// mtr_common.h — constants, argument blocks and scratch layouts shared by the host side of the
// C-ABI (mtr_abi.hip) and the gfx950 kernels (k1_ranges.hip.inc, k2_units.hip.inc).
//
// Reference constants: mTR.h:31-58.
#pragma once
#include <stdint.h>
#include <stddef.h>

#define MTR_WAVE 64

#define MTRC_MAX_PERIOD 500
#define MTRC_MIN_PERIOD 2
#define MTRC_MIN_NUM_FREQ_UNIT 5
#define MTRC_MIN_WINDOW 5
#define MTRC_MAX_WINDOW 10240
#define MTRC_MIN_KMER 5
#define MTRC_MAX_KMER 15
#define MTRC_MAX_TIEBREAKS 1024
#define MTRC_MAX_SEEDS 100
#define MTRC_WRAP_DP_SIZE 200000000LL
#define MTRC_MAX_INPUT_LENGTH 1000000
#define MTRC_MAX_PASSES 20          // k=1: 5,10,20  k=3: 5..80  k=5: 5..10240

// A read is usable only if L + 2r <= MAX_INPUT_LENGTH (the reference's buffers overflow beyond it).
#define MTRC_MAX_SUPPORTED_LENGTH 833333

// Per-cell traceback code of the wrap-around DP = four flags, the same in every cell format:
//   bit 0: H > 0   bit 1: H != diag - mismatch   bit 2: H != left - indel   bit 3: the bases match.
// The traceback's priority order (wrap_around_DP.c:304-329) reads: bit 3 -> match; bit 0 clear -> stop; bit 1
// clear -> mismatch; bit 2 clear -> deletion; else insertion.  The one-parameter kernels store the canonical
// values below; the packed two-parameter kernel stores the raw flags (a mismatch cell may also have bit 2 set).
enum { DPC_Z = 0, DPC_X = 1, DPC_DEL = 3, DPC_INS = 7, DPC_M = 8, DPC_PENDING = 15 };
#define DPC_IS_M(f) (((f) & 8) != 0)
#define DPC_IS_X(f) (((f) & 11) == 1)
#define DPC_IS_Z(f) (((f) & 9) == 0)          /* neither a match nor H > 0 */

// device counters (index = MTR counters in include/mtr_hip.h)
enum { CNT_DP_CALLS = 0, CNT_DP_CELLS, CNT_DP_ROWS, CNT_REV_CALLS, CNT_REV_CELLS, CNT_TABLES, CNT_LOOKUPS,
       CNT_RANGES_CAND, CNT_RANGES_EXEC, CNT_RECORDS, CNT_DI_PASSES, CNT_DI_POS, CNT_TB_STEPS, CNT_UNDEFINED,
       CNT_GLOBAL_TABLES, CNT_RESERVED15,
       // shader-clock cycles per phase, summed over waves (profiling aid)
       CYC_TOTAL = 16, CYC_DP_FWD, CYC_DP_TB, CYC_TAB_BUILD, CYC_SEEDS, CYC_WALK, CYC_POLISH, CYC_REVISE_VOTE, CYC_SLOT_COPY,
       CYC_DP_FWD_REV, CYC_DP_TB_REV, CYC_K1_CODES, CYC_K1_PASSES, CYC_K1_EXTRACT, CYC_K1_DEDUP, CYC_K1_TOTAL,
       // work the kernel proved it did not have to repeat (results identical by construction)
       CNT_MEMO_HITS = 32, CNT_MEMO_CELLS, CNT_TABLES_SKIPPED, CYC_TB_REFILL, CNT_TB_REFILLS, CNT_WALK_STEPS, CNT_WALK_SLOW, CYC_WALK_SLOW,
       CNT_WALK_CALLS = 40, CNT_WALK_CLOSED, CYC_WALK_FAST, CNT_SPARE43, CNT_SPARE44, CNT_SPARE45, CNT_SPARE46, CNT_SPARE47,
       // the four-per-wavefront passes (dp_quad.hip.inc): bytes of cell matrix a pass writes (every row of every 16-lane group up to the
       // pass's longest member) against the cells of the DPs it was run for; revisions answered by another revision of their range
       CNT_QPASS_BYTES_DP2 = 48, CNT_QPASS_CELLS_DP2, CNT_QPASS_BYTES_REV, CNT_QPASS_CELLS_REV, CNT_REV_SHARED, CNT_SPARE53, CNT_SPARE54, CNT_SPARE55,
       CNT_N = 56 };

// status word values written by the kernels (first error wins)
enum { DEV_OK = 0, DEV_ERR_RANGE_OVERFLOW = 1, DEV_ERR_RECORD_OVERFLOW = 2, DEV_ERR_DP_TOO_LARGE = 3, DEV_ERR_INTERNAL = 4,
       DEV_ERR_STAGED_OVERFLOW = 5 };    // a buffer of the staged mode was too small: the host reruns the batch with the per-read kernel

static inline __host__ __device__ int mtrc_rand_len(int L) { return L < 1000 ? 100 : L / 10; }   // handle_one_read.c:194-201
static inline __host__ __device__ size_t mtrc_align(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline __host__ __device__ int mtrc_range_cap(int L) { return L / 2 + 64; }

// ---- K1 per-wave scratch layout (all offsets in bytes, 16-byte aligned) -------------------------
struct K1Layout {
    size_t codes[3];      // uint16 [ncode] for k = 1, 3, 5
    size_t tmp;           // double [MAX_PASSES][n]
    size_t di;            // double [n]
    size_t end;           // int32  [n]
    size_t w;             // int32  [n]
    size_t l_pos, l_end, l_w, l_di;   // compact list: int32/int32/int32/double [cap]
    size_t total;
    int n, ncode, cap;
};
static inline __host__ __device__ K1Layout k1_layout(int Lmax)
{
    K1Layout y;
    int r = mtrc_rand_len(Lmax);
    y.n = Lmax + 2 * r;
    long nc = (long)Lmax + 4L * r; if (nc > MTRC_MAX_INPUT_LENGTH) nc = MTRC_MAX_INPUT_LENGTH;
    if (nc < y.n) nc = y.n;
    // the passes read up to position L + r - k + 2w (w < L/2, w <= 10240); beyond N everything is zero
    long wmax = Lmax / 2 < MTRC_MAX_WINDOW ? Lmax / 2 : MTRC_MAX_WINDOW;
    long reach = (long)Lmax + r + 2 * wmax + 8;
    if (nc < reach) nc = reach;
    y.ncode = (int)nc + 16;
    y.cap = mtrc_range_cap(Lmax) + Lmax / 2;    // every alive entry (also those ending beyond the read)
    size_t o = 0;
    for (int i = 0; i < 3; i++) { y.codes[i] = o; o = mtrc_align(o + (size_t)y.ncode * 2, 16); }
    y.tmp = o;  o = mtrc_align(o + (size_t)MTRC_MAX_PASSES * (size_t)y.n * 8, 16);
    y.di = o;   o = mtrc_align(o + (size_t)y.n * 8, 16);
    y.end = o;  o = mtrc_align(o + (size_t)y.n * 4, 16);
    y.w = o;    o = mtrc_align(o + (size_t)y.n * 4, 16);
    y.l_pos = o; o = mtrc_align(o + (size_t)y.cap * 4, 16);
    y.l_end = o; o = mtrc_align(o + (size_t)y.cap * 4, 16);
    y.l_w = o;   o = mtrc_align(o + (size_t)y.cap * 4, 16);
    y.l_di = o;  o = mtrc_align(o + (size_t)y.cap * 8, 16);
    y.total = mtrc_align(o, 256);
    return y;
}

// bytes of LDS the K1 kernel needs for reads up to Lmax: 3*4^k uint16 counters per (k,w) pass
// (passes: fill_directional_index.c:559-582)
static inline __host__ __device__ size_t k1_hist_bytes(int Lmax)
{
    size_t n = 0;
    for (int k = 1; k <= 5; k += 2) {
        int max_w = (k == 1) ? 20 : (k == 3 ? 80 : MTRC_MAX_WINDOW);
        for (int w = MTRC_MIN_WINDOW; w <= max_w && w < Lmax / 2; w *= 2) n += 3u * (1u << (2 * k));
    }
    return mtrc_align(n * 2 + 64, 256);
}

// ---- K2 per-wave scratch layout ------------------------------------------------------------------
#define K2_NSLOT_FIXED 4           // unit slots: 0 = best of range, 1 = best of k, 2 = candidate, 3 = revision tmp
#define K2_POOL 24                 // + the candidate units of one range: <= 11 k x 2 walk directions (k2_range)
#define K2_NSLOT (K2_NSLOT_FIXED + K2_POOL)
#define K2_SLOT_UNIT 1024          // bytes of unit codes per slot (a revised unit can reach 2*499)
#define K2_SLOT_SCORE 512          // int32 per slot
#define K2_MEMO_N 32               // wrap_around_DP results remembered per candidate range (<= 11 k x 2 directions are made)
#define K2_MEMO_UNIT 512           // bytes per remembered unit (MAX_PERIOD - 1 = 499 bases at most)
#define K2_STEP_CACHE 2048         // look-ahead decisions remembered per k-mer table and walk direction (direct-mapped), for a table in global memory;
#define K2_STEP_CACHE_LDS 128      // for a table in LDS (windows up to K2_TAB_MAX_WIDTH; cleared per table, so kept small)
#define K2_RMEMO_N 16              // revision rounds remembered per candidate range (the k of a range often polish to the same unit)
struct K2Layout {
    size_t codes;                  // uint8 [cells]  traceback codes
    size_t unit0;                  // uint8 [K2_NSLOT][1024]
    size_t score0;                 // int32 [K2_NSLOT][512]
    size_t pool_meta, pool_res;    // int32 [K2_POOL][4] (period, k, direction), int32 [K2_POOL][16] (the DP result of the candidate)
    size_t cons, miss;             // int32 [501*5], [501*4]
    size_t pol_u, pol_rev;         // int32 [512] each (polish work arrays)
    size_t gkeys, gvals;           // int32 [gcap]: counts of the k-mer table (always) and keys of windows that do not fit LDS
    size_t ckeys, cslots;          // int32 [gcap / 2] each: the keys of such a table once more, compact, and their slots (the look-ahead scans these)
    size_t gkf, gsf, gkb, gsb;     // int32 [gcap / 2] each: the same keys and slots grouped by their first five digits (f) / last five (b), made when a
                                   //   table's first general look-ahead step in that direction needs them (k2_units.hip.inc: idx_build)
    size_t ties;                   // int32 [2][1024] tie lists of the look-ahead
    size_t memo_unit, memo_res;    // DP memo of the current range: uint8 [K2_MEMO_N][512] units, int32 [K2_MEMO_N][16] results
    size_t step_cache;             // int32 [2][K2_STEP_CACHE][4]: node, next node, number of the table the entry is for, 0 - the walks' general look-ahead steps, per direction
    size_t rmemo_key, rmemo_kunit, rmemo_val, rmemo_vunit;   // revision memo: int32 [N][16] + uint8 [N][512] in, int32 [N][16] + uint8 [N][1024] out
    size_t total;
    size_t cells; unsigned gcap;
};
static inline __host__ __device__ size_t k2_max_cells(int Lmax)
{
    // rows <= L, unit <= min(499, 2*rows/5) (revision can double a unit of <= rows/5 bases)
    long long u = 2LL * Lmax / 5 + 2; if (u > 499) u = 499;
    long long c = (long long)(Lmax + 2) * u;
    if (c > MTRC_WRAP_DP_SIZE) c = MTRC_WRAP_DP_SIZE;
    return (size_t)c;
}
// cells_cap > 0: a layout whose traceback-cell region holds only that many bytes (the kernels of the staged mode that
// align nothing need none of it: the walks of 42 kb reads take 1.3 MB per wavefront instead of 21 MB)
static inline __host__ __device__ K2Layout k2_layout(int Lmax, long long cells_cap = 0)
{
    K2Layout y;
    y.cells = k2_max_cells(Lmax);
    if (cells_cap > 0 && (size_t)cells_cap < y.cells) y.cells = (size_t)cells_cap;
    unsigned g = 2048; while (g < 2u * (unsigned)(Lmax + 2)) g <<= 1;
    y.gcap = g;
    size_t o = 0;
    y.codes = o; o = mtrc_align(o + y.cells, 256);
    y.unit0 = o; o = mtrc_align(o + (size_t)K2_NSLOT * K2_SLOT_UNIT, 16);
    y.score0 = o; o = mtrc_align(o + (size_t)K2_NSLOT * K2_SLOT_SCORE * 4, 16);
    y.pool_meta = o; o = mtrc_align(o + (size_t)K2_POOL * 4 * 4, 16);
    y.pool_res = o; o = mtrc_align(o + (size_t)K2_POOL * 16 * 4, 16);
    y.cons = o; o = mtrc_align(o + 501 * 5 * 4, 16);
    y.miss = o; o = mtrc_align(o + 501 * 4 * 4, 16);
    y.pol_u = o; o = mtrc_align(o + 512 * 4, 16);
    y.pol_rev = o; o = mtrc_align(o + 1024 * 4, 16);   // also holds a revised unit (<= 2*499 bases)
    y.ties = o; o = mtrc_align(o + 2 * MTRC_MAX_TIEBREAKS * 4, 16);
    y.memo_unit = o; o = mtrc_align(o + (size_t)K2_MEMO_N * K2_MEMO_UNIT, 16);
    y.memo_res = o; o = mtrc_align(o + (size_t)K2_MEMO_N * 16 * 4, 16);
    y.step_cache = o; o = mtrc_align(o + (size_t)2 * K2_STEP_CACHE * 4 * 4, 16);
    y.rmemo_key = o; o = mtrc_align(o + (size_t)K2_RMEMO_N * 16 * 4, 16);
    y.rmemo_kunit = o; o = mtrc_align(o + (size_t)K2_RMEMO_N * K2_MEMO_UNIT, 16);
    y.rmemo_val = o; o = mtrc_align(o + (size_t)K2_RMEMO_N * 16 * 4, 16);
    y.rmemo_vunit = o; o = mtrc_align(o + (size_t)K2_RMEMO_N * K2_SLOT_UNIT, 16);
    y.gkeys = o; o = mtrc_align(o + (size_t)g * 4, 16);
    y.gvals = o; o = mtrc_align(o + (size_t)g * 4, 16);
    y.ckeys = o; o = mtrc_align(o + (size_t)(g / 2) * 4, 16);
    y.cslots = o; o = mtrc_align(o + (size_t)(g / 2) * 4, 16);
    y.gkf = o; o = mtrc_align(o + (size_t)(g / 2) * 4, 16);
    y.gsf = o; o = mtrc_align(o + (size_t)(g / 2) * 4, 16);
    y.gkb = o; o = mtrc_align(o + (size_t)(g / 2) * 4, 16);
    y.gsb = o; o = mtrc_align(o + (size_t)(g / 2) * 4, 16);
    y.total = mtrc_align(o, 256);
    return y;
}

// ---- device record (same memory layout as mtr_record in include/mtr_hip.h) ------------------------
struct DevRecord {
    int32_t f[14];                 // rep_start .. indel_penalty, reserved
    char    unit[MTRC_MAX_PERIOD + 4];
    int32_t unit_score[MTRC_MAX_PERIOD];
};

// ---- kernel argument blocks -------------------------------------------------------------------------
struct BatchView {
    const uint32_t *packed;        // 2 bit/base, MSB first, every read starts on a word and is followed by >= 3 zero words
    const int64_t  *woff;          // word offset of every read
    const int32_t  *lens;
    const int32_t  *order;         // processing order (longest first)
    int32_t n_reads;
};

struct K1Args {
    BatchView b;
    const uint8_t *mt;             // MT19937(seed 0) % 4 stream (MT.h; fill_directional_index.c:129-131)
    int32_t manhattan;
    int32_t Lmax;
    uint8_t *scratch; size_t scratch_per_wave;
    // outputs: usable ranges per read
    int32_t *r_count; const int64_t *r_off;
    int32_t *r_start, *r_end, *r_w; uint64_t *r_di;
    int32_t *status; unsigned int *work_counter; unsigned long long *counters;
    // file-order mode (mtr_upload_batch_in_file): what earlier, longer reads of the file left in the reference's
    // process-wide inputString_w_rand beyond the part this read rewrites - tail[tail_off[rd] + (p - E)] is the entry at
    // position p >= E = max(L + 2r, min(L + 4r, 1e6)); beyond the slice (and always when tail == nullptr): zero
    const uint16_t *tail; const int64_t *tail_off;
};

struct K2Args {
    BatchView b;
    float min_match_ratio;
    int32_t Lmax;
    uint8_t *scratch; size_t scratch_per_wave;
    const int32_t *r_count; const int64_t *r_off;
    const int32_t *r_start; int32_t *r_end; const int32_t *r_w;    // r_end is overwritten (-1 = pruned)
    DevRecord *records; int32_t max_rec_per_read; int32_t *rec_count;   // rec_count = records FOUND (may exceed the slots: the host reruns such reads)
    const int64_t *rec_base;       // slot index of read rd's first record, or nullptr = rd * max_rec_per_read
    int32_t *status; unsigned int *work_counter; unsigned long long *counters;
    int32_t *fail_read;            // atomicMin of the index of a read whose DP exceeded WrapDPsize (the reference exits there)
    int32_t *trace; int32_t trace_cap; unsigned int *trace_n; int32_t trace_mask;   // bit t = record events of type t
    int32_t dp16_max_rows;         // DPs of up to this many rows may use the 16-bit kernels (tests set 0 to force the 32-bit ones)
    long long cells_cap;           // > 0: the scratch layout's cell region is this small (k2_layout)
    long long packed_words;        // words readable at b.packed (the batch's image + its zero slack): clamp of window prefetches
};

// -a alignments (pretty_print_alignment, wrap_around_DP.c:57-213) of a list of reported repeats of the resident batch
struct AlignArgs {
    BatchView b;
    int32_t n_tasks;
    const int32_t *read_idx, *rep_start, *rep_end, *gain, *mism, *indel;
    const uint8_t *units; const int32_t *unit_off;     // unit codes 0..3, concatenated
    uint8_t *ops; const int64_t *ops_off; int32_t *ops_len;   // task t writes ops[ops_off[t] .. ops_off[t+1]) and its length
    int32_t *ends;                 // [2t] = row (1-origin in the window), [2t+1] = unit column (1-origin) of the path's last cell
    uint8_t *scratch; size_t scratch_per_wave; size_t cells_cap;
    int32_t *status; unsigned int *work_counter; unsigned long long *counters;
    int32_t dp16_max_rows;
};

struct DpTestArgs {
    BatchView b;
    int32_t n_tasks;
    const int32_t *read_idx, *qs, *qe; const uint8_t *units; const int32_t *unit_off;
    const int32_t *gain, *mism, *indel;
    int32_t *out8;
    uint8_t *scratch; size_t scratch_per_wave; size_t cells_cap;
    int32_t *status; unsigned int *work_counter; unsigned long long *counters;
    int32_t dp16_max_rows;
};
