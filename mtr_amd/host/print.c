/* print.c — report lines (chaining.cpp:127-143) and, with -a, the alignment block
 * (chaining.cpp:164-167 + pretty_print_alignment, wrap_around_DP.c:57-213).
 *
 * The alignment is recomputed at print time, as the reference does, for the reported repeats only: wrap-around
 * local alignment of org[rep_start-1+i], i = 1..rep_end-rep_start+1, against the unit with the record's own
 * (gain, mismatch, indel).  Rows are kept as two rolling int rows plus one traceback code per cell.
 */
#include "mtr_host.h"
#include <stdlib.h>
#include <string.h>

enum { T_STOP = 0, T_MATCH, T_MISMATCH, T_DEL, T_INS };
static const char BASE[4] = { 'A', 'C', 'G', 'T' };

static int code_of(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }

static void print_alignment(FILE *fp, const mtrh_read *rd, const mtr_record *r)
{
    const int U = r->rep_period, G = r->match_gain, MM = r->mismatch_penalty, D = r->indel_penalty;
    const int rows = r->rep_end - r->rep_start + 1, base = r->rep_start - 1;
    fprintf(fp, "match gain = %i, mismatch penalty = %i, indel penalty = %i\n\n", G, MM, D);
    if (U <= 0 || rows <= 0) return;
    int *unit = (int *)malloc(sizeof(int) * (size_t)(U + 1));
    for (int j = 1; j <= U; j++) unit[j] = code_of(r->unit[j - 1]);
    int *prev = (int *)calloc((size_t)U + 1, sizeof(int)), *cur = (int *)calloc((size_t)U + 1, sizeof(int));
    uint8_t *tb = (uint8_t *)malloc((size_t)rows * (size_t)U);
    if (!unit || !prev || !cur || !tb) { fprintf(stderr, "cannot allocate the alignment matrix\n"); exit(EXIT_FAILURE); }
    int best = 0, bi = 0, bj = 0;
    for (int i = 1; i <= rows; i++) {
        const int p = base + i;
        const int x = (p >= 0 && p < rd->len) ? rd->codes[p] : ((p >= rd->len && p < rd->len + 2) ? rd->after[p - rd->len] : 0);   /* one past the read: 'A' under isolated semantics */
        uint8_t *t = tb + (size_t)(i - 1) * (size_t)U;
        for (int j = 1; j <= U; j++) {
            int v, c;
            if (x == unit[j]) { v = prev[j - 1] + G; c = T_MATCH; }
            else {
                const int sub = prev[j - 1] - MM, ins = prev[j] - D;
                v = sub > ins ? sub : ins;
                int del = -1;
                if (j > 1) { del = cur[j - 1] - D; if (del > v) v = del; }
                if (v <= 0) { v = 0; c = T_STOP; }
                else if (v == sub) c = T_MISMATCH;
                else if (j > 1 && v == del) c = T_DEL;
                else c = T_INS;                                          /* column 1 is settled after the row */
            }
            cur[j] = v; t[j - 1] = (uint8_t)c;
            if (best < v) { best = v; bi = i; bj = j; }
        }
        cur[0] = cur[U];
        if (t[0] == T_INS && cur[1] == cur[0] - D) t[0] = T_DEL;         /* the traceback tests H(i,U)-D first (:166) */
        int *sw = prev; prev = cur; cur = sw;
    }
    size_t cap = (size_t)rows + (size_t)rows * 0 + (size_t)U * 2 + 16, ncol = 0;
    cap += (size_t)rows;                                                  /* deletions add columns */
    char *a_in = (char *)malloc(cap), *a_sym = (char *)malloc(cap), *a_rep = (char *)malloc(cap);
    int i = bi, j = bj;
    while (i > 0) {
        const int c = tb[(size_t)(i - 1) * (size_t)U + (size_t)(j - 1)];
        if (c == T_STOP) break;
        if (ncol + 1 >= cap) { cap *= 2; a_in = (char *)realloc(a_in, cap); a_sym = (char *)realloc(a_sym, cap); a_rep = (char *)realloc(a_rep, cap); }
        const int p = base + i;
        const char xb = BASE[(p >= 0 && p < rd->len) ? rd->codes[p] : ((p >= rd->len && p < rd->len + 2) ? rd->after[p - rd->len] : 0)];
        if (c == T_MATCH) { a_in[ncol] = xb; a_sym[ncol] = '|'; a_rep[ncol] = BASE[unit[j]]; i--; j--; }
        else if (c == T_MISMATCH) { a_in[ncol] = xb; a_sym[ncol] = ' '; a_rep[ncol] = BASE[unit[j]]; i--; j--; }
        else if (c == T_DEL) { a_in[ncol] = '-'; a_sym[ncol] = ' '; a_rep[ncol] = BASE[unit[j]]; j--; }
        else { a_in[ncol] = xb; a_sym[ncol] = ' '; a_rep[ncol] = '-'; i--; }
        ncol++;
        if (j == 0) j = U;
    }
    for (long s = (long)ncol - 1; 0 <= s; s -= MTRH_ALIGN_WIDTH) {
        const long e = (-1 <= s - MTRH_ALIGN_WIDTH) ? s - MTRH_ALIGN_WIDTH : -1;
        for (long q = s; e < q; q--) fputc(a_in[q], fp);
        fputc('\n', fp);
        for (long q = s; e < q; q--) fputc(a_sym[q], fp);
        fputc('\n', fp);
        for (long q = s; e < q; q--) fputc(a_rep[q], fp);
        fputs("\n\n", fp);
    }
    free(unit); free(prev); free(cur); free(tb); free(a_in); free(a_sym); free(a_rep);
}

/* The same block from the path the device returned (mtr_alignments): ops[] = one byte per column, last column first
 * (1 match, 2 mismatch, 3 gap in the read, 4 gap in the unit); end_pos = read position of the last aligned base,
 * end_col = 1-origin unit column it is aligned to. */
static void print_alignment_ops(FILE *fp, const mtrh_read *rd, const mtr_record *r, const uint8_t *ops, int64_t n_ops, int end_pos, int end_col)
{
    const int U = r->rep_period;
    fprintf(fp, "match gain = %i, mismatch penalty = %i, indel penalty = %i\n\n", r->match_gain, r->mismatch_penalty, r->indel_penalty);
    if (U <= 0 || n_ops <= 0) return;
    char *a_in = (char *)malloc((size_t)n_ops), *a_sym = (char *)malloc((size_t)n_ops), *a_rep = (char *)malloc((size_t)n_ops);
    if (!a_in || !a_sym || !a_rep) { fprintf(stderr, "cannot allocate the alignment rows\n"); exit(EXIT_FAILURE); }
    int p = end_pos, j = end_col;
    for (int64_t q = 0; q < n_ops; q++) {
        const int c = ops[q];
        const char xb = BASE[(p >= 0 && p < rd->len) ? rd->codes[p] : ((p >= rd->len && p < rd->len + 2) ? rd->after[p - rd->len] : 0)];
        const char ub = r->unit[j - 1];
        if (c == T_MATCH) { a_in[q] = xb; a_sym[q] = '|'; a_rep[q] = ub; p--; j--; }
        else if (c == T_MISMATCH) { a_in[q] = xb; a_sym[q] = ' '; a_rep[q] = ub; p--; j--; }
        else if (c == T_DEL) { a_in[q] = '-'; a_sym[q] = ' '; a_rep[q] = ub; j--; }
        else { a_in[q] = xb; a_sym[q] = ' '; a_rep[q] = '-'; p--; }
        if (j == 0) j = U;
    }
    for (long s = (long)n_ops - 1; 0 <= s; s -= MTRH_ALIGN_WIDTH) {
        const long e = (-1 <= s - MTRH_ALIGN_WIDTH) ? s - MTRH_ALIGN_WIDTH : -1;
        for (long q = s; e < q; q--) fputc(a_in[q], fp);
        fputc('\n', fp);
        for (long q = s; e < q; q--) fputc(a_sym[q], fp);
        fputc('\n', fp);
        for (long q = s; e < q; q--) fputc(a_rep[q], fp);
        fputs("\n\n", fp);
    }
    free(a_in); free(a_sym); free(a_rep);
}

/* report lines of one chain with the alignment blocks computed on the device: ops/off/ends as returned by
 * mtr_alignments for this chain's records, first_task = index of the chain's first record among the tasks */
void mtrh_print_chain_ops(FILE *fp, const mtrh_read *rd, const mtr_record *recs, const int *chain, int n_chain,
                          const uint8_t *ops, const int64_t *off, const int32_t *ends, int64_t first_task)
{
    for (int t = 0; t < n_chain; t++) {
        const mtr_record *r = &recs[chain[t]];
        fprintf(fp, "%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%f\t%d\t%d\t%d\t%s\n", rd->id, rd->len, r->rep_start + 1, r->rep_end + 1,
                r->repeat_len, r->rep_period, r->num_freq_unit, r->num_matches, (float)r->num_matches / r->repeat_len,
                r->num_mismatches, r->num_insertions, r->num_deletions, r->unit);
        const int64_t k = first_task + t;
        fputc('\n', fp);
        print_alignment_ops(fp, rd, r, ops + off[k], off[k + 1] - off[k], ends[2 * k], ends[2 * k + 1]);
        fflush(fp);
    }
}

void mtrh_print_chain(FILE *fp, const mtrh_read *rd, const mtr_record *recs, const int *chain, int n_chain, int print_align)
{
    for (int t = 0; t < n_chain; t++) {
        const mtr_record *r = &recs[chain[t]];
        fprintf(fp, "%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%f\t%d\t%d\t%d\t%s\n", rd->id, rd->len, r->rep_start + 1, r->rep_end + 1,
                r->repeat_len, r->rep_period, r->num_freq_unit, r->num_matches, (float)r->num_matches / r->repeat_len,
                r->num_mismatches, r->num_insertions, r->num_deletions, r->unit);
        if (print_align) { fputc('\n', fp); print_alignment(fp, rd, r); }
        fflush(fp);
    }
}
