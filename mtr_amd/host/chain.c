/* chain.c — maximum-score chain of non-overlapping repeats (reference chaining.cpp:243-363).
 *
 * The reference keeps the alignments of a read in a std::set ordered by heap address; under the isolated
 * semantics this project reproduces, that order is the insertion order.  Sweep over the
 * start events (key = start) and end events (key = end - 10) in key order, ties in insertion order:
 *   start of a: its predecessor is the last element of Y (ordered by end, ties by insertion) whose end is
 *               <= start + 10; a's score (initially its match count) grows by the predecessor's score;
 *   end of a:   a enters Y unless some y already there scores more; then every y with the same end and a lower
 *               score is erased - the reference's erase loop skips the element after each erased one (:316-328).
 * The chain printed is the one ending at the last element of Y.
 */
#include "mtr_host.h"
#include <stdlib.h>
#include <string.h>

typedef struct { int key, idx, seq; } event;
static int ev_cmp(const void *a, const void *b)
{
    const event *x = (const event *)a, *y = (const event *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->seq - y->seq;
}

int mtrh_chain(const mtrh_rec *recs, int n, int *chain)
{
    if (n <= 0) return 0;
    if (n == 1) {                                              /* a single alignment is its own chain if it enters the sweep at all */
        if (recs[0].h[MTRH_REP_START] + MTRH_OVERLAP <= recs[0].h[MTRH_REP_END]) { chain[0] = 0; return 1; }
        return 0;
    }
    /* small reads (the common case) work on the stack */
    enum { SMALL = 32 };
    event ev_s[2 * SMALL]; int score_s[SMALL], pred_s[SMALL], Y_s[SMALL], st_s[SMALL], en_s[SMALL];
    const int small = n <= SMALL;
    event *ev = small ? ev_s : (event *)malloc(sizeof(event) * 2 * (size_t)n);
    int *score = small ? score_s : (int *)malloc(sizeof(int) * (size_t)n), *pred = small ? pred_s : (int *)malloc(sizeof(int) * (size_t)n);
    int *Y = small ? Y_s : (int *)malloc(sizeof(int) * (size_t)n);
    int *rs = small ? st_s : (int *)malloc(sizeof(int) * (size_t)n), *re = small ? en_s : (int *)malloc(sizeof(int) * (size_t)n);
    int ne = 0, ny = 0;
    for (int i = 0; i < n; i++) {
        score[i] = recs[i].h[MTRH_MATCHES]; pred[i] = -1; rs[i] = recs[i].h[MTRH_REP_START]; re[i] = recs[i].h[MTRH_REP_END];
        if (rs[i] + MTRH_OVERLAP <= re[i]) {
            ev[ne].key = rs[i]; ev[ne].idx = i; ev[ne].seq = ne; ne++;
            ev[ne].key = re[i] - MTRH_OVERLAP; ev[ne].idx = i; ev[ne].seq = ne; ne++;
        }
    }
    if (ne <= 16) {                                            /* insertion sort: stable, no call through a function pointer */
        for (int a = 1; a < ne; a++) { event t = ev[a]; int b = a; while (b > 0 && ev_cmp(&ev[b - 1], &t) > 0) { ev[b] = ev[b - 1]; b--; } ev[b] = t; }
    } else qsort(ev, (size_t)ne, sizeof(event), ev_cmp);
    for (int e = 0; e < ne; e++) {
        const int a = ev[e].idx;
        if (ev[e].key == rs[a]) {                 /* Alignment::isStart */
            const int lim = rs[a] + MTRH_OVERLAP;
            int p = -1;
            for (int t = 0; t < ny; t++) {                    /* Y is sorted by end: the last one within the limit */
                if (re[Y[t]] <= lim) p = Y[t];
                else break;
            }
            if (p >= 0) { pred[a] = p; score[a] += score[p]; }
        } else {
            int better = 0;
            for (int t = 0; t < ny && re[Y[t]] <= re[a]; t++)
                if (score[Y[t]] > score[a]) better = 1;
            if (better) continue;
            int pos = ny;
            while (pos > 0 && re[Y[pos - 1]] > re[a]) pos--;
            memmove(Y + pos + 1, Y + pos, sizeof(int) * (size_t)(ny - pos));
            Y[pos] = a; ny++;
            for (int t = 0; t < ny; t++)
                if (re[Y[t]] >= re[a] && score[Y[t]] < score[a]) {
                    memmove(Y + t, Y + t + 1, sizeof(int) * (size_t)(ny - t - 1));
                    ny--;                                      /* t now names the successor; the loop's t++ skips it */
                }
        }
    }
    int len = 0;
    if (ny > 0) {
        for (int a = Y[ny - 1]; a >= 0; a = pred[a]) len++;
        int p = len;
        for (int a = Y[ny - 1]; a >= 0; a = pred[a]) chain[--p] = a;
    }
    if (!small) { free(ev); free(score); free(pred); free(Y); free(rs); free(re); }
    return len;
}
