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

int mtrh_chain(const mtr_record *recs, int n, int *chain)
{
    if (n <= 0) return 0;
    event *ev = (event *)malloc(sizeof(event) * 2 * (size_t)n);
    int *score = (int *)malloc(sizeof(int) * (size_t)n), *pred = (int *)malloc(sizeof(int) * (size_t)n);
    int *Y = (int *)malloc(sizeof(int) * (size_t)n);
    int ne = 0, ny = 0;
    for (int i = 0; i < n; i++) {
        score[i] = recs[i].num_matches; pred[i] = -1;
        if (recs[i].rep_start + MTRH_OVERLAP <= recs[i].rep_end) {
            ev[ne].key = recs[i].rep_start; ev[ne].idx = i; ev[ne].seq = ne; ne++;
            ev[ne].key = recs[i].rep_end - MTRH_OVERLAP; ev[ne].idx = i; ev[ne].seq = ne; ne++;
        }
    }
    qsort(ev, (size_t)ne, sizeof(event), ev_cmp);
    for (int e = 0; e < ne; e++) {
        const int a = ev[e].idx;
        if (ev[e].key == recs[a].rep_start) {                 /* Alignment::isStart */
            const int lim = recs[a].rep_start + MTRH_OVERLAP;
            int p = -1;
            for (int t = 0; t < ny; t++) {                    /* Y is sorted by end: the last one within the limit */
                if (recs[Y[t]].rep_end <= lim) p = Y[t];
                else break;
            }
            if (p >= 0) { pred[a] = p; score[a] += score[p]; }
        } else {
            int better = 0;
            for (int t = 0; t < ny && recs[Y[t]].rep_end <= recs[a].rep_end; t++)
                if (score[Y[t]] > score[a]) better = 1;
            if (better) continue;
            int pos = ny;
            while (pos > 0 && recs[Y[pos - 1]].rep_end > recs[a].rep_end) pos--;
            memmove(Y + pos + 1, Y + pos, sizeof(int) * (size_t)(ny - pos));
            Y[pos] = a; ny++;
            for (int t = 0; t < ny; t++)
                if (recs[Y[t]].rep_end >= recs[a].rep_end && score[Y[t]] < score[a]) {
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
    free(ev); free(score); free(pred); free(Y);
    return len;
}
