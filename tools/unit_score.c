/* unit_score.c — accuracy scorers for predicted repeat units.  Own implementation of the two
 * measures the reference's test_single_TR/test.sh reports (its util/count_match.cpp and util/comp_mTR_DP.cpp):
 *   us_is_rotation   1 if the predicted unit is a rotation of the true unit;
 *   us_match_ratio   wrap-around global alignment of the longer of the two strings against the shorter one taken
 *                    cyclically (match +1, mismatch -1, gap -1; free start position in the cycle), and the share
 *                    of matching columns on one optimal alignment (diagonal preferred, then a gap in the long
 *                    string, then a gap in the cycle).
 * Built on demand by tools/accuracy.py (gcc -O2 -shared -fPIC). */
#include <stdlib.h>
#include <string.h>

int us_is_rotation(const char *pred, const char *truth)
{
    size_t n = strlen(pred);
    if (n != strlen(truth)) return 0;
    if (n == 0) return 1;
    for (size_t r = 0; r < n; r++) {
        size_t i = 0;
        while (i < n && pred[i] == truth[(r + i) % n]) i++;
        if (i == n) return 1;
    }
    return 0;
}

double us_match_ratio(const char *s1, const char *s2)
{
    const char *a = s1, *b = s2;                 /* a = the longer string (columns), b = the cycle (rows 1..m, row 0 = row m) */
    if (strlen(a) < strlen(b)) { a = s2; b = s1; }
    const int n = (int)strlen(a), m = (int)strlen(b);
    if (n == 0 || m == 0) return 0.0;
    const int NEG = -(1 << 28);
    int *M = (int *)malloc(sizeof(int) * (size_t)(n + 1) * (size_t)(m + 1));
#define AT(i, j) M[(size_t)(j) * (size_t)(m + 1) + (size_t)(i)]
    for (int i = 0; i <= m; i++) AT(i, 0) = 0;
    for (int j = 1; j <= n; j++) {
        for (int i = 0; i <= m; i++) {
            int best = NEG;
            if (i != 0) {
                if (AT(i - 1, j) - 1 > best) best = AT(i - 1, j) - 1;                       /* gap in a */
                int d = AT(i - 1, j - 1) + (a[j - 1] == b[i - 1] ? 1 : -1);
                if (d > best) best = d;
            } else {
                int d = AT(m - 1, j - 1) + (a[j - 1] == b[m - 1] ? 1 : -1);                  /* row 0 continues row m */
                if (d > best) best = d;
            }
            if (AT(i, j - 1) - 1 > best) best = AT(i, j - 1) - 1;                           /* gap in the cycle */
            AT(i, j) = best;
        }
    }
    int x = m, best = AT(m, n);
    for (int i = 0; i <= m; i++) if (AT(i, n) > best) { best = AT(i, n); x = i; }
    int y = n, matches = 0, cols = 0;
    long guard = 4L * ((long)n + 2) * ((long)m + 2);
    while (y > 0 && guard-- > 0) {
        if (x == 0) {
            int eq = a[y - 1] == b[m - 1];
            if (AT(0, y) == AT(m - 1, y - 1) + (eq ? 1 : -1)) { matches += eq; cols++; x = m - 1; y--; continue; }
        } else {
            int eq = a[y - 1] == b[x - 1];
            if (AT(x, y) == AT(x - 1, y - 1) + (eq ? 1 : -1)) { matches += eq; cols++; x--; y--; continue; }
            if (AT(x, y) == AT(x - 1, y) - 1) { cols++; x--; continue; }
        }
        if (AT(x, y) == AT(x, y - 1) - 1) { cols++; y--; continue; }
        break;                                   /* cannot happen on a consistent matrix */
    }
#undef AT
    free(M);
    return cols > 0 ? (double)matches / (double)cols : 0.0;
}
