/*
 * mtr_oracle.c — TEST INFRASTRUCTURE ONLY (see mtr_oracle.h).
 *
 * CPU restatement of reference mTR's per-read pipeline under isolated semantics.  Every function
 * names the reference lines it follows.  The structure is deliberately the reference's (serial,
 * full int32 DP matrix, incremental window histograms) so that it is an independent check on the
 * HIP path, which uses different formulations (row scans, per-cell codes, parallel tables).
 *
 * Behaviour the reference leaves undefined is made explicit here and counted in
 * ctx->undefined_hits: int_unit[-1] in polish_repeat, a revised unit of length 0, rep_len < unit_len.
 */
#include "mtr_oracle.h"
#include "mtr_oracle_tables.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ---- constants: mTR.h:31-58 ------------------------------------------------------------------- */
#define MAX_PERIOD MTRO_MAX_PERIOD
#define MIN_PERIOD 2
#define MIN_NUM_FREQ_UNIT 5
#define MAX_LEN_OVERLAPPING 10
#define MIN_WINDOW 5
#define MAX_WINDOW 10240
#define MIN_KMER 5
#define MAX_KMER 15
#define MAX_TIEBREAKS 1024
#define MIN_JACCARD 0.98
/* mTR.h:51 WrapDPsize.  A variable so that the tests can make the failure reachable (MTR_TEST_WRAP_DP_SIZE, read by mtro_create - the same
 * variable the HIP library reads): no input of at most 833 333 bases is known to reach 2e8 cells */
static long long WRAP_DP_SIZE = 200000000;
#define COUNT_MAX_KMER 6
#define MAX_SEEDS 100
#define ALIGN_WIDTH 50

static int POW4[MAX_KMER + 1];

/* working record: the fields of repeat_in_read (mTR.h:99-119) that carry information */
typedef struct {
    int rep_start, rep_end, repeat_len, rep_period, copies, mat, mis, ins, del, kmer, G, MM, D;
    char str[MAX_PERIOD * 2 + 4];
    int score[MAX_PERIOD];
} rr_t;

struct mtro_ctx {
    int manhattan;
    float min_match_ratio;
    FILE *cap;
    int level;
    mtro_stats st;
    int64_t undefined_hits;
    int file_order;     /* 1 = the reference's own behaviour on a multi-read file: org[] and wrand[] are process-wide
                         * (handle_one_file.c:85, mTR.h:65-67) and keep what earlier reads left beyond the part the
                         * current read rewrites (leak A, and H2) */
    /* per-read buffers */
    int L;
    int *org;           /* L+2 entries, org[L]=org[L+1]=0 (isolated semantics) */
    int *wrand; size_t wrand_cap;
    double *di_tmp, *di; int *di_end, *di_w; size_t di_cap;
    int *hist[3];
    /* k-mer table of the current (window,k) */
    int *node; size_t node_cap;
    int tab_k;
    int *direct;        /* 4^6 counts */
    int *hkey, *hval; size_t hcap, hmask;
    /* DP matrix */
    int *dp; size_t dp_cap;
    char *al_in, *al_sym, *al_rep; size_t al_cap;
};

/* ---- MT19937 (MT.h, stock mt19937ar) ----------------------------------------------------------- */
typedef struct { uint32_t s[624]; int idx; } mt_t;
static void mt_seed(mt_t *m, uint32_t seed)
{
    m->s[0] = seed;
    for (int i = 1; i < 624; i++) m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}
static uint32_t mt_next(mt_t *m)
{
    if (m->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (m->s[i] & 0x80000000u) | (m->s[(i + 1) % 624] & 0x7fffffffu);
            m->s[i] = m->s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        m->idx = 0;
    }
    uint32_t y = m->s[m->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
void mtro_mt_bases(uint8_t *out, int n)
{
    mt_t m; mt_seed(&m, 0);
    for (int i = 0; i < n; i++) out[i] = (uint8_t)(mt_next(&m) % 4);
}

/* ---- record helpers: fill_directional_index.c:40-84 -------------------------------------------- */
static void rr_clear(rr_t *r)
{
    r->rep_start = r->rep_end = r->repeat_len = r->rep_period = r->copies = -1;
    r->mat = r->mis = r->ins = r->del = r->kmer = r->G = r->MM = r->D = -1;
    r->str[0] = 0;
    for (int i = 0; i < MAX_PERIOD; i++) r->score[i] = -1;
}
static float rr_ratio(const rr_t *r)
{   /* handle_one_read.c:137, consensus.c:562, wrap_around_DP.c:397: float / int-sum */
    return (float)r->mat / (r->mat + r->mis + r->ins + r->del);
}
static const char B2C[4] = { 'A', 'C', 'G', 'T' };
static int c2b(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }

static void *xrealloc(void *p, size_t n)
{
    void *q = realloc(p, n);
    if (!q) { fprintf(stderr, "mtr_oracle: out of memory (%zu bytes)\n", n); exit(EXIT_FAILURE); }
    return q;
}

mtro_ctx *mtro_create(int manhattan, float min_match_ratio)
{
    { const char *e = getenv("MTR_TEST_WRAP_DP_SIZE"); WRAP_DP_SIZE = e && atoll(e) > 0 ? atoll(e) : 200000000; }
    mtro_ctx *c = (mtro_ctx *)calloc(1, sizeof(*c));
    if (!c) return NULL;
    POW4[0] = 1; for (int i = 1; i <= MAX_KMER; i++) POW4[i] = POW4[i - 1] * 4;
    c->manhattan = manhattan; c->min_match_ratio = min_match_ratio;
    for (int v = 0; v < 3; v++) c->hist[v] = (int *)calloc(1024, sizeof(int));
    c->direct = (int *)calloc(4096, sizeof(int));
    return c;
}
void mtro_destroy(mtro_ctx *c)
{
    if (!c) return;
    free(c->org); free(c->wrand); free(c->di_tmp); free(c->di); free(c->di_end); free(c->di_w);
    for (int v = 0; v < 3; v++) free(c->hist[v]);
    free(c->node); free(c->direct); free(c->hkey); free(c->hval); free(c->dp);
    free(c->al_in); free(c->al_sym); free(c->al_rep);
    free(c);
}
void mtro_set_capture(mtro_ctx *c, FILE *cap, int level) { c->cap = cap; c->level = level; }
#define FILE_ORDER_WRAND ((size_t)MTRO_MAX_INPUT_LENGTH * 2 + (size_t)MTRO_MAX_INPUT_LENGTH / 2)
void mtro_set_file_order(mtro_ctx *c, int on)
{   /* the buffers become process-wide, zero at the start like the reference's freshly mapped malloc */
    c->file_order = on;
    if (!on) return;
    free(c->org); free(c->wrand);
    c->org = (int *)calloc((size_t)MTRO_MAX_INPUT_LENGTH + 4, sizeof(int));
    c->wrand_cap = FILE_ORDER_WRAND; c->wrand = (int *)calloc(c->wrand_cap, sizeof(int));
    if (!c->org || !c->wrand) { fprintf(stderr, "mtr_oracle: out of memory\n"); exit(EXIT_FAILURE); }
}
const mtro_stats *mtro_get_stats(const mtro_ctx *c) { return &c->st; }
void mtro_reset_stats(mtro_ctx *c) { memset(&c->st, 0, sizeof(c->st)); }

/* ================================================================================================
 * S1  candidate ranges: fill_directional_index.c:137-602
 * ==============================================================================================*/

/* init_inputString_surrounded_by_random_seq (fill_directional_index.c:137-169).  Buffer layout:
 * [r MT bases][read][r MT bases] rolling-encoded to k-mers for the first L+2r-k+1 slots; what
 * follows is k-1 raw flank bases, raw MT bases of the first fill pass up to min(L+4r,1e6), then
 * zeros (isolated semantics: fresh calloc). */
static void build_wrand(mtro_ctx *c, int k, int L, int r)
{
    if (!c->file_order) memset(c->wrand, 0, c->wrand_cap * sizeof(int));
    mt_t m; mt_seed(&m, 0);
    int *s = c->wrand;
    for (int i = 0; i < L + 4 * r && i < MTRO_MAX_INPUT_LENGTH; i++) s[i] = (int)(mt_next(&m) % 4);
    for (int i = 0; i < r; i++) s[i] = (int)(mt_next(&m) % 4);
    for (int i = 0; i < L; i++) s[r + i] = c->org[i];
    for (int i = 0; i < r; i++) s[r + L + i] = (int)(mt_next(&m) % 4);
    int carry = 0;
    for (int i = 0; i < k - 1; i++) carry = 4 * carry + s[i];
    for (int i = 0; i < L + 2 * r - k + 1; i++) {
        s[i] = 4 * carry + s[i + k - 1];
        carry = s[i] % POW4[k - 1];
    }
}

static inline int iabs(int x) { return x < 0 ? -x : x; }

/* one (k,w) pass: fill_directional_index_Manhattan / _PCC (fill_directional_index.c:171-450).
 * All window statistics are integers; they are kept exactly (int64) and converted to double only
 * where the reference forms the DI value, which reproduces its double arithmetic bit for bit. */
static void di_pass(mtro_ctx *c, int n, int w, int k, int r)
{
    const int *s = c->wrand;
    int nb = POW4[k];
    int *v0 = c->hist[0], *v1 = c->hist[1], *v2 = c->hist[2];
    for (int i = 0; i < n; i++) c->di_tmp[i] = -1;
    memset(v0, 0, 1024 * sizeof(int)); memset(v1, 0, 1024 * sizeof(int)); memset(v2, 0, 1024 * sizeof(int));
    for (int i = 0; i < w; i++) { v0[s[i]]++; v1[s[i + w]]++; v2[s[i + 2 * w]]++; }
    int64_t d01 = 0, d12 = 0, q0 = 0, q1 = 0, q2 = 0, ip01 = 0, ip12 = 0;
    for (int b = 0; b < nb; b++) {
        d01 += iabs(v0[b] - v1[b]); d12 += iabs(v1[b] - v2[b]);
        q0 += (int64_t)v0[b] * v0[b]; q1 += (int64_t)v1[b] * v1[b]; q2 += (int64_t)v2[b] * v2[b];
        ip01 += (int64_t)v0[b] * v1[b]; ip12 += (int64_t)v1[b] * v2[b];
    }
    const double sw = (double)w;             /* s_0 = s_1 = s_2 = w */
    const double nbd = (double)nb;
    int steps = n - w - r - k + 1;
    c->st.di_passes++; if (steps > 0) c->st.di_positions += steps;
    for (int i = 0; i < steps; i++) {
        double DI;
        if (c->manhattan) {
            DI = ((double)d01 - (double)d12) / (2 * sw);                    /* :211 */
        } else {
            double sd0 = sqrt((double)q0 * nbd - sw * sw);                  /* :339-357 */
            double sd1 = sqrt((double)q1 * nbd - sw * sw);
            double sd2 = sqrt((double)q2 * nbd - sw * sw);
            double p01 = (sd0 * sd1 > 0) ? ((double)ip01 * nbd - sw * sw) / (sd0 * sd1) : 0;
            double p12 = (sd1 * sd2 > 0) ? ((double)ip12 * nbd - sw * sw) / (sd1 * sd2) : 0;
            DI = p12 - p01;
        }
        c->di_tmp[i + w] = DI;                                              /* position i+w, :213 */
        /* slide the three windows by one: bins touched are x0..x3 */
        int x0 = s[i], x1 = s[i + w], x2 = s[i + 2 * w], x3 = s[i + 3 * w];
        int b01[3], n01 = 0, b12[3], n12 = 0;
        b01[n01++] = x0; if (x1 != x0) b01[n01++] = x1; if (x2 != x0 && x2 != x1) b01[n01++] = x2;
        b12[n12++] = x1; if (x2 != x1) b12[n12++] = x2; if (x3 != x1 && x3 != x2) b12[n12++] = x3;
        for (int t = 0; t < n01; t++) { int b = b01[t]; d01 -= iabs(v0[b] - v1[b]); ip01 -= (int64_t)v0[b] * v1[b]; }
        for (int t = 0; t < n12; t++) { int b = b12[t]; d12 -= iabs(v1[b] - v2[b]); ip12 -= (int64_t)v1[b] * v2[b]; }
        q0 -= (int64_t)v0[x0] * v0[x0]; if (x1 != x0) q0 -= (int64_t)v0[x1] * v0[x1];
        q1 -= (int64_t)v1[x1] * v1[x1]; if (x2 != x1) q1 -= (int64_t)v1[x2] * v1[x2];
        q2 -= (int64_t)v2[x2] * v2[x2]; if (x3 != x2) q2 -= (int64_t)v2[x3] * v2[x3];
        v0[x0]--; v0[x1]++; v1[x1]--; v1[x2]++; v2[x2]--; v2[x3]++;
        for (int t = 0; t < n01; t++) { int b = b01[t]; d01 += iabs(v0[b] - v1[b]); ip01 += (int64_t)v0[b] * v1[b]; }
        for (int t = 0; t < n12; t++) { int b = b12[t]; d12 += iabs(v1[b] - v2[b]); ip12 += (int64_t)v1[b] * v2[b]; }
        q0 += (int64_t)v0[x0] * v0[x0]; if (x1 != x0) q0 += (int64_t)v0[x1] * v0[x1];
        q1 += (int64_t)v1[x1] * v1[x1]; if (x2 != x1) q1 += (int64_t)v1[x2] * v1[x2];
        q2 += (int64_t)v2[x2] * v2[x2]; if (x3 != x2) q2 += (int64_t)v2[x3] * v2[x3];
    }
}

/* put_local_maximum_into_directional_index (fill_directional_index.c:467-503) */
static void extract_ranges(mtro_ctx *c, int n, int w)
{
    const double *tmp = c->di_tmp;
    double lmax = -1; int lmax_i = -1;
    for (int i = 0; i < n; i++) {
        if (lmax < tmp[i]) { lmax = tmp[i]; lmax_i = i; }
        if (lmax_i + w < i && lmax_i >= 0 /* DI[-1] is UB in the reference (H10) */ &&
            c->di[lmax_i] < lmax && 0 < lmax) {
            double lmin = 1; int lmin_j = lmax_i;
            for (int j = lmax_i; j < n; j++) {
                if (lmin > tmp[j]) { lmin = tmp[j]; lmin_j = j; }
                if (lmin_j + w < j) {
                    c->di[lmax_i] = lmax; c->di_w[lmax_i] = w; c->di_end[lmax_i] = lmin_j + w;
                    i = lmin_j + w;
                    break;
                }
            }
            lmax = -1;
        } else if (lmax_i + w < i && lmax_i < 0) {
            c->undefined_hits++;
        }
    }
}

/* remove_redundant_ranges (fill_directional_index.c:505-546) */
static void dedup_ranges(mtro_ctx *c, int L)
{
    for (int i = 0; i < L; i++) {
        int ib = i, ie = c->di_end[i];
        double idi = c->di[i];
        if (!(0 < idi)) continue;
        for (int j = i + 1; j <= ie; j++) {
            int jb = j, je = c->di_end[j];
            double jdi = c->di[j];
            if (!(0 < jdi)) continue;
            int mn_e = ie < je ? ie : je, mx_e = ie > je ? ie : je;
            int mx_b = ib > jb ? ib : jb, mn_b = ib < jb ? ib : jb;
            double jac = (double)(mn_e - mx_b) / (mx_e - mn_b);
            if (MIN_JACCARD < jac) {
                if (idi < jdi) { c->di[i] = -1; c->di_end[i] = -1; break; }
                else { c->di[j] = -1; c->di_end[j] = -1; }
            } else {
                if (ib >= jb && ie <= je && idi < jdi) { c->di[i] = -1; c->di_end[i] = -1; break; }
                if (ib <= jb && ie >= je && idi > jdi) { c->di[j] = -1; c->di_end[j] = -1; }
            }
        }
    }
}

/* fill_directional_index_with_end (fill_directional_index.c:549-602) */
static void fill_ranges(mtro_ctx *c, int L, int r)
{
    int n = L + 2 * r;
    for (int i = 0; i < n; i++) { c->di[i] = -1; c->di_end[i] = -1; c->di_w[i] = -1; }
    for (int k = 1; k <= 5; k += 2) {
        int max_w = (k == 1) ? 20 : (k == 3) ? 80 : MAX_WINDOW;
        build_wrand(c, k, L, r);
        for (int w = MIN_WINDOW; w <= max_w && w < L / 2; w *= 2) {
            di_pass(c, n, w, k, r);
            extract_ranges(c, n, w);
        }
    }
    for (int i = 0; i < L; i++) {
        c->di[i] = c->di[i + r]; c->di_end[i] = c->di_end[i + r] - r; c->di_w[i] = c->di_w[i + r];
    }
    for (int i = L; i < n; i++) { c->di[i] = -1; c->di_end[i] = -1; c->di_w[i] = -1; }
    dedup_ranges(c, L);
}

/* ================================================================================================
 * k-mer multiset of a window: consensus.c:37-253
 * ==============================================================================================*/
static inline size_t hslot(const mtro_ctx *c, int key) { return ((uint32_t)key * 2654435761u) & c->hmask; }

static int tab_get(mtro_ctx *c, int key)            /* freq_node, consensus.c:229-253 */
{
    c->st.kmer_lookups++;
    if (c->tab_k <= COUNT_MAX_KMER) return c->direct[key];
    size_t h = hslot(c, key);
    while (c->hkey[h] != -1) { if (c->hkey[h] == key) return c->hval[h]; h = (h + 1) & c->hmask; }
    return 0;
}
static int *tab_ref(mtro_ctx *c, int key)
{
    if (c->tab_k <= COUNT_MAX_KMER) return &c->direct[key];
    size_t h = hslot(c, key);
    while (c->hkey[h] != -1) { if (c->hkey[h] == key) return &c->hval[h]; h = (h + 1) & c->hmask; }
    c->hkey[h] = key; c->hval[h] = 0;
    return &c->hval[h];
}

/* init_inputString (consensus.c:37-60) + counting (consensus.c:138-146 / 167-196).  node[i-qs] for
 * i in [qs,qe]: the k-mer at i while i < min(qe, L-k+1), else the raw base code. */
static void tab_build(mtro_ctx *c, int k, int qs, int qe)
{
    int L = c->L, width = qe - qs + 1;
    if ((size_t)width > c->node_cap) { c->node_cap = (size_t)width * 2; c->node = (int *)xrealloc(c->node, c->node_cap * sizeof(int)); }
    int lim = qe < L - k + 1 ? qe : L - k + 1;
    int carry = 0;
    for (int i = qs; i < qs + k - 1; i++) carry = 4 * carry + (i < L ? c->org[i] : 0);
    for (int i = qs; i <= qe; i++) {
        if (i < lim) { int v = 4 * carry + c->org[i + k - 1]; c->node[i - qs] = v; carry = v % POW4[k - 1]; }
        else c->node[i - qs] = c->org[i];
    }
    c->tab_k = k;
    c->st.kmer_tables++;
    if (k <= COUNT_MAX_KMER) {
        memset(c->direct, 0, (size_t)POW4[k] * sizeof(int));
    } else {
        size_t need = 64; while (need < (size_t)width * 2 + 2) need <<= 1;
        if (need > c->hcap) { c->hcap = need; c->hkey = (int *)xrealloc(c->hkey, need * sizeof(int)); c->hval = (int *)xrealloc(c->hval, need * sizeof(int)); }
        c->hmask = need - 1;
        memset(c->hkey, 0xff, need * sizeof(int));
    }
    for (int i = 0; i < width; i++) (*tab_ref(c, c->node[i]))++;
}

/* generate_freqNode_return_list_maxNodes (consensus.c:132-229): max frequency, then the nodes
 * having it in first-occurrence order, each decremented once when listed, at most max_n. */
static int tab_seeds(mtro_ctx *c, int width, int *seeds, int max_n, int *max_freq)
{
    int mf = -1;
    for (int i = 0; i < width; i++) { int v = *tab_ref(c, c->node[i]); if (mf < v) mf = v; }
    int n = 0;
    for (int i = 0; i < width; i++) {
        int *p = tab_ref(c, c->node[i]);
        if (*p == mf) { seeds[n++] = c->node[i]; (*p)--; if (max_n <= n) break; }
    }
    *max_freq = mf;
    return n;
}

/* ================================================================================================
 * wrap-around DP: wrap_around_DP.c:222-354 (and the identical recurrence of consensus.c:851-962,
 * wrap_around_DP.c:57-185).  mode 0: counts; mode 1: votes; mode 2: alignment columns.
 * x_i = org[base+i], i = 1..rows.
 * ==============================================================================================*/
typedef struct {
    int end_i, end_j, stop_i;          /* max_i, max_j, i where the traceback stopped */
    int mat, mis, ins, del, scanned;
    int (*cons)[5]; int (*miss)[4];    /* mode 1 */
    int ncol;                          /* mode 2 */
} dp_out;

static int dp_run(mtro_ctx *c, int base, int rows, const int *unit /*1-origin*/, int U, int G, int MM, int D, int mode, dp_out *o)
{
    const int *x = c->org + base;      /* x[i] valid for base+i <= L+1 */
    int next = U + 1;
    size_t need = (size_t)next * (size_t)(rows + 1) + 1;
    /* the reference dies when a cell index next*i+j reaches WrapDPsize (wrap_around_DP.c:259-262) */
    if (U > 0 && rows > 0 && (size_t)next * (size_t)rows + (size_t)U >= (size_t)WRAP_DP_SIZE) return -1;
    if (need > c->dp_cap) { c->dp_cap = need + need / 2; free(c->dp); c->dp = (int *)malloc(c->dp_cap * sizeof(int)); if (!c->dp) { fprintf(stderr, "mtr_oracle: DP matrix alloc failed\n"); exit(EXIT_FAILURE);} }
    int *H = c->dp;
    if (rows < U) c->undefined_hits++;                        /* row 0 only cleared for j<=rows */
    for (int j = 0; j <= U; j++) H[j] = 0;
    int best = 0, bi = 0, bj = 0;
    for (int i = 1; i <= rows; i++) {
        int *cur = H + (size_t)next * i; const int *prv = cur - next;
        int xi = x[i];
        for (int j = 1; j <= U; j++) {
            int v;
            if (xi == unit[j]) v = prv[j - 1] + G;
            else {
                int a = prv[j - 1] - MM, b = prv[j] - D;
                v = a > b ? a : b;
                if (j > 1) { int d = cur[j - 1] - D; if (d > v) v = d; }
                if (v < 0) v = 0;
            }
            cur[j] = v;
            if (best < v) { best = v; bi = i; bj = j; }
        }
        cur[0] = cur[U];
    }
    if (mode == 0) { c->st.dp_calls++; c->st.dp_cells += (int64_t)rows * U; c->st.dp_rows += rows; if ((int64_t)rows * U > c->st.dp_max_cells) c->st.dp_max_cells = (int64_t)rows * U; }
    else if (mode == 1) { c->st.revise_dp_calls++; c->st.revise_dp_cells += (int64_t)rows * U; }
    /* traceback: wrap_around_DP.c:287-335 */
    int i = bi, j = bj, val = best;
    o->end_i = bi; o->end_j = bj; o->mat = o->mis = o->ins = o->del = o->scanned = 0; o->ncol = 0;
    if (mode == 2 && (size_t)(rows + U + 2) * 2 > c->al_cap) {
        c->al_cap = (size_t)(rows + U + 2) * 4;
        c->al_in = (char *)xrealloc(c->al_in, c->al_cap); c->al_sym = (char *)xrealloc(c->al_sym, c->al_cap); c->al_rep = (char *)xrealloc(c->al_rep, c->al_cap);
    }
    if (U == 0) { o->stop_i = i; return 0; }
    if (j == 0) j = U;
    while (i > 0 && H[(size_t)next * i + j] > 0) {
        const int *cur = H + (size_t)next * i; const int *prv = cur - next;
        int vm = prv[j - 1] + G, vx = prv[j - 1] - MM, vi = prv[j] - D, vd = cur[j - 1] - D;
        int same = (x[i] == unit[j]);
        int p = o->ncol;
        if (val == vm && same) {
            if (mode == 1) o->cons[j][x[i]]++;
            if (mode == 2) { c->al_in[p] = B2C[x[i]]; c->al_sym[p] = '|'; c->al_rep[p] = B2C[unit[j]]; o->ncol++; }
            val -= G; i--; j--; o->mat++; o->scanned++;
        } else if (val == vx && !same) {
            if (mode == 1) o->cons[j][x[i]]++;
            if (mode == 2) { c->al_in[p] = B2C[x[i]]; c->al_sym[p] = ' '; c->al_rep[p] = B2C[unit[j]]; o->ncol++; }
            val += MM; i--; j--; o->mis++; o->scanned++;
        } else if (val == vd) {
            if (mode == 1) o->cons[j][4]++;
            if (mode == 2) { c->al_in[p] = '-'; c->al_sym[p] = ' '; c->al_rep[p] = B2C[unit[j]]; o->ncol++; }
            val += D; j--; o->del++; o->scanned++;
        } else if (val == vi) {
            if (mode == 1) o->miss[j][x[i]]++;
            if (mode == 2) { c->al_in[p] = B2C[x[i]]; c->al_sym[p] = ' '; c->al_rep[p] = '-'; o->ncol++; }
            val += D; i--; o->ins++;
        } else if (val == 0) {
            break;
        } else {
            fprintf(stderr, "fatal error in wrap-around DP max_wrd = %i\n", val);
            exit(EXIT_FAILURE);
        }
        if (j == 0) j = U;
    }
    o->stop_i = i;
    return 0;
}

static void unit_codes(const char *s, int U, int *u1 /*1-origin, U+1 entries*/)
{
    for (int i = 0; i < U; i++) u1[i + 1] = c2b(s[i]);
}

/* wrap_around_DP_sub (wrap_around_DP.c:222-354) */
static void dp_sub(mtro_ctx *c, int qs, int qe, rr_t *r, int G, int MM, int D)
{
    int U = r->rep_period;
    int u1[MAX_PERIOD + 1];
    unit_codes(r->str, U, u1);
    dp_out o; memset(&o, 0, sizeof(o));
    int rows = qe - qs + 1;
    char unit_in[MAX_PERIOD + 1]; memcpy(unit_in, r->str, (size_t)U); unit_in[U] = 0;
    if (dp_run(c, qs, rows, u1, U, G, MM, D, 0, &o) != 0) { fprintf(stderr, "You need to increse the value of WrapDPsize.\n"); exit(EXIT_FAILURE); }
    r->rep_start = qs + o.stop_i + 1;
    r->rep_end = qs + o.end_i;
    r->repeat_len = o.end_i - o.stop_i;
    r->copies = U > 0 ? o.scanned / U : 0;
    r->mat = o.mat; r->mis = o.mis; r->ins = o.ins; r->del = o.del;
    r->G = G; r->MM = MM; r->D = D;
    if (c->cap && c->level >= 1)
        fprintf(c->cap, "{\"t\":\"G3\",\"qs\":%d,\"qe\":%d,\"unit\":\"%s\",\"G\":%d,\"MM\":%d,\"D\":%d,\"out\":[%d,%d,%d,%d,%d,%d,%d,%d]}\n",
                qs, qe, unit_in, G, MM, D, r->rep_start, r->rep_end, r->repeat_len, r->copies, r->mat, r->mis, r->ins, r->del);
}

/* wrap_around_DP (wrap_around_DP.c:357-429): (1,1,3) then (1,3,1), strictly better float ratio wins */
static void dp_two_params(mtro_ctx *c, int qs, int qe, rr_t *r)
{
    static const int P[2][3] = { { 1, 1, 3 }, { 1, 3, 1 } };
    rr_t best, t; rr_clear(&best);
    float best_ratio = -1;
    for (int p = 0; p < 2; p++) {
        t = *r;
        dp_sub(c, qs, qe, &t, P[p][0], P[p][1], P[p][2]);
        float ratio = rr_ratio(&t);
        if (best_ratio < ratio) { best = t; best_ratio = ratio; }
    }
    *r = best;
}

/* ================================================================================================
 * De Bruijn greedy cycle search: consensus.c:269-582
 * ==============================================================================================*/
static void put_unit(rr_t *r, const int *bases, const int *score, int n)
{
    for (int i = 0; i < n; i++) { r->str[i] = B2C[bases[i]]; r->score[i] = score[i]; }
    r->str[n] = 0;
}

static int walk(mtro_ctx *c, int backward, int qs, int qe, int k, int seed, rr_t *r)
{
    int bases[MAX_PERIOD], score[MAX_PERIOD];
    static int ties[MAX_TIEBREAKS], fresh[MAX_TIEBREAKS];
    int node = seed, period = 0;
    int max_steps = (qe - qs) / MIN_NUM_FREQ_UNIT; if (max_steps > MAX_PERIOD) max_steps = MAX_PERIOD;
    for (int l = 0; l < max_steps; l++) {
        if (!backward) { bases[l] = node / POW4[k - 1]; score[l] = tab_get(c, node); }       /* :286-293 */
        int look = (l < 10) ? 1 : k;                                                          /* :299-303 */
        int nt = 1, m, best_digit = 0; ties[0] = 0;
        for (m = 1; m <= look; m++) {
            int best = -1, nf = 0; best_digit = 0;
            for (int i = 0; i < nt; i++)
                for (int j = 0; j < 4; j++) {
                    int digit, cand;
                    if (!backward) { digit = 4 * ties[i] + j; cand = POW4[m] * (node % POW4[k - m]) + digit; }  /* :311-312 */
                    else { digit = j * POW4[m - 1] + ties[i]; cand = digit * POW4[k - m] + node / POW4[m]; }    /* :392-393 */
                    int cnt = tab_get(c, cand);
                    if (best < cnt) { best = cnt; best_digit = digit; nf = 0; fresh[nf++] = digit; }
                    else if (best == cnt) { if (nf < MAX_TIEBREAKS) fresh[nf++] = digit; }
                }
            if (nf <= 1) break;         /* forward tests ==1, backward <=1; nf is never 0 */
            memcpy(ties, fresh, (size_t)nf * sizeof(int)); nt = nf;
        }
        if (!backward) node = 4 * (node % POW4[k - 1]) + best_digit / POW4[m - 1];            /* :335, H6 */
        else {
            node = (best_digit % 4) * POW4[k - 1] + node / 4;                                  /* :418 */
            bases[l] = node / POW4[k - 1]; score[l] = tab_get(c, node);                        /* :421-426 */
        }
        if (node == seed) { period = l + 1; if (MAX_PERIOD <= period) period = 0; break; }
    }
    r->rep_period = period;
    if (period == 0) return 0;   /* (backward failure path of the reference computes a sub-goal nobody reads) */
    if (backward)
        for (int i = 0; i < period / 2; i++) {
            int t = bases[period - 1 - i]; bases[period - 1 - i] = bases[i]; bases[i] = t;
            t = score[period - 1 - i]; score[period - 1 - i] = score[i]; score[i] = t;
        }
    put_unit(r, bases, score, period);
    return 1;
}

static void cap_rr(FILE *f, const rr_t *r)
{
    fprintf(f, "{\"rep_start\":%d,\"rep_end\":%d,\"repeat_len\":%d,\"period\":%d,\"copies\":%d,"
               "\"mat\":%d,\"mis\":%d,\"ins\":%d,\"del\":%d,\"k\":%d,\"G\":%d,\"MM\":%d,\"D\":%d,\"unit\":\"",
            r->rep_start, r->rep_end, r->repeat_len, r->rep_period, r->copies, r->mat, r->mis, r->ins, r->del,
            r->kmer, r->G, r->MM, r->D);
    if (r->rep_period > 0 && r->rep_period < MAX_PERIOD) fputs(r->str, f);
    fprintf(f, "\",\"score\":[");
    if (r->rep_period > 0 && r->rep_period < MAX_PERIOD)
        for (int i = 0; i < r->rep_period; i++) fprintf(f, "%s%d", i ? "," : "", r->score[i]);
    fprintf(f, "]}");
}

/* search_De_Bruijn_graph (consensus.c:507-582).  Returns the found flag of the LAST attempt (H4). */
static int search_unit(mtro_ctx *c, int qs, int qe, rr_t *r)
{
    int k = r->kmer, width = qe - qs + 1;
    tab_build(c, k, qs, qe);
    int seeds[MAX_SEEDS], max_freq;
    int ns = tab_seeds(c, width, seeds, MAX_SEEDS, &max_freq);
    rr_t best, t; rr_clear(&best);
    float best_ratio = -1;
    int found = 0;
    if (MIN_NUM_FREQ_UNIT < max_freq) {
        c->st.searches_passing_maxfreq++;
        for (int dir = 0; dir < 2; dir++)
            for (int i = 0; i < ns; i++) {
                t = *r;
                found = walk(c, dir, qs, qe, k, seeds[i], &t);
                if (found) {
                    dp_two_params(c, qs, qe, &t);
                    float ratio = rr_ratio(&t);
                    if (best_ratio < ratio && c->min_match_ratio <= ratio && MIN_NUM_FREQ_UNIT < t.copies &&
                        MIN_PERIOD <= t.rep_period && t.rep_period < MAX_PERIOD) { best_ratio = ratio; best = t; }
                    break;
                }
            }
    }
    *r = best;
    if (c->cap && (c->level >= 3 || (c->level >= 2 && (found || r->rep_period != -1)))) {
        fprintf(c->cap, "{\"t\":\"G2\",\"qs\":%d,\"qe\":%d,\"k\":%d,\"found\":%d,\"rr\":", qs, qe, k, found);
        cap_rr(c->cap, r);
        fprintf(c->cap, "}\n");
    }
    return found;
}

/* ================================================================================================
 * unit polishing and revision: consensus.c:584-1087
 * ==============================================================================================*/
static int align_score(mtro_ctx *c, int start, int k, int node, int period, const int *u)
{   /* score_for_alignment, consensus.c:584-595 */
    int sum = 0;
    for (int j = start; 0 <= j && start - k < j; j--) { node = u[j % period] * POW4[k - 1] + node / 4; sum += tab_get(c, node); }
    return sum;
}
static int suspicious(const rr_t *r, int j)
{   /* consensus.c:597-608 */
    int cnt = 0;
    for (int i = 0; i < r->kmer - 1 && 0 <= j - i; i++) if (r->score[j - i] < 2) cnt++;
    return (r->kmer - 1) * 0.8 < (double)cnt;
}

/* polish_repeat (consensus.c:610-704) */
static void polish(mtro_ctx *c, rr_t *r)
{
    int k = r->kmer, period = r->rep_period;
    char unit_in[MAX_PERIOD + 1]; int Uin = period < 0 ? 0 : period; memcpy(unit_in, r->str, (size_t)Uin); unit_in[Uin] = 0;
    int rs = r->rep_start, re = r->rep_end;
    if (period > k) {
        tab_build(c, k, rs, re);
        int u[MAX_PERIOD], rev[MAX_PERIOD];
        for (int i = 0; i < period; i++) u[i] = c2b(r->str[i]);
        int jr = MAX_PERIOD - 1, ok = 1;
        int node = 0;
        for (int i = 0; i < k; i++) node += u[i] * POW4[k - 1 - i];
        for (int j = period - 1; 0 <= j;) {
            int ref = u[j] * POW4[k - 1] + node / 4;
            int bestf = tab_get(c, ref);
            node = ref;
            if (r->score[j] == 1 && suspicious(r, j)) {
                for (int l = 0; l < 4; l++) {
                    int alt = (ref + (l - u[j]) * POW4[k - 1]) % POW4[k];
                    if (bestf < tab_get(c, alt)) { bestf = tab_get(c, alt); node = alt; }
                }
                if (node == ref) rev[jr--] = u[j--];
                else {
                    int sd = align_score(c, j, k, node, period, u);
                    int ss = align_score(c, j - 1, k, node, period, u);
                    int si = -1;
                    int prevb;
                    if (j == 0) { c->undefined_hits++; prevb = -1; }      /* int_unit[-1] in the reference (H10) */
                    else prevb = u[(j - 1) % period];
                    if (node / POW4[k - 1] == prevb) si = align_score(c, j - 2, k, node, period, u);
                    rev[jr--] = node / POW4[k - 1];
                    int mx = sd > ss ? sd : ss; if (si > mx) mx = si;
                    if (mx == sd) { /* keep j */ } else if (mx == ss) j -= 1; else j -= 2;
                }
            } else rev[jr--] = u[j--];
            if (jr < 0) { ok = 0; break; }
        }
        if (ok) {
            int np = (MAX_PERIOD - 1) - jr;
            r->rep_period = np;
            for (int i = 0; i < np; i++) r->str[i] = B2C[rev[i + jr + 1]];
            r->str[np] = 0;
        }
    }
    if (c->cap && c->level >= 1) {
        fprintf(c->cap, "{\"t\":\"G3p\",\"rep_start\":%d,\"rep_end\":%d,\"k\":%d,\"in\":\"%s\",\"in_score\":[", rs, re, k, unit_in);
        for (int i = 0; i < Uin; i++) fprintf(c->cap, "%s%d", i ? "," : "", r->score[i]);
        fprintf(c->cap, "],\"out\":\"%s\"}\n", r->str);
    }
}

static int min_missing(int period, double err, int coverage)
{   /* consensus.c:777-820 */
    int a = period > 200 ? 0 : period > 150 ? 1 : period > 100 ? 2 : period > 75 ? 3 : period > 50 ? 4 :
            period > 30 ? 5 : period > 20 ? 6 : period > 10 ? 7 : period > 5 ? 8 : 9;
    int b = err > 0.25 ? 0 : err > 0.225 ? 1 : err > 0.2 ? 2 : err > 0.175 ? 3 : err > 0.15 ? 4 :
            err > 0.125 ? 5 : err > 0.1 ? 6 : err > 0.075 ? 7 : err > 0.05 ? 8 : 9;
    int cc = coverage <= 1 ? 0 : coverage >= 20 ? 19 : coverage - 1;
    return MTR_ORACLE_MIN_MISSING_DIGITS[(a * 10 + b) * 20 + cc] - '0';
}

/* revise_representative_unit_sub (consensus.c:851-1046) */
static void revise_sub(mtro_ctx *c, rr_t *r, int G, int MM, int D)
{
    int U = r->rep_period, rs = r->rep_start, re = r->rep_end;
    char unit_in[MAX_PERIOD + 1]; memcpy(unit_in, r->str, (size_t)U); unit_in[U] = 0;
    r->G = G; r->MM = MM; r->D = D;
    int u1[MAX_PERIOD + 1]; unit_codes(r->str, U, u1);
    static int cons[MAX_PERIOD + 1][5], miss[MAX_PERIOD + 1][4];
    memset(cons, 0, sizeof(cons)); memset(miss, 0, sizeof(miss));
    dp_out o; memset(&o, 0, sizeof(o)); o.cons = cons; o.miss = miss;
    if (dp_run(c, rs, re - rs + 1, u1, U, G, MM, D, 1, &o) != 0) { fprintf(stderr, "You need to increse the value of WrapDPsize.\n"); exit(EXIT_FAILURE); }
    int out[MAX_PERIOD * 2 + 2], n = 0;
    int coverage = r->repeat_len / r->rep_period;
    for (int j = 1; j <= U; j++) {
        int mv = -1, mb = -1;
        for (int q = 0; q < 5; q++) if (mv < cons[j][q]) { mv = cons[j][q]; mb = q; }
        if (mb < 4) out[n++] = mb;
        mv = -1; int mm = -1;
        for (int q = 0; q < 4; q++) if (mv < miss[j][q]) { mv = miss[j][q]; mm = q; }
        if (5 <= coverage && coverage <= 20) {
            double ratio = (double)(r->mis + r->ins + r->del) / r->repeat_len;
            if (min_missing(r->rep_period, ratio, coverage) <= mv && 0 <= mm && mm <= 3) out[n++] = mm;
        }
    }
    r->rep_period = n;
    for (int i = 0; i < n; i++) r->str[i] = B2C[out[i]];
    r->str[n] = 0;
    if (c->cap && c->level >= 1)
        fprintf(c->cap, "{\"t\":\"G3r\",\"rep_start\":%d,\"rep_end\":%d,\"repeat_len\":%d,\"mis\":%d,\"ins\":%d,\"del\":%d,"
                        "\"G\":%d,\"MM\":%d,\"D\":%d,\"in\":\"%s\",\"out_period\":%d,\"out\":\"%s\"}\n",
                rs, re, r->repeat_len, r->mis, r->ins, r->del, G, MM, D, unit_in, n, r->str);
}

/* revise_representative_unit (consensus.c:1048-1087) */
static void revise(mtro_ctx *c, rr_t *r)
{
    static const int P[2][3] = { { 5, 1, 1 }, { 1, 1, 3 } };
    polish(c, r);
    float base_ratio = rr_ratio(r);               /* not refreshed between the rounds (H7) */
    for (int p = 0; p < 2; p++) {
        rr_t t = *r;
        revise_sub(c, &t, P[p][0], P[p][1], P[p][2]);
        if (t.rep_period < MAX_PERIOD) {
            if (t.rep_period <= 0) { c->undefined_hits++; continue; }   /* reference divides by zero */
            dp_sub(c, t.rep_start, t.rep_end, &t, P[p][0], P[p][1], P[p][2]);
            if (base_ratio < rr_ratio(&t)) *r = t;
        }
    }
}

/* ================================================================================================
 * per range / per read: handle_one_read.c:77-261
 * ==============================================================================================*/
static void find_unit_for_k(mtro_ctx *c, int qs, int qe, rr_t *r)
{   /* find_tandem_repeat_sub, handle_one_read.c:77-100 */
    int found = search_unit(c, qs, qe, r);
    if (!found) { rr_clear(r); return; }
    if ((int64_t)r->rep_period * (qe - qs + 1) > WRAP_DP_SIZE) { fprintf(stderr, "You need to increse the value of WrapDPsize.\n"); rr_clear(r); return; }
    int coverage = r->repeat_len / r->rep_period;
    if (5 <= coverage && coverage <= 20 && 5 < r->rep_period) revise(c, r);
}

static void find_unit(mtro_ctx *c, int qs, int qe, int w, rr_t *out)
{   /* find_tandem_repeat, handle_one_read.c:102-154 */
    int min_k, max_k;
    if (w < 100) { min_k = MIN_KMER - 3; max_k = MAX_KMER - 5; }
    else if (w < 1000) { min_k = MIN_KMER - 3; max_k = MAX_KMER - 3; }
    else { min_k = MIN_KMER; max_k = MAX_KMER; }
    float best = -1;
    for (int k = min_k; k <= max_k; k++) {
        rr_t t; rr_clear(&t); t.kmer = k;
        find_unit_for_k(c, qs, qe, &t);
        float ratio = rr_ratio(&t);
        if (best < ratio && c->min_match_ratio <= ratio && MIN_NUM_FREQ_UNIT < t.copies && MIN_PERIOD <= t.rep_period) { best = ratio; *out = t; }
    }
}

static void ensure_read_buffers(mtro_ctx *c, int L, int r)
{
    size_t n = (size_t)L + 2 * (size_t)r;
    if (n + 8 > c->di_cap) {
        c->di_cap = n + 8 + n / 4;
        c->di_tmp = (double *)xrealloc(c->di_tmp, c->di_cap * sizeof(double));
        c->di = (double *)xrealloc(c->di, c->di_cap * sizeof(double));
        c->di_end = (int *)xrealloc(c->di_end, c->di_cap * sizeof(int));
        c->di_w = (int *)xrealloc(c->di_w, c->di_cap * sizeof(int));
    }
    size_t wn = (size_t)L * 2 + (size_t)r * 4 + 64;      /* reads reach L + r - k + 2w, w < L/2 */
    if (wn > c->wrand_cap) { c->wrand_cap = wn + wn / 4; c->wrand = (int *)xrealloc(c->wrand, c->wrand_cap * sizeof(int)); }
}

static void load_read(mtro_ctx *c, const uint8_t *codes, int L)
{
    c->L = L;
    if (c->file_order) { for (int i = 0; i < L; i++) c->org[i] = codes[i]; return; }   /* handle_one_file.c:284-285 */
    c->org = (int *)xrealloc(c->org, ((size_t)L + 4) * sizeof(int));
    for (int i = 0; i < L; i++) c->org[i] = codes[i];
    c->org[L] = c->org[L + 1] = c->org[L + 2] = c->org[L + 3] = 0;
}

static int rand_len(int L) { return L < 1000 ? 100 : L / 10; }   /* handle_one_read.c:194-201 */

static void cap_g1(mtro_ctx *c, int L, int r)
{
    fprintf(c->cap, "{\"t\":\"G1\",\"L\":%d,\"r\":%d,\"ranges\":[", L, r);
    int first = 1;
    for (int i = 0; i < L; i++)
        if (c->di[i] != -1 || c->di_end[i] > -1) {
            uint64_t bits; memcpy(&bits, &c->di[i], 8);
            fprintf(c->cap, "%s[%d,%d,%d,\"%016llx\"]", first ? "" : ",", i, c->di_end[i], c->di_w[i], (unsigned long long)bits);
            first = 0;
        }
    fprintf(c->cap, "]}\n");
}

int mtro_ranges(mtro_ctx *c, const uint8_t *codes, int L, double *di, int32_t *end, int32_t *w)
{
    int r = rand_len(L);
    load_read(c, codes, L);
    ensure_read_buffers(c, L, r);
    fill_ranges(c, L, r);
    int n = 0;
    for (int i = 0; i < L; i++) {
        di[i] = c->di[i]; end[i] = c->di_end[i]; w[i] = c->di_w[i];
        if (-1 < end[i] && end[i] < L) n++;
    }
    return n;
}

/* handle_one_TR (handle_one_read.c:190-261) without the chaining call */
int mtro_process_read(mtro_ctx *c, const char *read_id, const uint8_t *codes, int L, mtro_record **out)
{
    if (L <= 0 || L > MTRO_MAX_INPUT_LENGTH) return -1;
    int r = rand_len(L);
    load_read(c, codes, L);
    ensure_read_buffers(c, L, r);
    fill_ranges(c, L, r);
    if (c->cap && c->level >= 1) cap_g1(c, L, r);
    for (int i = 0; i < L; i++) if (-1 < c->di_end[i] && c->di_end[i] < L) c->st.ranges_candidate++;

    mtro_record *recs = NULL; int n = 0, cap = 0;
    for (int qs = 0; qs < L; qs++) {
        int qe = c->di_end[qs];
        if (!(-1 < qe && qe < L)) continue;
        rr_t t; rr_clear(&t);
        find_unit(c, qs, qe, c->di_w[qs], &t);
        c->st.ranges_executed++;
        if (t.repeat_len > 0 && t.rep_start + MIN_PERIOD * MIN_NUM_FREQ_UNIT < t.rep_end) {
            if (n == cap) { cap = cap ? cap * 2 : 8; recs = (mtro_record *)xrealloc(recs, (size_t)cap * sizeof(*recs)); }
            mtro_record *o = &recs[n++];
            memset(o, 0, sizeof(*o));
            o->rep_start = t.rep_start; o->rep_end = t.rep_end; o->repeat_len = t.repeat_len; o->rep_period = t.rep_period;
            o->num_freq_unit = t.copies; o->num_matches = t.mat; o->num_mismatches = t.mis; o->num_insertions = t.ins;
            o->num_deletions = t.del; o->kmer = t.kmer; o->match_gain = t.G; o->mismatch_penalty = t.MM; o->indel_penalty = t.D;
            strcpy(o->unit, t.str);
            memcpy(o->unit_score, t.score, sizeof(t.score));
            c->st.records++;
            if (c->cap) {
                fprintf(c->cap, "{\"t\":\"G4\",\"id\":\"%s\",\"L\":%d,\"rep_start\":%d,\"rep_end\":%d,\"repeat_len\":%d,\"period\":%d,"
                                "\"copies\":%d,\"mat\":%d,\"mis\":%d,\"ins\":%d,\"del\":%d,\"k\":%d,\"G\":%d,\"MM\":%d,\"D\":%d,\"unit\":\"%s\",\"score\":[",
                        read_id ? read_id : "", L, t.rep_start, t.rep_end, t.repeat_len, t.rep_period, t.copies, t.mat, t.mis, t.ins, t.del,
                        t.kmer, t.G, t.MM, t.D, t.str);
                for (int i = 0; i < t.rep_period && i < MAX_PERIOD; i++) fprintf(c->cap, "%s%d", i ? "," : "", t.score[i]);
                fprintf(c->cap, "]}\n");
            }
            /* remove_redundant_ranges_from_directional_index, handle_one_read.c:178-188 */
            for (int i = t.rep_start; i < t.rep_end; i++)
                if (c->di[i] != -1 && c->di_end[i] < t.rep_end) { c->di[i] = -1; c->di_end[i] = -1; c->di_w[i] = -1; }
        }
    }
    *out = recs;
    return n;
}

int mtro_wrap_dp(const uint8_t *codes, int L, int qs, int qe, const uint8_t *unit, int U, int G, int MM, int D, mtro_dp_result *res)
{
    mtro_ctx *c = mtro_create(1, 0.6f);
    load_read(c, codes, L);
    int u1[MAX_PERIOD + 1];
    for (int i = 0; i < U; i++) u1[i + 1] = unit[i];
    dp_out o; memset(&o, 0, sizeof(o));
    int rc = dp_run(c, qs, qe - qs + 1, u1, U, G, MM, D, 0, &o);
    if (rc == 0) {
        res->rep_start = qs + o.stop_i + 1; res->rep_end = qs + o.end_i; res->repeat_len = o.end_i - o.stop_i;
        res->num_freq_unit = U > 0 ? o.scanned / U : 0; res->num_matches = o.mat; res->num_mismatches = o.mis;
        res->num_insertions = o.ins; res->num_deletions = o.del;
    }
    mtro_destroy(c);
    return rc;
}

/* ================================================================================================
 * chaining + printing: chaining.cpp:125-177, 243-363 (ties in insertion order), wrap_around_DP.c:57-213
 * ==============================================================================================*/
typedef struct { int key, idx, is_start, seq; } ev_t;
static int ev_cmp(const void *a, const void *b)
{
    const ev_t *x = (const ev_t *)a, *y = (const ev_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->seq - y->seq;
}

int mtro_chain(const mtro_record *recs, int n, int *chain_idx)
{
    if (n <= 0) return 0;
    ev_t *ev = (ev_t *)malloc(sizeof(ev_t) * 2 * (size_t)n);
    int *score = (int *)malloc(sizeof(int) * (size_t)n), *pred = (int *)malloc(sizeof(int) * (size_t)n);
    int *Y = (int *)malloc(sizeof(int) * (size_t)n); int ny = 0;      /* multimap by end, insertion order on ties */
    int ne = 0;
    for (int i = 0; i < n; i++) {
        score[i] = recs[i].num_matches; pred[i] = -1;
        if (recs[i].rep_start + MAX_LEN_OVERLAPPING <= recs[i].rep_end) {
            ev[ne] = (ev_t){ recs[i].rep_start, i, 1, ne }; ne++;
            ev[ne] = (ev_t){ recs[i].rep_end - MAX_LEN_OVERLAPPING, i, 0, ne }; ne++;
        }
    }
    qsort(ev, (size_t)ne, sizeof(ev_t), ev_cmp);
    for (int e = 0; e < ne; e++) {
        int a = ev[e].idx;
        /* isStart(key): key == start_x (chaining.cpp:186-191) */
        if (ev[e].key == recs[a].rep_start) {
            if (ny > 0) {
                int lim = recs[a].rep_start + MAX_LEN_OVERLAPPING, p = -1;
                /* last y (in Y order) with end <= lim such that its successor has end > lim, or the very last */
                int t, prev = 0, hit = 0;
                for (t = 0, prev = 0; t < ny; prev = t, t++)
                    if (recs[Y[prev]].rep_end <= lim && recs[Y[t]].rep_end > lim) { p = Y[prev]; hit = 1; break; }
                if (!hit && recs[Y[prev]].rep_end <= lim) p = Y[prev];
                if (p >= 0) { pred[a] = p; score[a] += score[p]; }
            }
        } else {
            int ins = 1;
            for (int t = 0; t < ny; t++) {
                if (recs[Y[t]].rep_end <= recs[a].rep_end && score[Y[t]] > score[a]) ins = 0;
                if (recs[Y[t]].rep_end > recs[a].rep_end) break;
            }
            if (ny == 0) ins = 1;
            if (ins) {
                int pos = ny;                                   /* upper bound of key = end */
                while (pos > 0 && recs[Y[pos - 1]].rep_end > recs[a].rep_end) pos--;
                memmove(Y + pos + 1, Y + pos, sizeof(int) * (size_t)(ny - pos)); Y[pos] = a; ny++;
                /* erase-with-skip loop, chaining.cpp:316-328 */
                for (int t = 0; t < ny; t++) {
                    if (recs[Y[t]].rep_end >= recs[a].rep_end && score[Y[t]] < score[a]) {
                        memmove(Y + t, Y + t + 1, sizeof(int) * (size_t)(ny - t - 1)); ny--;
                        /* iterator now points at the successor; the for's ++ skips it */
                    }
                }
            }
        }
    }
    int len = 0;
    if (ny > 0) {
        for (int a = Y[ny - 1]; a >= 0; a = pred[a]) len++;
        int p = len;
        for (int a = Y[ny - 1]; a >= 0; a = pred[a]) chain_idx[--p] = a;
    }
    free(ev); free(score); free(pred); free(Y);
    return len;
}

static void print_alignment(mtro_ctx *c, FILE *fp, const mtro_record *r)
{   /* pretty_print_alignment, wrap_around_DP.c:57-213: x_i = org[rep_start-1+i], i=1..rep_end-rep_start+1 */
    int U = r->rep_period;
    int u1[MAX_PERIOD + 1]; unit_codes(r->unit, U, u1);
    dp_out o; memset(&o, 0, sizeof(o));
    if (dp_run(c, r->rep_start - 1, r->rep_end - r->rep_start + 1, u1, U, r->match_gain, r->mismatch_penalty, r->indel_penalty, 2, &o) != 0) {
        fprintf(stderr, "You need to increse the value of WrapDPsize.\n"); exit(EXIT_FAILURE);
    }
    fprintf(fp, "match gain = %i, mismatch penalty = %i, indel penalty = %i\n\n", r->match_gain, r->mismatch_penalty, r->indel_penalty);
    for (int s = o.ncol - 1; 0 <= s; s -= ALIGN_WIDTH) {
        int e = (-1 <= s - ALIGN_WIDTH) ? s - ALIGN_WIDTH : -1;
        for (int i = s; e < i; i--) fputc(c->al_in[i], fp);
        fputc('\n', fp);
        for (int i = s; e < i; i--) fputc(c->al_sym[i], fp);
        fputc('\n', fp);
        for (int i = s; e < i; i--) fputc(c->al_rep[i], fp);
        fputs("\n\n", fp);
    }
}

static void print_chain_with(mtro_ctx *c, FILE *fp, const char *read_id, int L, const mtro_record *recs,
                             const int *chain_idx, int n_chain, int print_align);
void mtro_print_chain(FILE *fp, const char *read_id, int L, const uint8_t *codes, const mtro_record *recs,
                      const int *chain_idx, int n_chain, int print_align)
{
    mtro_ctx *c = NULL;
    if (print_align) { c = mtro_create(1, 0.6f); load_read(c, codes, L); }
    print_chain_with(c, fp, read_id, L, recs, chain_idx, n_chain, print_align);
    if (c) mtro_destroy(c);
}
/* the same for the read the context processed LAST, on the context's own orgInputString: in file-order mode a repeat
 * can end on org[L], a base of an earlier read, and the reference prints that alignment from its global array */
void mtro_print_chain_of_last_read(mtro_ctx *c, FILE *fp, const char *read_id, const mtro_record *recs,
                                   const int *chain_idx, int n_chain, int print_align)
{
    print_chain_with(c, fp, read_id, c->L, recs, chain_idx, n_chain, print_align);
}
static void print_chain_with(mtro_ctx *c, FILE *fp, const char *read_id, int L, const mtro_record *recs,
                             const int *chain_idx, int n_chain, int print_align)
{
    for (int t = 0; t < n_chain; t++) {
        const mtro_record *r = &recs[chain_idx[t]];
        fprintf(fp, "%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%f\t%d\t%d\t%d\t%s\n", read_id, L, r->rep_start + 1, r->rep_end + 1,
                r->repeat_len, r->rep_period, r->num_freq_unit, r->num_matches, (float)r->num_matches / r->repeat_len,
                r->num_mismatches, r->num_insertions, r->num_deletions, r->unit);
        if (print_align) { fputc('\n', fp); print_alignment(c, fp, r); }
    }
}

/* ================================================================================================
 * FASTA reader: handle_one_file.c:169-269 (4096-byte fgets chunks, whole header line = ID,
 * ACGT/acgt only, stop at the first empty record)
 * ==============================================================================================*/
int mtro_read_fasta(const char *path, char ***ids_o, uint8_t ***seqs_o, int **lens_o)
{
    FILE *fp = fopen(path, "r");
    if (!fp) { fprintf(stderr, "fatal error: cannot open %s\n", path); exit(EXIT_FAILURE); }
    char buf[4096];
    char **ids = NULL; uint8_t **seqs = NULL; int *lens = NULL; int n = 0, cap = 0;
    uint8_t *cur = NULL; size_t cur_n = 0, cur_cap = 0;
    char *cur_id = NULL; int have_header = 0, stop = 0;
    while (!stop && fgets(buf, sizeof(buf), fp)) {
        if (buf[0] == '>') {
            if (have_header) {
                if (cur_n == 0) { stop = 1; break; }           /* empty record ends the run (:283) */
                if (n == cap) { cap = cap ? cap * 2 : 64; ids = (char **)xrealloc(ids, sizeof(char *) * (size_t)cap); seqs = (uint8_t **)xrealloc(seqs, sizeof(uint8_t *) * (size_t)cap); lens = (int *)xrealloc(lens, sizeof(int) * (size_t)cap); }
                ids[n] = cur_id; seqs[n] = cur; lens[n] = (int)cur_n; n++;
                cur = NULL; cur_n = cur_cap = 0; cur_id = NULL;
            }
            size_t i = 1; while (buf[i] && buf[i] != '\n' && buf[i] != '\r') i++;
            cur_id = (char *)malloc(i); memcpy(cur_id, buf + 1, i - 1); cur_id[i - 1] = 0;
            have_header = 1;
        } else {
            for (size_t i = 0; buf[i] && buf[i] != '\n' && buf[i] != '\r'; i++) {
                int code;
                switch (buf[i]) {
                case 'A': case 'a': code = 0; break; case 'C': case 'c': code = 1; break;
                case 'G': case 'g': code = 2; break; case 'T': case 't': code = 3; break;
                default: fprintf(stderr, "Invalid character: %c \n", buf[i]); exit(EXIT_FAILURE);
                }
                if (cur_n == cur_cap) { cur_cap = cur_cap ? cur_cap * 2 : 4096; cur = (uint8_t *)xrealloc(cur, cur_cap); }
                cur[cur_n++] = (uint8_t)code;
                if (MTRO_MAX_INPUT_LENGTH <= (int)cur_n) { fprintf(stderr, "fatal error: The length %d is tentatively at most %i.\n", (int)cur_n, MTRO_MAX_INPUT_LENGTH); exit(EXIT_FAILURE); }
            }
        }
    }
    if (!stop && have_header && cur_n > 0) {
        if (n == cap) { cap = cap ? cap * 2 : 64; ids = (char **)xrealloc(ids, sizeof(char *) * (size_t)cap); seqs = (uint8_t **)xrealloc(seqs, sizeof(uint8_t *) * (size_t)cap); lens = (int *)xrealloc(lens, sizeof(int) * (size_t)cap); }
        ids[n] = cur_id; seqs[n] = cur; lens[n] = (int)cur_n; n++;
    } else { free(cur); free(cur_id); }
    fclose(fp);
    *ids_o = ids; *seqs_o = seqs; *lens_o = lens;
    return n;
}
